// push.hip -- advance_p: relativistic Boris push + charge-conserving current deposition, and the
// cell-crossing path (move_p), for gfx950.
//
// Reference behaviour restated (arithmetic and operation order kept so that, compiled with
// -ffp-contract=off, every particle comes out bit-identical to the scalar CPU pipeline):
//   src/species_advance/standard/advance_p.cxx:68-177   per-particle push + in-cell deposit
//   src/species_advance/standard/move_p.c:34-134        streak splitting across cell faces
//   src/species_advance/standard/advance_p.cxx:399-472  host wrapper (constants, mover list)
//
// MI355X design (not the reference's pipeline structure):
//   * particles are struct-of-arrays and approximately cell-sorted, so a 256-thread workgroup
//     owns a contiguous chunk of PUSH_ITERS*256 particles whose cells form a short index window;
//   * the per-cell current accumulators of that window -- the chunk's own row of cells plus the
//     same x-range in the four y/z neighbour rows, where cell-crossers deposit -- live in LDS
//     ([component][slot], padded so that neither the deposits nor the flush bank-conflict);
//   * the deposition scatter conflict (many lanes, same 12 addresses) is resolved in registers:
//     lanes of a wavefront that hit the same cell are summed with DPP butterflies and one lane
//     issues the 12 LDS atomics; keys outside the window fall back to global float atomics;
//   * the window is flushed once per workgroup with fully coalesced global float atomics
//     (consecutive lanes = consecutive floats of consecutive accumulators);
//   * the accumulator array is a single copy: there is no per-pipeline replica to reduce.
// No MFMA: there is no dense contraction on this path.  The bound is HBM: 32 B read + 24 B
// written per particle (i and q are not rewritten for in-cell particles).
#include "push_device.h"

namespace vpichip {

struct PushParams {
  float qdt_2mc, cdt_dx, cdt_dy, cdt_dz;
  int np, max_nm;
  GridK g;
};

// ---- wavefront sum with DPP ------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(t);
}
// Sum over the 64 lanes; the total is returned wave-uniform.  Tree order, fp32.
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);         // row_half_mirror
  v = dpp_add<0x140>(v);         // row_mirror           -> every lane holds its row's sum
  v = dpp_add<0x142, 0xa>(v);    // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);    // row_bcast:31 into rows 2,3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__global__ __launch_bounds__(PUSH_THREADS)
void advance_p_kernel(ParticlesK p, const float4 *__restrict__ fi, float *__restrict__ g_acc,
                      vpic_particle_mover_t *__restrict__ pm, int *__restrict__ nm_counter,
                      const PushParams P, const unsigned n_chunks) {
  __shared__ float s_acc[12 * NSLOT_PAD];
  __shared__ int s_wbase;

  const unsigned chunk = xcd_block(blockIdx.x, gridDim.x);
  if (chunk >= n_chunks) return;                       // whole workgroup leaves together
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int first = (int)chunk * (PUSH_THREADS * PUSH_ITERS);

  for (int k = tid; k < 12 * NSLOT_PAD; k += PUSH_THREADS) s_acc[k] = 0.f;
  if (tid == 0) s_wbase = p.i[first] - WMARGIN;
  __syncthreads();
  const int wbase = s_wbase;
  const GridK &g = P.g;

  const float one = 1.f, one_third = 1. / 3., two_fifteenths = 2. / 15.;
  const float qdt_2mc = P.qdt_2mc, cdt_dx = P.cdt_dx, cdt_dy = P.cdt_dy, cdt_dz = P.cdt_dz;

#pragma unroll 1
  for (int it = 0; it < PUSH_ITERS; it++) {
    const int idx = first + it * PUSH_THREADS + tid;
    const bool active = idx < P.np;
    int key = -1;
    float a[12];
#pragma unroll
    for (int k = 0; k < 12; k++) a[k] = 0.f;

    if (active) {
      float dx = p.dx[idx], dy = p.dy[idx], dz = p.dz[idx];
      const int ii = p.i[idx];
      float ux = p.ux[idx], uy = p.uy[idx], uz = p.uz[idx];
      const float q = p.q[idx];
      const float4 *f = fi + (size_t)ii * 5;
      const float4 fe_x = f[0], fe_y = f[1], fe_z = f[2], fb0 = f[3];
      const float2 fb1 = *reinterpret_cast<const float2 *>(f + 4);

      // advance_p.cxx:74-82
      const float hax = qdt_2mc * ((fe_x.x + dy * fe_x.y) + dz * (fe_x.z + dy * fe_x.w));
      const float hay = qdt_2mc * ((fe_y.x + dz * fe_y.y) + dx * (fe_y.z + dz * fe_y.w));
      const float haz = qdt_2mc * ((fe_z.x + dx * fe_z.y) + dy * (fe_z.z + dx * fe_z.w));
      const float cbx = fb0.x + dx * fb0.y;
      const float cby = fb0.z + dy * fb0.w;
      const float cbz = fb1.x + dz * fb1.y;
      float v0, v1, v2, v3, v4, v5;
      // advance_p.cxx:87-105
      ux += hax; uy += hay; uz += haz;
      v0 = qdt_2mc / sqrtf(one + (ux * ux + (uy * uy + uz * uz)));
      v1 = cbx * cbx + (cby * cby + cbz * cbz);
      v2 = (v0 * v0) * v1;
      v3 = v0 * (one + v2 * (one_third + v2 * two_fifteenths));
      v4 = v3 / (one + v1 * (v3 * v3));
      v4 += v4;
      v0 = ux + v3 * (uy * cbz - uz * cby);
      v1 = uy + v3 * (uz * cbx - ux * cbz);
      v2 = uz + v3 * (ux * cby - uy * cbx);
      ux += v4 * (v1 * cbz - v2 * cby);
      uy += v4 * (v2 * cbx - v0 * cbz);
      uz += v4 * (v0 * cby - v1 * cbx);
      ux += hax; uy += hay; uz += haz;
      const float nux = ux, nuy = uy, nuz = uz;          // new momentum (advance_p.cxx:106-108)
      // advance_p.cxx:109-122
      v0 = one / sqrtf(one + (ux * ux + (uy * uy + uz * uz)));
      ux *= cdt_dx; uy *= cdt_dy; uz *= cdt_dz;
      ux *= v0; uy *= v0; uz *= v0;
      v0 = dx + ux; v1 = dy + uy; v2 = dz + uz;
      v3 = v0 + ux; v4 = v1 + uy; v5 = v2 + uz;

      if (v3 <= one && v4 <= one && v5 <= one && -v3 <= one && -v4 <= one && -v5 <= one) {
        p.dx[idx] = v3; p.dy[idx] = v4; p.dz[idx] = v5;
        p.ux[idx] = nux; p.uy[idx] = nuy; p.uz[idx] = nuz;
        streak12(a, q, v0, v1, v2, ux, uy, uz, q * ux * uy * uz * one_third);
        key = ii;
      } else {
        // advance_p.cxx:166-175: leaves its cell.  move_p works on position, cell and (on
        // reflection) momentum.
        int pi = ii;
        float pux = nux, puy = nuy, puz = nuz;
        float mdx = ux, mdy = uy, mdz = uz;
        const int stuck = move_p_lane(dx, dy, dz, pi, pux, puy, puz, q, mdx, mdy, mdz,
                                      s_acc, g_acc, wbase, g);
        p.dx[idx] = dx; p.dy[idx] = dy; p.dz[idx] = dz; p.i[idx] = pi;
        p.ux[idx] = pux; p.uy[idx] = puy; p.uz[idx] = puz;
        if (stuck) {
          const int slot = atomicAdd(nm_counter, 1);
          if (slot < P.max_nm) {
            vpic_particle_mover_t m; m.dispx = mdx; m.dispy = mdy; m.dispz = mdz; m.i = idx;
            pm[slot] = m;
          }
        }
      }
    }

    // ---- in-cell deposits: sum lanes that share a cell, one lane deposits ---------------------
    unsigned long long todo = __ballot(key >= 0);
    for (int gi = 0; todo && gi < MAX_GROUP_ITERS; gi++) {
      const int lead = __ffsll((long long)todo) - 1;
      const int k0 = __builtin_amdgcn_readlane(key, lead);
      const bool mine = (key == k0);
      const unsigned long long m = __ballot(mine);
      if (__popcll(m) >= MIN_GROUP) {
        float r[12];
#pragma unroll
        for (int k = 0; k < 12; k++) r[k] = wave_sum(mine ? a[k] : 0.f);
        if (lane == lead) deposit12(s_acc, g_acc, k0, window_slot(k0, wbase, g.sy, g.sz), r);
      } else if (mine) {
        deposit12(s_acc, g_acc, key, window_slot(key, wbase, g.sy, g.sz), a);
      }
      todo &= ~m;
    }
    if ((todo >> lane) & 1) deposit12(s_acc, g_acc, key, window_slot(key, wbase, g.sy, g.sz), a);
  }

  // ---- flush the window: consecutive lanes -> consecutive floats of consecutive accumulators --
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSEG; s++) {
    const int seg_base = wbase + ((s == 0) ? 0 : (s == 1) ? g.sy : (s == 2) ? -g.sy : (s == 3) ? g.sz : -g.sz);
    for (int fidx = tid; fidx < WX * 12; fidx += PUSH_THREADS) {
      const int cell = fidx / 12, k = fidx - cell * 12;
      const float v = s_acc[k * NSLOT_PAD + s * WX + cell];
      if (v != 0.f) atomicAdd(&g_acc[(size_t)(seg_base + cell) * 12 + k], v);
    }
  }
}

// ---- host side -------------------------------------------------------------------------------
static int begin_profile(Engine *e, int64_t particles) {
  if (!e->profile) return -1;
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    e->ev_pool.push_back({a, b});
    e->ev_particles.push_back(0);
  }
  const int k = (int)e->ev_used++;
  e->ev_particles[k] = particles;
  (void)hipEventRecord(e->ev_pool[k].first, e->stream);
  return k;
}

int k_advance_p(Engine *e, Species &s) {
  const vpic_hip_grid_t &g = e->grid;
  PushParams P;
  // advance_p.cxx:425-428: double for qdt_2mc, float for the cdt_d*
  P.qdt_2mc = (float)(0.5 * s.q_m * g.dt / g.cvac);
  P.cdt_dx = g.cvac * g.dt * g.rdx;
  P.cdt_dy = g.cvac * g.dt * g.rdy;
  P.cdt_dz = g.cvac * g.dt * g.rdz;
  P.np = (int)s.np;
  P.max_nm = (int)s.max_nm;
  P.g = e->gk;
  VH_CHECK(hipMemsetAsync(e->counters, 0, sizeof(int), e->stream));
  s.nm = 0;
  if (s.np > 0) {
    const unsigned n_chunks = (unsigned)((s.np + PUSH_THREADS * PUSH_ITERS - 1) / (PUSH_THREADS * PUSH_ITERS));
    const unsigned grid = (n_chunks + 7u) & ~7u;
    const int ev = begin_profile(e, s.np);
    hipLaunchKernelGGL(advance_p_kernel, dim3(grid), dim3(PUSH_THREADS), 0, e->stream,
                       s.p, reinterpret_cast<const float4 *>(e->fi), reinterpret_cast<float *>(e->acc),
                       s.pm, e->counters, P, n_chunks);
    if (ev >= 0) (void)hipEventRecord(e->ev_pool[ev].second, e->stream);
    VH_CHECK(hipGetLastError());
  }
  // the mover count decides what boundary_p does next: read it back
  VH_CHECK(hipMemcpyAsync(e->host_counters, e->counters, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  int64_t nm = e->host_counters[0];
  if (nm > s.max_nm) {
    // advance_p.cxx:463-465: a warning, the excess movers are dropped
    fprintf(stderr, "vpic_hip: advance_p ran out of storage for %lld movers\n", (long long)(nm - s.max_nm));
    nm = s.max_nm;
  }
  s.nm = nm;
  s.partition_valid = false;
  return 0;
}

// ---- energy_p: species_advance/standard/energy_p.cxx:31-47,124-157 ---------------------------
__global__ __launch_bounds__(256)
void energy_p_kernel(ParticlesK p, const float4 *__restrict__ fi, double *__restrict__ partial,
                     float qdt_2mc, int np) {
  __shared__ double s_sum[4];
  double en = 0;
  const float one = 1.f;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < np; idx += gridDim.x * 256) {
    const float dx = p.dx[idx], dy = p.dy[idx], dz = p.dz[idx];
    const float4 *f = fi + (size_t)p.i[idx] * 5;
    const float4 fe_x = f[0], fe_y = f[1], fe_z = f[2];
    float v0 = p.ux[idx] + qdt_2mc * ((fe_x.x + dy * fe_x.y) + dz * (fe_x.z + dy * fe_x.w));
    float v1 = p.uy[idx] + qdt_2mc * ((fe_y.x + dz * fe_y.y) + dx * (fe_y.z + dz * fe_y.w));
    float v2 = p.uz[idx] + qdt_2mc * ((fe_z.x + dx * fe_z.y) + dy * (fe_z.z + dx * fe_z.w));
    v0 = v0 * v0 + v1 * v1 + v2 * v2;
    v0 /= sqrtf(one + v0) + one;
    en += (double)v0 * (double)p.q[idx];
  }
  for (int off = 32; off; off >>= 1) en += __shfl_down(en, off);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = en;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}

int k_energy_p(Engine *e, Species &s, double *energy) {
  const vpic_hip_grid_t &g = e->grid;
  const float qdt_2mc = (float)(0.5 * s.q_m * g.dt / g.cvac);
  const int nb = (int)e->dsum_count;
  hipLaunchKernelGGL(energy_p_kernel, dim3(nb), dim3(256), 0, e->stream, s.p,
                     reinterpret_cast<const float4 *>(e->fi), e->dsum, qdt_2mc, (int)s.np);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipMemcpyAsync(e->host_dsum, e->dsum, sizeof(double) * nb, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  double sum = 0;
  for (int k = 0; k < nb; k++) sum += e->host_dsum[k];      // fixed order: reproducible
  *energy = (double)g.cvac * (double)g.cvac * sum / (double)s.q_m;
  return 0;
}

}  // namespace vpichip

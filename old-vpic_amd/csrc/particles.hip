// particles.hip -- particle storage conversions, sort_p and boundary_p on the device.
//
// Reference behaviour restated:
//   src/species_advance/standard/sort_p.c:16-102       counting sort by voxel, partition[nv+1]
//   src/species_advance/standard/boundary_p.c:9-71     accumulate_rhob
//   src/species_advance/standard/boundary_p.c:194-320  mover classification, removal, injectors
//   src/species_advance/standard/boundary_p.c:457-497  injection + finishing the move
// The device forms differ where the reference is inherently serial: removal back-fills holes
// with a parallel compaction instead of the reverse-order walk (the particle ORDER afterwards
// differs, the particle SET does not), and the counting sort hands out slots with atomics (order
// within a voxel is not the stable order of sort_p.c:67; the partition is identical).
#include "push_device.h"
#include <cstddef>
#include <algorithm>

namespace vpichip {

// ---- AoS <-> SoA -------------------------------------------------------------------------------
__global__ void particles_from_aos_kernel(ParticlesK p, int64_t *tag, int64_t *tag2,
                                          const vpic_particle_t *__restrict__ src, int64_t first, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const vpic_particle_t s = src[t];
  const int64_t k = first + t;
  p.dx[k] = s.dx; p.dy[k] = s.dy; p.dz[k] = s.dz; p.i[k] = s.i;
  p.ux[k] = s.ux; p.uy[k] = s.uy; p.uz[k] = s.uz; p.q[k] = s.q;
  if (tag) { tag[k] = s.tag; tag2[k] = s.tag2; }
}
__global__ void particles_to_aos_kernel(ParticlesK p, const int64_t *tag, const int64_t *tag2,
                                        vpic_particle_t *__restrict__ dst, int64_t first, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int64_t k = first + t;
  vpic_particle_t s;
  s.dx = p.dx[k]; s.dy = p.dy[k]; s.dz = p.dz[k]; s.i = p.i[k];
  s.ux = p.ux[k]; s.uy = p.uy[k]; s.uz = p.uz[k]; s.q = p.q[k];
  s.tag = tag ? tag[k] : 0; s.tag2 = tag ? tag2[k] : 0;
  dst[t] = s;
}

static int alloc_particles_raw_impl(ParticlesK &p, int64_t n);
static int alloc_particles_raw(ParticlesK &p, int64_t n) { return alloc_particles_raw_impl(p, n); }
static const int64_t CHUNK = 8 << 20;   // particles per staging round trip (384 MiB)

static int ensure_tags(Engine *e, Species &s) {
  if (s.tag) return 0;
  VH_CHECK(hipMalloc(&s.tag, sizeof(int64_t) * s.max_np));
  VH_CHECK(hipMalloc(&s.tag2, sizeof(int64_t) * s.max_np));
  VH_CHECK(hipMemsetAsync(s.tag, 0, sizeof(int64_t) * s.max_np, e->stream));
  VH_CHECK(hipMemsetAsync(s.tag2, 0, sizeof(int64_t) * s.max_np, e->stream));
  return 0;
}

// host[0..n) -> particles [at, at+n) of the species; at = 0 replaces the list, at = np appends to it
int k_particles_from_aos(Engine *e, Species &s, const vpic_particle_t *host, int64_t n_new, int64_t at) {
  const int64_t np = at + n_new;
  if (np > s.max_np) VH_FAIL("species particles: np=%lld exceeds max_np=%lld", (long long)np, (long long)s.max_np);
  {                                                       // a particle outside the interior voxels would send the push out of bounds
    const GridK &g = e->gk;
    for (int64_t k = 0; k < n_new; k++) {
      const int v = host[k].i, z = v / g.sz, r = v - z * g.sz, y = r / g.sy, x = r - y * g.sy;
      if (v < 0 || (unsigned)(x - 1) >= (unsigned)g.nx || (unsigned)(y - 1) >= (unsigned)g.ny || (unsigned)(z - 1) >= (unsigned)g.nz)
        VH_FAIL("particle %lld is not in an interior voxel (i = %d)", (long long)k, v);
    }
  }
  bool any_tag = false;
  for (int64_t k = 0; k < n_new && !any_tag; k++) any_tag = host[k].tag != 0 || host[k].tag2 != 0;
  if (any_tag) { if (ensure_tags(e, s)) return 1; s.has_tags = true; }
  bool any_q = false;                                    // a species of charge-0 copies (tracers) deposits nothing
  for (int64_t k = 0; k < n_new; k++) { const float aq = fabsf(host[k].q); if (aq > s.q_max) s.q_max = aq; any_q = any_q || aq != 0; }
  s.chargeless = np > 0 && !any_q && (at == 0 || s.np == 0 || s.chargeless);
  for (int64_t first = 0; first < n_new; first += CHUNK) {
    const int n = (int)((n_new - first < CHUNK) ? n_new - first : CHUNK);
    if (ensure_stage(e, sizeof(vpic_particle_t) * (size_t)n)) return 1;
    VH_CHECK(hipMemcpyAsync(e->stage, host + first, sizeof(vpic_particle_t) * (size_t)n, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(particles_from_aos_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, s.p,
                       s.has_tags ? s.tag : nullptr, s.tag2, (const vpic_particle_t *)e->stage, at + first, n);
    VH_CHECK(hipGetLastError());
    VH_CHECK(hipStreamSynchronize(e->stream));
  }
  s.np = np; if (at == 0) { s.nm = 0; s.tile_valid = false; s.n_holes = 0; } s.partition_valid = false; s.hist_valid = false;
  return 0;
}

int k_particles_to_aos(Engine *e, Species &s, vpic_particle_t *host, int64_t cap, int64_t from, int64_t count) {
  if (s.n_holes > 0 && k_compact(e, s)) return 1;        // the host sees live particles only
  if (count < 0) count = s.np - from;
  if (from < 0 || from + count > s.np) VH_FAIL("species_get_particles: range [%lld, %lld) of %lld particles", (long long)from, (long long)(from + count), (long long)s.np);
  if (cap < count) VH_FAIL("species_get_particles: buffer holds %lld, asked for %lld", (long long)cap, (long long)count);
  host -= from;
  for (int64_t first = from; first < from + count; first += CHUNK) {
    const int n = (int)((from + count - first < CHUNK) ? from + count - first : CHUNK);
    if (ensure_stage(e, sizeof(vpic_particle_t) * (size_t)n)) return 1;
    hipLaunchKernelGGL(particles_to_aos_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, s.p,
                       s.has_tags ? s.tag : nullptr, s.tag2, (vpic_particle_t *)e->stage, first, n);
    VH_CHECK(hipGetLastError());
    VH_CHECK(hipMemcpyAsync(host + first, e->stage, sizeof(vpic_particle_t) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    VH_CHECK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

// ---- synthetic loader -------------------------------------------------------------------------
// What a deck's `repeat(N) inject_particle(uniform_rand..., maxwellian_rand...)` loop does
// (src/vpic/vpic.hxx:491-505, src/vpic/misc.cxx:16-105), done on the device for benchmark-sized
// species: ppc particles in every interior cell, uniform in the cell, drifting Maxwellian momenta.
// Cell-sorted by construction.  A counter-based hash replaces the host Mersenne twister, so the
// particles are synthetic, not the ones a reference deck would load.
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float u01(unsigned seed, unsigned idx, unsigned k) {
  const unsigned h = hash32(hash32(idx * 9u + k) ^ hash32(seed * 0x9e3779b9u + k));
  return ((h >> 8) + 0.5f) * (1.0f / 16777216.0f);     // (0,1)
}
__global__ __launch_bounds__(256)
void load_maxwellian_kernel(ParticlesK p, GridK g, int ppc, int np, unsigned seed, float q,
                            float ux0, float uy0, float uz0, float vth) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= np) return;
  const int cell = idx / ppc;
  const int cz = cell / (g.nx * g.ny), r = cell - cz * g.nx * g.ny, cy = r / g.nx, cx = r - cy * g.nx;
  p.i[idx] = (cx + 1) + g.sy * (cy + 1) + g.sz * (cz + 1);
  p.dx[idx] = 2.f * u01(seed, idx, 0) - 1.f;
  p.dy[idx] = 2.f * u01(seed, idx, 1) - 1.f;
  p.dz[idx] = 2.f * u01(seed, idx, 2) - 1.f;
  const float r1 = sqrtf(-2.f * logf(u01(seed, idx, 3))), a1 = 6.2831853f * u01(seed, idx, 4);
  const float r2 = sqrtf(-2.f * logf(u01(seed, idx, 5))), a2 = 6.2831853f * u01(seed, idx, 6);
  p.ux[idx] = ux0 + vth * r1 * cosf(a1);
  p.uy[idx] = uy0 + vth * r1 * sinf(a1);
  p.uz[idx] = uz0 + vth * r2 * cosf(a2);
  p.q[idx] = q;
}

int k_load_maxwellian(Engine *e, Species &s, int ppc, unsigned seed, float q, float ux, float uy, float uz, float vth) {
  const int64_t np = (int64_t)e->gk.nx * e->gk.ny * e->gk.nz * ppc;
  if (ppc < 1 || np > s.max_np) VH_FAIL("load_maxwellian: %lld particles exceed max_np=%lld", (long long)np, (long long)s.max_np);
  hipLaunchKernelGGL(load_maxwellian_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, e->stream, s.p, e->gk,
                     ppc, (int)np, seed, q, ux, uy, uz, vth);
  VH_CHECK(hipGetLastError());
  s.np = np; s.nm = 0; s.partition_valid = false; s.tile_valid = false; s.n_holes = 0; s.hist_valid = false;
  s.q_max = std::max(s.q_max, fabsf(q));
  s.chargeless = q == 0.f;                             // tracer copies (decks/trecon-part/tracer.cxx:64-70): nothing to deposit
  return 0;
}

// ---- sort_p: sort_p.c:48-58 (count, partition) and :62-67 (out-of-place placement) -------------
// Lanes of a wavefront that hold the same voxel are counted together: one atomic per distinct
// voxel per wavefront (the leader = first lane holding it), whatever their positions -- a few
// steps after the last sort neighbouring cells interleave in the array and plain run-merging would
// degenerate to one atomic per particle.  leader/rank/cnt describe this lane's group; voxels beyond
// MAX_GROUPS distinct ones per wavefront fall back to groups of one.
constexpr int MAX_GROUPS = 16;
__device__ __forceinline__ void group_info(int key, bool valid, int lane, int &leader, int &rank, int &cnt) {
  unsigned long long todo = __ballot(valid);
  leader = lane; rank = 0; cnt = 1;
  for (int it = 0; todo && it < MAX_GROUPS; ++it) {
    const int lead = __ffsll((long long)todo) - 1;
    const int k0 = __builtin_amdgcn_readlane(key, lead);
    const bool mine = valid && key == k0;
    const unsigned long long m = __ballot(mine);
    if (mine) {
      leader = lead;
      rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      cnt = __popcll(m);
    }
    todo &= ~m;
  }
}

TileK make_tile_k(const GridK &g) {
  TileK t;
  t.sy = g.sy; t.sz = g.sz;
  t.ntx = (g.nx + TILE_EDGE - 1) / TILE_EDGE; t.nty = (g.ny + TILE_EDGE - 1) / TILE_EDGE; t.ntz = (g.nz + TILE_EDGE - 1) / TILE_EDGE;
  t.ntiles = t.ntx * t.nty * t.ntz;
  magic_div((unsigned)g.sy, t.mul_sy, t.sh_sy);
  magic_div((unsigned)g.sz, t.mul_sz, t.sh_sz);
  return t;
}

template <bool TILE>
__global__ __launch_bounds__(256)
void sort_count_kernel(const int *__restrict__ cell, int np, int *__restrict__ count, const TileK t) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int key = idx < np ? cell[idx] : -1;
  const bool valid = key >= 0;                            // (a dead slot -- engine.h, Species::n_holes -- is not counted: the sort drops it)
  int leader, rank, cnt;
  group_info(key, valid, lane, leader, rank, cnt);
  if (valid && lane == leader) atomicAdd(&count[sort_key<TILE>(key, t)], cnt);
}

// exclusive scan of count[0..n) -> out[0..n], three phases, 1024 entries per workgroup
__global__ __launch_bounds__(256)
void scan_local_kernel(const int *__restrict__ in, int *__restrict__ out, int *__restrict__ block_sum, int n) {
  __shared__ int s_wave[4];
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
  int v[4], t = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { v[k] = (base + k < n) ? in[base + k] : 0; t += v[k]; }
  int incl = t;
  const int lane = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
  if (lane == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wave_off += s_wave[w];
  int run = wave_off + incl - t;
#pragma unroll
  for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
  if (threadIdx.x == 255) block_sum[blockIdx.x] = wave_off + incl;
}
__global__ __launch_bounds__(256)
void scan_blocks_kernel(int *__restrict__ block_sum, int nb) {
  __shared__ int s_wave[4];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int base = 0; base < nb; base += 256) {
    const int k = base + threadIdx.x;
    const int v = k < nb ? block_sum[k] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
    if (lane == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wave_off += s_wave[w];
    const int carry = s_carry;
    if (k < nb) block_sum[k] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256)
void scan_add_kernel(int *__restrict__ out, int *__restrict__ copy, const int *__restrict__ block_sum, int n) {
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
  const int add = block_sum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < n) { const int v = out[base + k] + add; out[base + k] = v; copy[base + k] = v; }
}

// placement: each group of equal voxels reserves cnt consecutive slots with one returning atomic
template <bool TILE>
__global__ __launch_bounds__(256)
void sort_scatter_kernel(ParticlesK in, ParticlesK out, const int64_t *tin, const int64_t *t2in,
                         int64_t *tout, int64_t *t2out, int np, int *__restrict__ next, const TileK t) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int key = idx < np ? in.i[idx] : -1;
  const bool valid = key >= 0;
  // issue the particle loads before the slot reservation so that both round trips overlap
  const int li = idx < np ? idx : 0;
  const float dx = in.dx[li], dy = in.dy[li], dz = in.dz[li];
  const float ux = in.ux[li], uy = in.uy[li], uz = in.uz[li], q = in.q[li];
  int leader, rank, cnt;
  group_info(key, valid, lane, leader, rank, cnt);
  int base = 0;
  if (valid && lane == leader) base = atomicAdd(&next[sort_key<TILE>(key, t)], cnt);
  base = __shfl(base, leader);
  if (!valid) return;
  const int dst = base + rank;
  out.dx[dst] = dx; out.dy[dst] = dy; out.dz[dst] = dz; out.i[dst] = key;
  out.ux[dst] = ux; out.uy[dst] = uy; out.uz[dst] = uz; out.q[dst] = q;
  if (tin) { tout[dst] = tin[idx]; t2out[dst] = t2in[idx]; }
}

// ---- the same counting sort, a workgroup at a time (round 3) ---------------------------------------------------------------
// One atomic per distinct key per WAVEFRONT (above) still costs the count 4.3 GB at 1.2 TB/s and leaves the scatter writing
// fragments of a few particles (0.37 of the roofline).  A workgroup's 2048 consecutive particles hold some 50-150 distinct
// keys a few steps after a sort: they are counted in an LDS hash table (key -> count), ONE global atomic per distinct key per
// workgroup counts / reserves, and the scatter moves each of the eight arrays through an 8 KB LDS buffer in destination order,
// so that what goes to one key leaves as one contiguous run.  Same result as the kernels above (partition[] / tpart[]
// identical; the order within a key is the atomics' order, as before).
#ifndef VPIC_HIP_SORT_THREADS
#define VPIC_HIP_SORT_THREADS 256
#endif
// The scatter's chunks are dealt out so that every XCD walks ONE contiguous eighth of the array (xcd_block, engine.h): what a
// chunk sends to the neighbouring tiles then lands in lines its own XCD's L2 holds -- the fragments of a few particles merge
// there instead of reaching HBM as partial lines from eight L2s.  Measured (A/B, one box, sorts of a species 10 steps after the
// last): 256^3 x 64 ppc 23.8 -> 19.6 ms, 128^3 x 32 ppc 1.56 -> 1.71 ms on the first two such sorts and level after.
#ifndef VPIC_HIP_SORT_XCD
#define VPIC_HIP_SORT_XCD 1
#endif
constexpr int WG_T = VPIC_HIP_SORT_THREADS, WG_PER_THREAD = 8, WG_CHUNK = WG_T * WG_PER_THREAD, WG_TABLE = 512, WG_EPT = WG_TABLE / WG_T;
static_assert(WG_T == 256 || WG_T == 512, "threads of a sort workgroup");

// slot of `skey` in the workgroup's table (claims one when the key is new), or -1 when the table is full
__device__ __forceinline__ int wg_slot(int *s_key, int skey) {
  unsigned h = ((unsigned)skey * 2654435761u) >> 23;                  // 9 bits
  for (int probe = 0; probe < WG_TABLE; probe++) {
    const int prev = atomicCAS(&s_key[h], -1, skey);
    if (prev == -1 || prev == skey) return (int)h;
    h = (h + 1) & (WG_TABLE - 1);
  }
  return -1;
}

template <bool TILE>
__global__ __launch_bounds__(WG_T)
void wg_count_kernel(const int *__restrict__ cell, int np, int *__restrict__ count, const TileK t) {
  __shared__ int s_key[WG_TABLE], s_cnt[WG_TABLE];
  for (int k = threadIdx.x; k < WG_TABLE; k += WG_T) { s_key[k] = -1; s_cnt[k] = 0; }
  __syncthreads();
  const int first = blockIdx.x * WG_CHUNK;
  int key[WG_PER_THREAD];
#pragma unroll
  for (int j = 0; j < WG_PER_THREAD; j++) { const int idx = first + j * WG_T + threadIdx.x; key[j] = idx < np ? cell[idx] : -1; }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < WG_PER_THREAD; j++) {
    // the lanes of a wavefront that hold the same key act as one (group_info): in a species that is still close to sorted most
    // of the 64 do, and 64 LDS atomics on one address would take their turns
    const bool valid = key[j] >= 0;                                    // (not beyond the end, not a dead slot)
    int leader, rank, cnt;
    group_info(key[j], valid, lane, leader, rank, cnt);
    if (valid && lane == leader) {
      const int skey = sort_key<TILE>(key[j], t);
      const int h = wg_slot(s_key, skey);
      if (h >= 0) atomicAdd(&s_cnt[h], cnt); else atomicAdd(&count[skey], cnt);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < WG_TABLE; k += WG_T) if (s_key[k] >= 0) atomicAdd(&count[s_key[k]], s_cnt[k]);
}

// COARSE: grouped by tile only (every cell of a tile shares the tile's first key): the same machinery with a few dozen long
// runs per chunk (see the sort by tile below)
template <bool TILE, bool COARSE = false>
__global__ __launch_bounds__(WG_T)
void wg_scatter_kernel(ParticlesK in, ParticlesK out, const int64_t *tin, const int64_t *t2in,
                       int64_t *tout, int64_t *t2out, int np, int *__restrict__ next, const TileK t) {
  // (the four tables are dead once every particle knows its place: their 8 KB then serve as the second staging buffer)
  __shared__ int s_tables[WG_CHUNK > 4 * WG_TABLE ? WG_CHUNK : 4 * WG_TABLE];
  int *const s_key = s_tables, *const s_cnt = s_tables + WG_TABLE, *const s_lbase = s_tables + 2 * WG_TABLE, *const s_gbase = s_tables + 3 * WG_TABLE;
  __shared__ int s_dst[WG_CHUNK];
  __shared__ float s_stage[WG_CHUNK];
  __shared__ int s_wave[WG_T / 64], s_total;
  for (int k = threadIdx.x; k < WG_TABLE; k += WG_T) { s_key[k] = -1; s_cnt[k] = 0; }
  __syncthreads();
#if VPIC_HIP_SORT_XCD
  if ((long long)xcd_block(blockIdx.x, gridDim.x) * WG_CHUNK >= np) return;   // (the grid is a multiple of 8; the whole workgroup leaves)
  const int first = (int)xcd_block(blockIdx.x, gridDim.x) * WG_CHUNK;
#else
  if ((long long)blockIdx.x * WG_CHUNK >= np) return;
  const int first = blockIdx.x * WG_CHUNK;
#endif
  int slot[WG_PER_THREAD], rank[WG_PER_THREAD], key[WG_PER_THREAD];
#pragma unroll
  for (int j = 0; j < WG_PER_THREAD; j++) { const int idx = first + j * WG_T + threadIdx.x; key[j] = idx < np ? in.i[idx] : -1; }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < WG_PER_THREAD; j++) {
    const bool valid = key[j] >= 0;
    int leader, r, cnt, h = -2, base = 0;
    const int gkey = COARSE ? (valid ? sort_key<true>(key[j], t) / TILE_CELLS : -1) : key[j];
    group_info(gkey, valid, lane, leader, r, cnt);                     // equal keys of a wavefront reserve together (see wg_count_kernel)
    if (valid && lane == leader) {
      const int skey = COARSE ? gkey * TILE_CELLS : sort_key<TILE>(key[j], t);
      h = wg_slot(s_key, skey);
      if (h >= 0) base = atomicAdd(&s_cnt[h], cnt);
      else base = atomicAdd(&next[skey], cnt);                         // table full (several hundred distinct keys in one chunk): places of their own
    }
    h = __shfl(h, leader); base = __shfl(base, leader);
    slot[j] = valid ? h : -2; rank[j] = base + r;
  }
  __syncthreads();
  // where each key's run begins: in this workgroup's staging order (exclusive scan over the table) and in the output
  {
    int c[WG_EPT], t = 0;
#pragma unroll
    for (int k = 0; k < WG_EPT; k++) { c[k] = s_cnt[WG_EPT * threadIdx.x + k]; t += c[k]; }
    int incl = t;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
    if (lane == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wave_off += s_wave[w];
    int run = wave_off + incl - t;
#pragma unroll
    for (int k = 0; k < WG_EPT; k++) {
      s_lbase[WG_EPT * threadIdx.x + k] = run; run += c[k];
      if (c[k]) s_gbase[WG_EPT * threadIdx.x + k] = atomicAdd(&next[s_key[WG_EPT * threadIdx.x + k]], c[k]);
    }
    if (threadIdx.x == WG_T - 1) s_total = wave_off + incl;
  }
  __syncthreads();
  int local[WG_PER_THREAD];
#pragma unroll
  for (int j = 0; j < WG_PER_THREAD; j++) {
    local[j] = -1;
    if (slot[j] >= 0) { local[j] = s_lbase[slot[j]] + rank[j]; s_dst[local[j]] = s_gbase[slot[j]] + rank[j]; }
  }
  __syncthreads();
  const int n_staged = s_total;
  // one array at a time through a staging buffer: read in array order, written in destination order.  Two buffers take
  // turns and the next array's loads are issued before this one's stores: one barrier per array, and the loads' round trip
  // runs behind the stores instead of in front of them.
  float *const src[8] = {in.dx, in.dy, in.dz, reinterpret_cast<float *>(in.i), in.ux, in.uy, in.uz, in.q};
  float *const dst[8] = {out.dx, out.dy, out.dz, reinterpret_cast<float *>(out.i), out.ux, out.uy, out.uz, out.q};
  float *const stage2 = reinterpret_cast<float *>(s_tables);
  float v[WG_PER_THREAD];
#pragma unroll
  for (int j = 0; j < WG_PER_THREAD; j++) { const int idx = first + j * WG_T + threadIdx.x; v[j] = (slot[j] != -2) ? src[0][idx] : 0.f; }
#pragma unroll
  for (int f = 0; f < 8; f++) {                                        // (unrolled: the array pointers stay in scalar registers)
    float *const stage = (f & 1) ? stage2 : s_stage;
    float vn[WG_PER_THREAD];
    if (f + 1 < 8) {
#pragma unroll
      for (int j = 0; j < WG_PER_THREAD; j++) { const int idx = first + j * WG_T + threadIdx.x; vn[j] = (slot[j] != -2) ? src[f + 1][idx] : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < WG_PER_THREAD; j++) {
      if (local[j] >= 0) stage[local[j]] = v[j];
      else if (slot[j] == -1) dst[f][rank[j]] = v[j];                  // (table overflow: straight to its place)
    }
    __syncthreads();                                                   // (the buffer is written again two arrays on, behind the next barrier)
#pragma unroll
    for (int j = 0; j < WG_PER_THREAD; j++) { const int k = j * WG_T + threadIdx.x; if (k < n_staged) dst[f][s_dst[k]] = stage[k]; }
    if (f + 1 < 8) {
#pragma unroll
      for (int j = 0; j < WG_PER_THREAD; j++) v[j] = vn[j];
    }
  }
  if (tin) {                                                           // tags ride along unstaged (cold: species that carry tags are small)
#pragma unroll
    for (int j = 0; j < WG_PER_THREAD; j++) {
      if (slot[j] == -2) continue;
      const int idx = first + j * WG_T + threadIdx.x;
      const int d = local[j] >= 0 ? s_dst[local[j]] : rank[j];
      tout[d] = tin[idx]; t2out[d] = t2in[idx];
    }
  }
}

// ---- sort by tile only (species whose particles mostly change cell every step) --------------------------------------------
// One step after a sort such a species has no runs of equal cells left whatever the sort did inside a tile, but a sort by
// cell pays for the finer order with fragments of one or two particles per destination (3.6 ms per 67 M particles after
// three steps at vth = 0.6 c, against 1.1 for a plain copy).  Grouped by tile only, a workgroup's 2048 consecutive
// particles go to a few dozen destinations at most: the workgroup counts them in a small LDS table (key = tile), reserves
// ONE range per destination with one global atomic, and every particle's place is that range's start plus its rank in the
// workgroup -- few atomics on any one counter, and writes in runs of tens to hundreds of particles.  The scatter is
// wg_scatter_kernel<true, COARSE = true> (round 3: the arrays staged through LDS in destination order like the sort by cell;
// the first version wrote each lane's particle straight to its place, in the LDS atomics' order: 1.61 -> 1.04 ms per 67 M).
constexpr int COARSE_CHUNK = 2048, COARSE_PER_THREAD = COARSE_CHUNK / 256, COARSE_TABLE = 128;

__device__ __forceinline__ int tile_of(int voxel, const TileK &t) { return sort_key<true>(voxel, t) / TILE_CELLS; }

// slot of `tile` in the workgroup's table (claims one when the tile is new), or -1 when the table is full
__device__ __forceinline__ int coarse_slot(int *s_key, int tile) {
  unsigned h = ((unsigned)tile * 2654435761u) >> 25;                 // 7 bits
  for (int probe = 0; probe < COARSE_TABLE; probe++) {
    const int prev = atomicCAS(&s_key[h], -1, tile);
    if (prev == -1 || prev == tile) return (int)h;
    h = (h + 1) & (COARSE_TABLE - 1);
  }
  return -1;
}

__global__ __launch_bounds__(256)
void coarse_count_kernel(const int *__restrict__ cell, int np, int *__restrict__ count, const TileK t) {
  __shared__ int s_key[COARSE_TABLE], s_cnt[COARSE_TABLE];
  if (threadIdx.x < COARSE_TABLE) { s_key[threadIdx.x] = -1; s_cnt[threadIdx.x] = 0; }
  __syncthreads();
  const int first = blockIdx.x * COARSE_CHUNK;
#pragma unroll
  for (int j = 0; j < COARSE_PER_THREAD; j++) {
    const int idx = first + j * 256 + threadIdx.x;
    if (idx < np && cell[idx] >= 0) {
      const int tile = tile_of(cell[idx], t);
      const int h = coarse_slot(s_key, tile);
      if (h >= 0) atomicAdd(&s_cnt[h], 1);
      else atomicAdd(&count[tile * TILE_CELLS], 1);                  // table full (cannot happen with sane input)
    }
  }
  __syncthreads();
  if (threadIdx.x < COARSE_TABLE && s_key[threadIdx.x] >= 0) atomicAdd(&count[s_key[threadIdx.x] * TILE_CELLS], s_cnt[threadIdx.x]);
}

// Arrays carry at least one push tile of padding behind max_np: the lanes of the push kernel's last, partly filled
// wavefront tile store into it instead of branching around their stores (push.hip); the pad is zeroed once.
int alloc_particles(ParticlesK &p, int64_t n_req) {
  const int64_t n = (n_req + PUSH_TILE + PARTICLE_PAD - 1) / PARTICLE_PAD * PARTICLE_PAD;
  if (alloc_particles_raw(p, n)) return 1;
  float *arr[8] = {p.dx, p.dy, p.dz, reinterpret_cast<float *>(p.i), p.ux, p.uy, p.uz, p.q};
  for (float *a : arr) VH_CHECK(hipMemset(a + (n - PARTICLE_PAD), 0, sizeof(float) * PARTICLE_PAD));
  // The fills run on the null stream, which does not order itself against the engine's (non-blocking) stream: without this
  // wait the first sort into a fresh array could be overtaken by them -- and for a species of fewer than 2048 particles
  // the pad IS the array (found with four engines sharing one GPU: one of them lost all its particles to zeros).
  VH_CHECK(hipDeviceSynchronize());
  return 0;
}
// (Eight allocations, not one block: tried in round 3 for the sake of base + k * stride addressing in kernels short of scalar
// registers -- one physically contiguous block costs the push 7 % at 256^3 x 64 ppc and 20 % on the 512-ppc drift deck, with or
// without a skewed stride; the driver's placement of separate allocations spreads the eight streams better than any stride tried.)
static int alloc_particles_raw_impl(ParticlesK &p, int64_t n) {
  VH_CHECK(hipMalloc(&p.dx, sizeof(float) * n)); VH_CHECK(hipMalloc(&p.dy, sizeof(float) * n));
  VH_CHECK(hipMalloc(&p.dz, sizeof(float) * n)); VH_CHECK(hipMalloc(&p.i, sizeof(int) * n));
  VH_CHECK(hipMalloc(&p.ux, sizeof(float) * n)); VH_CHECK(hipMalloc(&p.uy, sizeof(float) * n));
  VH_CHECK(hipMalloc(&p.uz, sizeof(float) * n)); VH_CHECK(hipMalloc(&p.q, sizeof(float) * n));
  return 0;
}

// particles of the fullest tile, handed to the host through mapped pinned memory (256 tiles per thread at 256^3: a
// workgroup per 64 K tiles, atomicMax into the mapped word; one block took 0.66 ms there)
__global__ __launch_bounds__(256) void tile_max_kernel(const int *__restrict__ tpart, int ntiles, unsigned *__restrict__ host_word) {
  __shared__ int s_max[4];
  int m = 0;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) m = max(m, tpart[(t + 1) * TILE_CELLS] - tpart[t * TILE_CELLS]);
  for (int off = 32; off; off >>= 1) m = max(m, __shfl_down(m, off));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(host_word, (unsigned)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));   // (a DEVICE word: see k_sort_p)
}
__global__ void clear_word_kernel(unsigned *__restrict__ w) { *w = 0; }
__global__ void publish_word_kernel(unsigned *__restrict__ host_word, const unsigned *__restrict__ dev_word) { *host_word = *dev_word; }

int k_sort_p(Engine *e, Species &s, bool tile_order, bool may_fuse) {
  const TileK tk = make_tile_k(e->gk);
  // by tile only: species most of whose particles change cell every step (push.hip keeps the fraction); VPIC_HIP_TILE_COARSE=0|1 overrides
  // (measured, 128^3 x 32 ppc two-stream with adaptive sorting: vth = 0.6 c 7.6 -> 5.8 ms per step, 0.24 c 5.8 -> 4.8, 0.1 c
  // even, cold beams 3 % slower by tile only: the switch is at a fifth of the particles crossing per step)
  // Which of the two is a matter of sizes too (3 M particles per species at 50 ppc, the production deck at test size: by
  // tile only is 20 % SLOWER), so where the engine's sort policy times the cycles it is decided by measurement: the
  // other flavour is tried for a few cycles now and then, the cheaper one per step is kept (sort_due records the costs).
  // Without timings: by tile only from a fifth of the particles crossing per step.
  if (tile_order) {
    // (a species of a few million particles does not fill the GPU with 2048-particle workgroups: there the count's serial
    // LDS chains and the unordered push are slower -- not even tried below 8 M)
    const bool eligible = s.np >= ((int64_t)8 << 20) && (s.cross_frac > 0.20 || (s.coarse_order && s.cross_frac > 0.15));
    if (!eligible) { s.coarse_order = false; s.flavour_cost[0] = s.flavour_cost[1] = 0; s.flavour_cycles = 0; }
    else if (!s.adaptive) s.coarse_order = true;
    else {
      const int cur = s.coarse_order ? 1 : 0, other = 1 - cur;
      bool change = false;
      if ((s.n_cycle & 127) == 127) s.flavour_cost[other] = 0;                       // look again now and then
      if (s.flavour_cycles >= 4 && s.flavour_cost[cur] > 0) {
        if (s.flavour_cost[other] == 0) change = true;                                // never tried (or forgotten): try it
        else if (s.flavour_cost[other] < 0.97 * s.flavour_cost[cur]) change = true;   // on record and cheaper
      }
      if (change) {
        s.coarse_order = !s.coarse_order; s.flavour_cycles = 0;
      }
    }
  }
  bool coarse = tile_order && (s.coarse_order || s.chargeless);   // (nothing to deposit: no runs of equal cells to keep together)
  if (tile_order && e->knobs.tile_coarse >= 0) coarse = e->knobs.tile_coarse != 0;
  const int nv = e->gk.nv;
  // keys: voxels (the reference's order; partition[] as sort_p.c:32 leaves it), or tile-major (see engine.h)
  const int n1 = (tile_order ? tk.ntiles * TILE_CELLS : nv) + 1;
  if (!s.partition) VH_CHECK(hipMalloc(&s.partition, sizeof(int) * (nv + 1)));       // sort_p.c:32
  if (tile_order && s.tpart_count < n1) {
    if (s.tpart) VH_CHECK(hipFree(s.tpart));
    s.tpart = nullptr; s.tpart_count = 0;
    VH_CHECK(hipMalloc(&s.tpart, sizeof(int) * n1));
    s.tpart_count = n1;
  }
  const bool was_tile_valid = s.tile_valid;
  s.tile_valid = false;
  if (s.np == s.n_holes) { s.np = 0; s.n_holes = 0; }                         // nothing alive
  if (s.np == 0) return 0;                                                     // sort_p.c:35
  if (!s.aux.dx && alloc_particles(s.aux, s.max_np)) return 1;
  if (s.has_tags && !s.tag_aux) {
    VH_CHECK(hipMalloc(&s.tag_aux, sizeof(int64_t) * s.max_np));
    VH_CHECK(hipMalloc(&s.tag2_aux, sizeof(int64_t) * s.max_np));
  }
  const int np = (int)s.np;
  // The sort inside the push (Species::fuse_pending): by tile and cell, counted by the push before, the species as that push
  // left it, and the caller pushes it next -- then nothing moves here.  (Not for: tags, which ride outside the push; the
  // deterministic mode and the phased push, which run other instances; the adaptive policy, which times sort and push apart.)
  s.fuse_pending = false;
  if (may_fuse && tile_order && !coarse && was_tile_valid && !s.coarse_sorted && !s.tile_unbalanced && s.hist_valid && s.hist && s.hist_count >= n1 &&
      !s.has_tags && !e->det_acc && !e->time_kernels && !e->knobs.old_sort && s.np <= ((int64_t)1 << 30) && s.np == s.n_sorted) {
    if (s.tpart2_count < n1) {
      if (s.tpart2) VH_CHECK(hipFree(s.tpart2));
      s.tpart2 = nullptr; s.tpart2_count = 0;
      VH_CHECK(hipMalloc(&s.tpart2, sizeof(int) * n1));
      s.tpart2_count = n1;
    }
    s.fuse_pending = true; s.tile_valid = true;         // (the array is as it was: the push that follows sorts it)
    return 0;
  }
  // Which count / scatter: a workgroup at a time (LDS table of up to 512 distinct keys per 2048 particles), or -- for a hot
  // species in the reference's order, where a chunk's particles sit in nearly as many voxels as there are particles and
  // the count's table would overflow into one global atomic per particle -- the COUNT a wavefront at a time (measured, 67 M
  // particles at vth = 0.6 c by voxel: count 2.4 ms against 1.8; the scatter stays by workgroup there, 2.7 ms against 7.3)
  const bool by_wave = e->knobs.old_sort, count_by_wave = by_wave || (!tile_order && s.cross_frac > 0.15);
  int *starts = tile_order ? s.tpart : s.partition;
  if (e->time_kernels) { if (!s.ev[0]) for (int i = 0; i < 4; i++) VH_CHECK(hipEventCreate(&s.ev[i])); (void)hipEventRecord(s.ev[2], e->stream); }
  // the push before this sort may have counted already (Species::hist, push.hip): then the sort starts at its scan
  const bool counted = tile_order && !coarse && s.hist_valid && s.hist && s.hist_count >= n1;
  const int *counts = counted ? s.hist : e->sort_next;
  s.hist_valid = false;
  if (counted) {}
  else VH_CHECK(hipMemsetAsync(e->sort_next, 0, sizeof(int) * n1, e->stream));
  if (counted) {}
  else if (coarse) hipLaunchKernelGGL(coarse_count_kernel, dim3((np + COARSE_CHUNK - 1) / COARSE_CHUNK), dim3(256), 0, e->stream, s.p.i, np, e->sort_next, tk);
  else if (count_by_wave) {
    if (tile_order) hipLaunchKernelGGL(sort_count_kernel<true>, dim3((np + 255) / 256), dim3(256), 0, e->stream, s.p.i, np, e->sort_next, tk);
    else hipLaunchKernelGGL(sort_count_kernel<false>, dim3((np + 255) / 256), dim3(256), 0, e->stream, s.p.i, np, e->sort_next, tk);
  }
  else if (tile_order) hipLaunchKernelGGL(wg_count_kernel<true>, dim3((np + WG_CHUNK - 1) / WG_CHUNK), dim3(WG_T), 0, e->stream, s.p.i, np, e->sort_next, tk);
  else hipLaunchKernelGGL(wg_count_kernel<false>, dim3((np + WG_CHUNK - 1) / WG_CHUNK), dim3(WG_T), 0, e->stream, s.p.i, np, e->sort_next, tk);
  if (k_sort_scan(e, counts, starts, n1)) return 1;
  if (coarse) hipLaunchKernelGGL((wg_scatter_kernel<true, true>), dim3(((np + WG_CHUNK - 1) / WG_CHUNK + 7) / 8 * 8), dim3(WG_T), 0, e->stream, s.p, s.aux,
                             s.has_tags ? s.tag : nullptr, s.tag2, s.tag_aux, s.tag2_aux, np, e->sort_next, tk);
  else if (by_wave) {
    if (tile_order) hipLaunchKernelGGL(sort_scatter_kernel<true>, dim3((np + 255) / 256), dim3(256), 0, e->stream, s.p, s.aux,
                                       s.has_tags ? s.tag : nullptr, s.tag2, s.tag_aux, s.tag2_aux, np, e->sort_next, tk);
    else hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3((np + 255) / 256), dim3(256), 0, e->stream, s.p, s.aux,
                            s.has_tags ? s.tag : nullptr, s.tag2, s.tag_aux, s.tag2_aux, np, e->sort_next, tk);
  }
  else if (tile_order) hipLaunchKernelGGL(wg_scatter_kernel<true>, dim3(((np + WG_CHUNK - 1) / WG_CHUNK + 7) / 8 * 8), dim3(WG_T), 0, e->stream, s.p, s.aux,
                                     s.has_tags ? s.tag : nullptr, s.tag2, s.tag_aux, s.tag2_aux, np, e->sort_next, tk);
  else hipLaunchKernelGGL(wg_scatter_kernel<false>, dim3(((np + WG_CHUNK - 1) / WG_CHUNK + 7) / 8 * 8), dim3(WG_T), 0, e->stream, s.p, s.aux,
                          s.has_tags ? s.tag : nullptr, s.tag2, s.tag_aux, s.tag2_aux, np, e->sort_next, tk);
  VH_CHECK(hipGetLastError());
  if (counted && k_sort_check(e, s, starts, n1)) return 1;   // (counts taken by the push before: checked against what the scatter did)
  return k_sort_finish(e, s, tile_order, coarse);
}

int k_sort_scan(Engine *e, const int *counts, int *starts, int n1) {
  const int nb = (n1 + 1023) / 1024;
  hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(256), 0, e->stream, counts, starts, e->scan_tmp, n1);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, e->stream, e->scan_tmp, nb);
  hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, e->stream, starts, e->sort_next, e->scan_tmp, n1);
  VH_CHECK(hipGetLastError());
  return 0;
}

// the sorted particles are in s.aux (and, in tile order, their keys' first places in s.tpart): swap, book
int k_sort_finish(Engine *e, Species &s, bool tile_order, bool coarse) {
  const TileK tk = make_tile_k(e->gk);
  if (tile_order) {
    unsigned *word = reinterpret_cast<unsigned *>(e->counters + 200);      // (scratch word of the counter block; the maximum reaches the host's mapped word by a plain store)
    hipLaunchKernelGGL(clear_word_kernel, dim3(1), dim3(1), 0, e->stream, word);
    hipLaunchKernelGGL(tile_max_kernel, dim3((unsigned)std::min(256, (tk.ntiles + 255) / 256)), dim3(256), 0, e->stream, s.tpart, tk.ntiles, word);
    hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(1), 0, e->stream, s.crossed_host_dev + 1, (const unsigned *)word);
  }
  VH_CHECK(hipGetLastError());
  std::swap(s.p, s.aux);
  if (s.has_tags) { std::swap(s.tag, s.tag_aux); std::swap(s.tag2, s.tag2_aux); }
  s.np -= s.n_holes; s.n_holes = 0;                     // the dead slots were not copied
  s.partition_valid = !tile_order;
  s.tile_valid = tile_order; s.n_sorted = s.np; s.coarse_sorted = coarse;
  if (tile_order) s.tile_unbalanced = false;          // the push looks at the fullest tile of THIS sort
  if (s.crossed_dev) VH_CHECK(hipMemsetAsync(s.crossed_dev + 2, 0, sizeof(unsigned), e->stream));   // the windows sit on their tiles again (PushParams::follow)
  if (e->time_kernels) { (void)hipEventRecord(s.ev[3], e->stream); s.sort_timed = true; }
  s.sorted_once = true; s.sorted_after = s.n_push; s.prev_sum = s.t_sum; s.t_sum = 0; s.n_push = 0; s.n_cycle++;
  if (tile_order) s.flavour_cycles++;
  return 0;
}

int k_compact(Engine *e, Species &s) { return s.n_holes > 0 ? k_sort_p(e, s, s.tile_valid) : 0; }

// Room for more particles / movers (the reference grows its arrays by 1.3125 when boundary_p runs out, boundary_p.c:416-448;
// here the caller reserves BETWEEN steps, before an exchange can run out: SlabDomain does at 85 % full).  The sort's
// second buffer is dropped and allocated again by the next sort.
int k_species_reserve(Engine *e, Species &s, int64_t max_np, int64_t max_nm) {
  VH_CHECK(hipStreamSynchronize(e->stream));
  if (max_np > (1ll << 31) - 8192) VH_FAIL("a species holds at most 2^31 - 8192 particles");
  if (max_np > s.max_np) {
    ParticlesK bigger{};
    if (alloc_particles(bigger, max_np)) return 1;
    float *src[8] = {s.p.dx, s.p.dy, s.p.dz, reinterpret_cast<float *>(s.p.i), s.p.ux, s.p.uy, s.p.uz, s.p.q};
    float *dst[8] = {bigger.dx, bigger.dy, bigger.dz, reinterpret_cast<float *>(bigger.i), bigger.ux, bigger.uy, bigger.uz, bigger.q};
    for (int a = 0; a < 8; a++) {
      if (s.np > 0) VH_CHECK(hipMemcpyAsync(dst[a], src[a], sizeof(float) * (size_t)s.np, hipMemcpyDeviceToDevice, e->stream));
    }
    VH_CHECK(hipStreamSynchronize(e->stream));
    for (int a = 0; a < 8; a++) (void)hipFree(src[a]);
    s.p = bigger;
    if (s.aux.dx) {
      float *aux[8] = {s.aux.dx, s.aux.dy, s.aux.dz, reinterpret_cast<float *>(s.aux.i), s.aux.ux, s.aux.uy, s.aux.uz, s.aux.q};
      for (float *a : aux) (void)hipFree(a);
      s.aux = ParticlesK{};
    }
    if (s.tag) {
      int64_t *t = nullptr, *t2 = nullptr;
      VH_CHECK(hipMalloc(&t, sizeof(int64_t) * max_np)); VH_CHECK(hipMalloc(&t2, sizeof(int64_t) * max_np));
      VH_CHECK(hipMemsetAsync(t, 0, sizeof(int64_t) * max_np, e->stream)); VH_CHECK(hipMemsetAsync(t2, 0, sizeof(int64_t) * max_np, e->stream));
      if (s.np > 0) {
        VH_CHECK(hipMemcpyAsync(t, s.tag, sizeof(int64_t) * (size_t)s.np, hipMemcpyDeviceToDevice, e->stream));
        VH_CHECK(hipMemcpyAsync(t2, s.tag2, sizeof(int64_t) * (size_t)s.np, hipMemcpyDeviceToDevice, e->stream));
      }
      VH_CHECK(hipStreamSynchronize(e->stream));
      (void)hipFree(s.tag); (void)hipFree(s.tag2); (void)hipFree(s.tag_aux); (void)hipFree(s.tag2_aux);
      s.tag = t; s.tag2 = t2; s.tag_aux = s.tag2_aux = nullptr;
    }
    s.max_np = max_np;
  }
  if (max_nm > s.max_nm) {
    vpic_particle_mover_t *pm = nullptr;
    VH_CHECK(hipMalloc(&pm, sizeof(vpic_particle_mover_t) * max_nm));
    if (s.nm > 0) VH_CHECK(hipMemcpy(pm, s.pm, sizeof(vpic_particle_mover_t) * (size_t)s.nm, hipMemcpyDeviceToDevice));
    (void)hipFree(s.pm);
    s.pm = pm; s.max_nm = max_nm;
    // the cell-crossing path of advance_p reads the list's address and size from the species' record (push.hip)
    VH_CHECK(hipMemcpy(reinterpret_cast<char *>(s.drain_k) + offsetof(DrainParams, pm), &pm, sizeof(pm), hipMemcpyHostToDevice));
    const int cap = (int)max_nm;
    VH_CHECK(hipMemcpy(reinterpret_cast<char *>(s.drain_k) + offsetof(DrainParams, max_nm), &cap, sizeof(cap), hipMemcpyHostToDevice));
  }
  return 0;
}

// TILE order: the particles appended since the last sort (arrivals from neighbour domains, injection) sit behind the
// tiles' ranges in arrival order, all over the boundary planes; pushed like that every deposit of theirs is twelve global
// atomics (measured: 1 % of a species appended costs advance_p +50 %, 3 % a factor 2.6).  So before a push they are
// regrouped by tile among themselves -- a counting sort of the tail alone, a few per cent of the species -- and each
// tile gets a second workgroup for its share of them, with the tile's window (push.hip).
__global__ __launch_bounds__(256)
void tail_copy_back_kernel(ParticlesK dst, ParticlesK src, const int64_t *tsrc, const int64_t *t2src, int64_t *tdst, int64_t *t2dst, int n,
                           const int *__restrict__ n_live) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  if (k >= *n_live) { dst.i[k] = -1; return; }            // the tail held dead slots: they gather at its end
  dst.dx[k] = src.dx[k]; dst.dy[k] = src.dy[k]; dst.dz[k] = src.dz[k]; dst.i[k] = src.i[k];
  dst.ux[k] = src.ux[k]; dst.uy[k] = src.uy[k]; dst.uz[k] = src.uz[k]; dst.q[k] = src.q[k];
  if (tsrc) { tdst[k] = tsrc[k]; t2dst[k] = t2src[k]; }
}

static ParticlesK offset_particles(const ParticlesK &p, int64_t at) {
  ParticlesK r = p;
  r.dx += at; r.dy += at; r.dz += at; r.i += at; r.ux += at; r.uy += at; r.uz += at; r.q += at;
  return r;
}

int k_tail_sort(Engine *e, Species &s) {
  s.tail_sorted = false;
  if (!s.tile_valid || s.np <= s.n_sorted) return 0;
  const TileK tk = make_tile_k(e->gk);
  const int n1 = tk.ntiles * TILE_CELLS + 1;
  const int n = (int)(s.np - s.n_sorted), nb = (n1 + 1023) / 1024;
  if (!s.ttail) VH_CHECK(hipMalloc(&s.ttail, sizeof(int) * s.tpart_count));
  if (!s.aux.dx && alloc_particles(s.aux, s.max_np)) return 1;
  if (s.has_tags && !s.tag_aux) {
    VH_CHECK(hipMalloc(&s.tag_aux, sizeof(int64_t) * s.max_np));
    VH_CHECK(hipMalloc(&s.tag2_aux, sizeof(int64_t) * s.max_np));
  }
  const ParticlesK in = offset_particles(s.p, s.n_sorted), out = offset_particles(s.aux, s.n_sorted);
  const int64_t *tin = s.has_tags ? s.tag + s.n_sorted : nullptr, *t2in = s.has_tags ? s.tag2 + s.n_sorted : nullptr;
  int64_t *tout = s.has_tags ? s.tag_aux + s.n_sorted : nullptr, *t2out = s.has_tags ? s.tag2_aux + s.n_sorted : nullptr;
  VH_CHECK(hipMemsetAsync(e->sort_next, 0, sizeof(int) * n1, e->stream));
  hipLaunchKernelGGL(sort_count_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, in.i, n, e->sort_next, tk);
  hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(256), 0, e->stream, e->sort_next, s.ttail, e->scan_tmp, n1);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, e->stream, e->scan_tmp, nb);
  hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, e->stream, s.ttail, e->sort_next, e->scan_tmp, n1);
  hipLaunchKernelGGL(sort_scatter_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, in, out, tin, t2in, tout, t2out, n, e->sort_next, tk);
  hipLaunchKernelGGL(tail_copy_back_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, in, out, (const int64_t *)tout, (const int64_t *)t2out,
                     s.has_tags ? s.tag + s.n_sorted : nullptr, s.has_tags ? s.tag2 + s.n_sorted : nullptr, n,
                     (const int *)(s.ttail + (size_t)tk.ntiles * TILE_CELLS));
  VH_CHECK(hipGetLastError());
  s.tail_sorted = true;
  return 0;
}

// ---- boundary_p ------------------------------------------------------------------------------
// (device counters: engine.h)

// How far has a species drifted from cell order?  Descents of the voxel index along the array
// (0 right after a sort), counted on every 8th block of 256 particles: 0.5 B per particle of traffic.
// vpic_hip_step's adaptive sorting looks at it.
__global__ __launch_bounds__(256)
void disorder_kernel(const int *__restrict__ cell, int np, int *__restrict__ count) {
  const long long idx = (long long)blockIdx.x * 2048 + threadIdx.x;
  bool descent = false;
  if (idx < np && threadIdx.x > 0) descent = cell[idx] < cell[idx - 1];
  const int n = __popcll(__ballot(descent));
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(count, n);
}
int k_measure_disorder(Engine *e, Species &s, int slot) {
  VH_CHECK(hipMemsetAsync(e->counters + C_DISORDER, 0, sizeof(int), e->stream));
  if (s.np > 0) {
    hipLaunchKernelGGL(disorder_kernel, dim3((unsigned)((s.np + 2047) / 2048)), dim3(256), 0, e->stream, s.p.i, (int)s.np, e->counters + C_DISORDER);
    VH_CHECK(hipGetLastError());
  }
  VH_CHECK(hipMemcpyAsync(&e->host_miss[slot], e->counters + C_DISORDER, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  return 0;
}


struct SpeciesTable {
  ParticlesK p[MAX_SPECIES];
  vpic_particle_mover_t *pm[MAX_SPECIES];
  int max_np[MAX_SPECIES], max_nm[MAX_SPECIES];
  int64_t *tag[MAX_SPECIES], *tag2[MAX_SPECIES];   // null for species that carry no tags
  int n;
};
struct SendTable { vpic_particle_injector_t *buf[6]; int cap; vpic_particle_injector_t *local; int capf[6]; };   // capf[f] != 0: capacity of face f (else cap)
struct RetryMover { vpic_particle_mover_t m; int sp, pad[3]; };                     // a mover whose message was full (resident exchange)
// the reflux handlers as one species sees them (maxwellian_reflux.c:60-62)
struct RefluxK { int n; int code[4]; float ut_para[4], ut_perp[4]; unsigned seed, call; const float *draws; int draws_n; };   // draws: test mode, see vpic_hip_set_reflux_draws

// three uniforms in (0,1) from a counter: seed, call, species, mover (lowbias32 mixing)
__device__ __forceinline__ unsigned mix32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float unit_open(unsigned r) { return ((float)(r >> 8) + 0.5f) * (1.f / 16777216.f); }

// boundary_p.c:9-71 with atomics (several absorbed particles may share a node)
__device__ __forceinline__ void accumulate_rhob_dev(float *rhob, float dx, float dy, float dz, float q, int pi,
                                    const GridK &g, float rdx, float rdy, float rdz) {
  float w0, w1, w2, w3, w4, w5, w6, w7, t;
  t = dx; w0 = (float)(0.125 * (double)q * (double)rdx * (double)rdy * (double)rdz);
  t *= w0; w1 = w0 + t; w0 -= t;
  t = dy; w3 = 1 + t; w2 = w0 * w3; w3 *= w1; t = 1 - t; w0 *= t; w1 *= t;
  t = dz; w7 = 1 + t; w4 = w0 * w7; w5 = w1 * w7; w6 = w2 * w7; w7 *= w3;
  t = 1 - t; w0 *= t; w1 *= t; w2 *= t; w3 *= t;
  const int k = pi / g.sz, rem = pi - k * g.sz, j = rem / g.sy, i = rem - j * g.sy;
  if (i == 1)    { w0 += w0; w2 += w2; w4 += w4; w6 += w6; }
  if (i == g.nx) { w1 += w1; w3 += w3; w5 += w5; w7 += w7; }
  if (j == 1)    { w0 += w0; w1 += w1; w4 += w4; w5 += w5; }
  if (j == g.ny) { w2 += w2; w3 += w3; w6 += w6; w7 += w7; }
  if (k == 1)    { w0 += w0; w1 += w1; w2 += w2; w3 += w3; }
  if (k == g.nz) { w4 += w4; w5 += w5; w6 += w6; w7 += w7; }
  atomicAdd(&rhob[pi], w0); atomicAdd(&rhob[pi + 1], w1);
  atomicAdd(&rhob[pi + g.sy], w2); atomicAdd(&rhob[pi + g.sy + 1], w3);
  atomicAdd(&rhob[pi + g.sz], w4); atomicAdd(&rhob[pi + g.sz + 1], w5);
  atomicAdd(&rhob[pi + g.sz + g.sy], w6); atomicAdd(&rhob[pi + g.sz + g.sy + 1], w7);
}

// accumulate_rhob for a list of particles handed over by the host (inject_particle with update_rhob,
// src/vpic/misc.cxx:87-91: the particle's charge, negated, is left behind as bound charge)
__global__ __launch_bounds__(256)
void accumulate_rhob_list_kernel(float *__restrict__ rhob, const vpic_particle_t *__restrict__ p, int n, float q_scale,
                                 GridK g, float rdx, float rdy, float rdz) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const vpic_particle_t s = p[t];
  accumulate_rhob_dev(rhob, s.dx, s.dy, s.dz, q_scale * s.q, s.i, g, rdx, rdy, rdz);
}
int k_accumulate_rhob(Engine *e, const vpic_particle_t *host, int64_t n, float q_scale) {
  for (int64_t first = 0; first < n; first += CHUNK) {
    const int m = (int)((n - first < CHUNK) ? n - first : CHUNK);
    if (ensure_stage(e, sizeof(vpic_particle_t) * (size_t)m)) return 1;
    VH_CHECK(hipMemcpyAsync(e->stage, host + first, sizeof(vpic_particle_t) * (size_t)m, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(accumulate_rhob_list_kernel, dim3((m + 255) / 256), dim3(256), 0, e->stream, e->f.c[F_RHOB],
                       (const vpic_particle_t *)e->stage, m, q_scale, e->gk, e->grid.rdx, e->grid.rdy, e->grid.rdz);
    VH_CHECK(hipGetLastError());
    VH_CHECK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

// boundary_p.c:194-320: one thread per mover.  Every mover leaves the particle list: absorbed
// into rhob, or turned into an injector for the neighbour across the face it sits on.
// counts: null = nm / np are the host's values; else the device-resident {movers, particles} of this species
// (the exchange that never reads them back, vpic_hip_exchange_*): nm = min(counts[0], nm), np = counts[1].
__global__ __launch_bounds__(256)
void boundary_classify_kernel(ParticlesK p, const vpic_particle_mover_t *__restrict__ pm, int nm, int np,
                              int sp_id, GridK g, float rdx, float rdy, float rdz, float *__restrict__ rhob,
                              SendTable send, int *__restrict__ counters, int *__restrict__ tail_flag,
                              int *__restrict__ holes, RefluxK rk, float gdx, float gdy, float gdz,
                              const int *__restrict__ nm_dev = nullptr, const int *__restrict__ np_dev = nullptr,
                              RetryMover *__restrict__ retry = nullptr) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (nm_dev) {
    const int have = *nm_dev;
    if (have > nm && t == 0) atomicOr(&counters[C_OVER], 1);     // more movers than this launch covers: the rest waits for the next round
    nm = min(have, nm); np = *np_dev;
  }
  if (t >= nm) return;
  const vpic_particle_mover_t m = pm[t];
  bool parked = false;
  const int idx = m.i, new_np = np - nm;
  const float dx = p.dx[idx], dy = p.dy[idx], dz = p.dz[idx];
  const float ux = p.ux[idx], uy = p.uy[idx], uz = p.uz[idx], q = p.q[idx];
  const int pi = p.i[idx];
  bool absorb = true;                          // boundary_p.c:312-316: unknown interactions absorb
  for (int face = 0; face < 6; face++) {
    const int axis = face % 3, hi = face >= 3;
    const float d = axis == 0 ? dx : axis == 1 ? dy : dz, u = axis == 0 ? ux : axis == 1 ? uy : uz;
    const bool cond = hi ? ((d == 1.f) & (u > 0)) : ((d == -1.f) & (u < 0));     // boundary_p.c:304-309
    if (!cond) continue;
    const int code = pbc_of(g, face);
    if (code == VPIC_ABSORB_PARTICLES) break;
    if (code < VPIC_ABSORB_PARTICLES) {
      // maxwellian_reflux.c:116-175: new momentum from the flux of a Maxwellian at the wall; what is left
      // of the step is travelled with it: dr' = u' sqrt(((1+|u|^2)|dr|^2) / ((1+|u'|^2)|u|^2))
      int h = -1;
      for (int k = 0; k < rk.n; k++) if (rk.code[k] == code) h = k;
      if (h < 0) break;                                    // no parameters: absorbed (boundary_p.c:312-316)
      float u[3];
      if (rk.draws && idx < rk.draws_n) {
        // test mode: the handler's three draws (uniform, normal, normal: maxwellian_reflux.c:120-122) come from a
        // table indexed by particle, so that every refluxed particle can be compared with the reference's
        u[0] = rk.ut_para[h] * (hi ? -1.41421356237309504880f : 1.41421356237309504880f) * sqrtf(-logf(rk.draws[3 * idx]));
        u[1] = rk.ut_perp[h] * rk.draws[3 * idx + 1];
        u[2] = rk.ut_perp[h] * rk.draws[3 * idx + 2];
      } else {
        const unsigned c0 = mix32(rk.seed ^ mix32(rk.call * 0x9e3779b9u + (unsigned)sp_id) ^ mix32((unsigned)t * 3u + 1u));
        const float r0 = unit_open(mix32(c0)), r1 = unit_open(mix32(c0 + 0x68bc21ebu)), r2 = unit_open(mix32(c0 + 0x02e5be93u));
        const float rad = sqrtf(-2.f * logf(r1));
        u[0] = rk.ut_para[h] * (hi ? -1.41421356237309504880f : 1.41421356237309504880f) * sqrtf(-logf(r0));
        u[1] = rk.ut_perp[h] * rad * cosf(6.28318530717958647692f * r2);
        u[2] = rk.ut_perp[h] * rad * sinf(6.28318530717958647692f * r2);
      }
      // axis of the face gets u[0]; the other two follow cyclically (perm[][] of the reference)
      const float nux = axis == 0 ? u[0] : axis == 1 ? u[2] : u[1];
      const float nuy = axis == 0 ? u[1] : axis == 1 ? u[0] : u[2];
      const float nuz = axis == 0 ? u[2] : axis == 1 ? u[1] : u[0];
      float ddx = gdx * m.dispx, ddy = gdy * m.dispy, ddz = gdz * m.dispz;
      float ratio = ux * ux + uy * uy + uz * uz;
      ratio = sqrtf(((1.f + ratio) * (ddx * ddx + ddy * ddy + ddz * ddz)) /
                    ((1.f + (nux * nux + nuy * nuy + nuz * nuz)) * (1.17549435e-38f + ratio)));
      const int slot = atomicAdd(&counters[C_LOCAL], 1);
      if (slot < send.cap) {
        vpic_particle_injector_t inj;
        inj.dx = dx; inj.dy = dy; inj.dz = dz; inj.i = pi;
        inj.ux = nux; inj.uy = nuy; inj.uz = nuz; inj.q = q;
        inj.dispx = nux * ratio * rdx; inj.dispy = nuy * ratio * rdy; inj.dispz = nuz * ratio * rdz; inj.sp_id = sp_id;
        send.local[slot] = inj;
      }
      absorb = false;
      break;
    }
    if (code >= 0 && code != g.rank) {
      // (the lanes that get here in this turn of the loop all leave through `face`: they reserve their slots together --
      // one returning atomic per wavefront instead of one per mover on the same word)
      int slot;
      {
        const unsigned long long m = __ballot(true);
        const int lane = threadIdx.x & 63, lead = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == lead) base = atomicAdd(&counters[C_SEND + face], __popcll(m));
        slot = __builtin_amdgcn_readlane(base, lead) + __popcll(m & ((1ull << lane) - 1ull));
      }
      const int fcap = nm_dev ? send.capf[face] : (send.capf[face] ? send.capf[face] : send.cap);
      if (slot >= fcap) {
        // The message is full.  The reference grows its buffers (boundary_p.c:131-150); here both ends of a message must
        // know its size beforehand, so the particle stays where it is and its mover is parked: it is offered again in
        // the next round, and the host is told (the header's `wanted` exceeds the capacity at both ends).
        atomicOr(&counters[C_OVER], 2);
        if (retry) {
          const int r = atomicAdd(&counters[C_RETRY], 1);
          if (r < RETRY_CAP) { retry[r].m = m; retry[r].sp = sp_id; parked = true; }
          else atomicOr(&counters[C_OVER], 16);
        }
      }
      if (slot < fcap) {
        const int n = axis == 0 ? g.nx : axis == 1 ? g.ny : g.nz;
        const int stride = axis == 0 ? 1 : axis == 1 ? g.sy : g.sz;
        vpic_particle_injector_t inj;
        inj.dx = axis == 0 ? -dx : dx; inj.dy = axis == 1 ? -dy : dy; inj.dz = axis == 2 ? -dz : dz;  // :251-253
        inj.i = pi + (hi ? -(n - 1) : (n - 1)) * stride;      // :254 with ops.c:157-171, equal-sized neighbours
        inj.ux = ux; inj.uy = uy; inj.uz = uz; inj.q = q;
        inj.dispx = m.dispx; inj.dispy = m.dispy; inj.dispz = m.dispz; inj.sp_id = sp_id;
        send.buf[face][slot] = inj;
      }
      absorb = false;
      break;
    }
  }
  if (absorb) accumulate_rhob_dev(rhob, dx, dy, dz, q, pi, g, rdx, rdy, rdz);
  if (nm_dev) {
    // device-resident exchange: the slot is marked dead and stays where it is (Species::n_holes, engine.h) -- nothing
    // moves under a push that is still to come, and the tile order keeps its ranges
    if (!parked) p.i[idx] = -1;
    const unsigned long long gone = __ballot(!parked);          // (the lanes that got here; one atomic per wavefront)
    if (gone && (int)(threadIdx.x & 63) == __ffsll((long long)gone) - 1) atomicAdd(&counters[C_NHOLE + sp_id], __popcll(gone));
    return;
  }
  if (idx >= new_np) tail_flag[idx - new_np] = 1;
  else holes[atomicAdd(&counters[C_HOLES], 1)] = idx;
}

// survivors of the tail [new_np, np) fill the holes below new_np (boundary_p.c:264 r[0]=p0[--np])
__global__ void boundary_fills_kernel(const int *__restrict__ tail_flag, int nm, int new_np,
                                      int *__restrict__ counters, int *__restrict__ fills,
                                      const int *__restrict__ nm_dev = nullptr, const int *__restrict__ np_dev = nullptr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (nm_dev) { nm = min(*nm_dev, nm); new_np = *np_dev - nm; }
  if (t < nm && !tail_flag[t]) fills[atomicAdd(&counters[C_FILLS], 1)] = new_np + t;
}
// after a round has classified a species' movers: those the launch did not cover move to the front of the list
// (the extent of the particle array does not change: removals leave dead slots)
__global__ void exchange_removed_kernel(int *__restrict__ nm_dev, vpic_particle_mover_t *__restrict__ pm, int nm_cap) {
  const int have = *nm_dev, left = max(have - nm_cap, 0);
  __syncthreads();
  for (int base = 0; base < left; base += blockDim.x) {     // (left <= nm_cap or not, the copy runs front to back in whole sweeps)
    const int t = base + threadIdx.x;
    vpic_particle_mover_t m; m.dispx = m.dispy = m.dispz = 0; m.i = 0;
    if (t < left) m = pm[nm_cap + t];
    __syncthreads();
    if (t < left) pm[t] = m;
    __syncthreads();
  }
  if (threadIdx.x == 0) *nm_dev = left;
}
// the movers a round parked because their message was full go back onto their species' lists (behind what the round left)
__global__ void exchange_unpark_kernel(int *__restrict__ counters, const RetryMover *__restrict__ retry, const SpeciesTable *__restrict__ Tp) {
  const int n = min(counters[C_RETRY], RETRY_CAP);
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int s = retry[t].sp;
    const int slot = atomicAdd(&counters[C_NMS + s], 1);
    if (slot < Tp->max_nm[s]) Tp->pm[s][slot] = retry[t].m; else atomicOr(&counters[C_OVER], 8);
  }
  __syncthreads();
  if (threadIdx.x == 0) counters[C_RETRY] = 0;
}
// header of an exchange message: {injectors in the payload, 0, 0, 0}
__global__ void exchange_header_kernel(int *__restrict__ counters, int *const *__restrict__ hdr, const int *__restrict__ cap) {
  const int f = threadIdx.x;
  if (f < 6 && hdr[f]) { hdr[f][0] = min(counters[C_SEND + f], cap[f]); hdr[f][1] = counters[C_SEND + f]; hdr[f][2] = 0; hdr[f][3] = 0; counters[C_SEND + f] = 0; }
}
__global__ void boundary_backfill_kernel(ParticlesK p, int64_t *tag, int64_t *tag2, const int *__restrict__ counters,
                                         const int *__restrict__ holes, const int *__restrict__ fills) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= counters[C_HOLES]) return;
  const int d = holes[t], s = fills[t];
  p.dx[d] = p.dx[s]; p.dy[d] = p.dy[s]; p.dz[d] = p.dz[s]; p.i[d] = p.i[s];
  p.ux[d] = p.ux[s]; p.uy[d] = p.uy[s]; p.uz[d] = p.uz[s]; p.q[d] = p.q[s];
  if (tag) { tag[d] = tag[s]; tag2[d] = tag2[s]; }
}

static int ensure_lists(Engine *e, int64_t n) {
  if (n <= e->list_cap) return 0;
  if (e->hole_list) { (void)hipFree(e->hole_list); (void)hipFree(e->fill_list); (void)hipFree(e->tail_flag); }
  const int64_t cap = n + (n >> 2) + 1024;
  VH_CHECK(hipMalloc(&e->hole_list, sizeof(int) * cap));
  VH_CHECK(hipMalloc(&e->fill_list, sizeof(int) * cap));
  VH_CHECK(hipMalloc(&e->tail_flag, sizeof(int) * cap));
  e->list_cap = cap; e->tail_clean = false;
  return 0;
}

static int ensure_send(Engine *e, int64_t n) {
  if (n <= e->send_cap) return 0;
  const int64_t cap = n + (n >> 2) + 1024;
  for (int f = 0; f < 6; f++) {
    const int code = e->gk.pbc[f];
    if (!(code >= 0 && code != e->gk.rank)) continue;
    if (e->send_buf[f]) (void)hipFree(e->send_buf[f]);
    VH_CHECK(hipMalloc(&e->send_buf[f], sizeof(vpic_particle_injector_t) * cap));
  }
  if (!e->reflux.empty()) {
    if (e->local_buf) (void)hipFree(e->local_buf);
    VH_CHECK(hipMalloc(&e->local_buf, sizeof(vpic_particle_injector_t) * cap));
  }
  e->send_cap = cap;
  return 0;
}

int k_boundary_p_pack(Engine *e) {
  int64_t nm_total = 0, nm_max = 0;
  for (auto &s : e->species) { nm_total += s.nm; if (s.nm > nm_max) nm_max = s.nm; }
  for (int f = 0; f < 6; f++) e->send_count[f] = 0;
  if (nm_total == 0) return 0;
  // worst case every mover leaves through one face (boundary_p.c:131-150)
  if (!e->reflux.empty() && !e->local_buf) e->send_cap = 0;     // handlers registered after the buffers were sized
  if (ensure_send(e, nm_total) || ensure_lists(e, nm_max)) return 1;
  VH_CHECK(hipMemsetAsync(e->counters + C_SEND, 0, sizeof(int) * 6, e->stream));
  VH_CHECK(hipMemsetAsync(e->counters + C_LOCAL, 0, sizeof(int), e->stream));
  SendTable send;
  for (int f = 0; f < 6; f++) send.buf[f] = e->send_buf[f];
  send.cap = (int)e->send_cap;
  send.local = e->local_buf;
  for (int f = 0; f < 6; f++) send.capf[f] = 0;
  e->reflux_calls++;
  const vpic_hip_grid_t &G = e->grid;
  for (size_t k = 0; k < e->species.size(); k++) {
    Species &s = e->species[k];
    if (s.nm == 0) continue;
    const int nm = (int)s.nm, np = (int)s.np, nb = (nm + 255) / 256;
    VH_CHECK(hipMemsetAsync(e->counters + C_HOLES, 0, sizeof(int) * 2, e->stream));
    VH_CHECK(hipMemsetAsync(e->tail_flag, 0, sizeof(int) * nm, e->stream));
    e->tail_clean = false;
    RefluxK rk = {};
    rk.n = (int)e->reflux.size(); rk.seed = e->reflux_seed; rk.call = e->reflux_calls;
    rk.draws = e->reflux_draws; rk.draws_n = (int)e->reflux_draws_n;
    for (int h = 0; h < rk.n; h++) { rk.code[h] = e->reflux[h].code; rk.ut_para[h] = e->reflux[h].ut_para[k]; rk.ut_perp[h] = e->reflux[h].ut_perp[k]; }
    hipLaunchKernelGGL(boundary_classify_kernel, dim3(nb), dim3(256), 0, e->stream, s.p, s.pm, nm, np, (int)k,
                       e->gk, G.rdx, G.rdy, G.rdz, e->f.c[F_RHOB], send, e->counters, e->tail_flag, e->hole_list,
                       rk, G.dx, G.dy, G.dz);
    hipLaunchKernelGGL(boundary_fills_kernel, dim3(nb), dim3(256), 0, e->stream, e->tail_flag, nm, np - nm,
                       e->counters, e->fill_list);
    hipLaunchKernelGGL(boundary_backfill_kernel, dim3(nb), dim3(256), 0, e->stream, s.p,
                       s.has_tags ? s.tag : nullptr, s.tag2, e->counters, e->hole_list, e->fill_list);
    VH_CHECK(hipGetLastError());
    s.np -= s.nm;
    s.nm = 0;
    s.partition_valid = false; s.hist_valid = false;
  }
  VH_CHECK(hipMemcpyAsync(e->host_counters + C_LOCAL, e->counters + C_LOCAL, sizeof(int) * (C_SEND + 6 - C_LOCAL), hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  for (int f = 0; f < 6; f++) {
    if (e->host_counters[C_SEND + f] > e->send_cap) VH_FAIL("boundary_p: injector buffer overflow on face %d", f);
    e->send_count[f] = e->host_counters[C_SEND + f];
  }
  // refluxed particles re-enter this same domain (boundary_p.c:457-497 handles them with the received ones)
  const int n_local = e->host_counters[C_LOCAL];
  if (n_local > e->send_cap) VH_FAIL("boundary_p: reflux buffer overflow");
  if (n_local > 0 && k_boundary_p_inject(e, e->local_buf, n_local)) return 1;
  return 0;
}

// boundary_p.c:457-497: append each injector to its species and finish its move; a particle that
// stops on yet another face becomes a mover for the next round.
__global__ __launch_bounds__(256)
void boundary_inject_kernel(const SpeciesTable *__restrict__ Tp, const vpic_particle_injector_t *__restrict__ in, int n, GridK g,
                            float *__restrict__ g_acc, int *__restrict__ counters, const int64_t *__restrict__ tags,
                            const int *__restrict__ n_dev = nullptr, const double det_scale = 0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (n_dev) n = min(*n_dev, n);                       // the count travels in the message header
  if (t >= n) return;
  const vpic_particle_injector_t inj = in[t];
  const int s = inj.sp_id;
  if (s < 0 || s >= Tp->n) return;
  // One counter word takes ~90 returning atomics per microsecond: the lanes of a wavefront that append to the same species
  // reserve their slots together (a message of the per-species rounds holds one species: one atomic per wavefront).
  int idx = 0;
  {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(true);                   // the lanes that got here
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const int s0 = __builtin_amdgcn_readlane(s, lead);
      const unsigned long long m = __ballot(s == s0) & todo;
      if (s == s0) {
        int base = 0;
        if (lane == lead) base = atomicAdd(&counters[C_NP + s0], __popcll(m));
        base = __builtin_amdgcn_readlane(base, lead);
        idx = base + __popcll(m & ((1ull << lane) - 1ull));
      }
      const unsigned long long charged = __ballot(s == s0 && inj.q != 0.f);
      if (charged && lane == lead) atomicOr(&counters[C_CHARGED], 1 << s0);   // (which species received charge: once per wavefront)
      todo &= ~m;
    }
  }
  if (idx >= Tp->max_np[s]) { atomicOr(&counters[C_OVER], 4); return; }   // counted; the host reports the overflow
  float dx = inj.dx, dy = inj.dy, dz = inj.dz, ux = inj.ux, uy = inj.uy, uz = inj.uz;
  float mx = inj.dispx, my = inj.dispy, mz = inj.dispz;
  int pi = inj.i;
  const int stuck = move_p_lane<false>(dx, dy, dz, pi, ux, uy, uz, inj.q, mx, my, mz, nullptr, g_acc, NO_WINDOW, g, false, det_scale);   // (det_scale != 0: g_acc is the fixed-point accumulator)
  const ParticlesK p = Tp->p[s];
  p.dx[idx] = dx; p.dy[idx] = dy; p.dz[idx] = dz; p.i[idx] = pi;
  p.ux[idx] = ux; p.uy[idx] = uy; p.uz[idx] = uz; p.q[idx] = inj.q;
  // an injector record has no tag fields (species_advance.h:48-55): a migrating particle arrives untagged; a
  // particle the host injects with an age brings its tags along
  if (Tp->tag[s]) { Tp->tag[s][idx] = tags ? tags[2 * t] : 0; Tp->tag2[s][idx] = tags ? tags[2 * t + 1] : 0; }

  if (stuck) {
    const int slot = atomicAdd(&counters[C_NMS + s], 1);
    if (slot >= Tp->max_nm[s]) atomicOr(&counters[C_OVER], 8);
    if (slot < Tp->max_nm[s]) {
      vpic_particle_mover_t m; m.dispx = mx; m.dispy = my; m.dispz = mz; m.i = idx;
      Tp->pm[s][slot] = m;
    }
  }
}

int k_boundary_p_inject(Engine *e, const vpic_particle_injector_t *inj, int n, const int64_t *tags) {
  if (n <= 0) return 0;
  const int ns = (int)e->species.size();
  if (ns > MAX_SPECIES) VH_FAIL("boundary_p: more than %d species", MAX_SPECIES);
  SpeciesTable T;
  T.n = ns;
  for (int k = 0; k < ns; k++) {
    Species &s = e->species[k];
    T.p[k] = s.p; T.pm[k] = s.pm; T.max_np[k] = (int)s.max_np; T.max_nm[k] = (int)s.max_nm;
    T.tag[k] = s.has_tags ? s.tag : nullptr; T.tag2[k] = s.has_tags ? s.tag2 : nullptr;
    e->host_counters[C_NP + k] = (int)s.np;
    e->host_counters[C_NMS + k] = (int)s.nm;
    s.hist_valid = false;                                // (arrivals: the push's histogram no longer describes the array)
  }
  e->host_counters[C_CHARGED] = 0;
  VH_CHECK(hipMemcpyAsync(e->counters + C_NP, e->host_counters + C_NP, sizeof(int) * (2 * MAX_SPECIES + 1), hipMemcpyHostToDevice, e->stream));
  if (ensure_stage(e, sizeof(SpeciesTable))) return 1;
  VH_CHECK(hipMemcpyAsync(e->stage, &T, sizeof(SpeciesTable), hipMemcpyHostToDevice, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));     // T lives on this stack frame
  if (e->det_acc && acc_prepare_det(e)) return 1;
  hipLaunchKernelGGL(boundary_inject_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream,
                     (const SpeciesTable *)e->stage, inj, n, e->gk,
                     e->det_acc ? reinterpret_cast<float *>(e->acc64) : reinterpret_cast<float *>(e->acc), e->counters, tags,
                     (const int *)nullptr, e->det_acc ? e->acc_scale : 0.0);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipMemcpyAsync(e->host_counters + C_NP, e->counters + C_NP, sizeof(int) * (2 * MAX_SPECIES + 1), hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  for (int k = 0; k < ns; k++) {
    Species &s = e->species[k];
    if (e->host_counters[C_CHARGED] >> k & 1) s.chargeless = false;
    const int64_t np = e->host_counters[C_NP + k], nm = e->host_counters[C_NMS + k];
    if (np > s.max_np) VH_FAIL("boundary_p: species %d needs %lld particle slots, has %lld (the reference would grow the array, boundary_p.c:416-432)", k, (long long)np, (long long)s.max_np);
    if (nm > s.max_nm) VH_FAIL("boundary_p: species %d needs %lld mover slots, has %lld", k, (long long)nm, (long long)s.max_nm);
    if (np != s.np) s.partition_valid = false;
    s.np = np; s.nm = nm;
  }
  return 0;
}

// ---- boundary_p without a round trip to the host -----------------------------------------------------
// vpic_hip_exchange_pack / _inject / _finish (include/vpic_hip.h): the mover counts of advance_p, the particle
// counts and the injector counts stay in device memory; kernels are launched over capacities and read the counts
// there; a message to a neighbour is a fixed-capacity buffer {int32 header[4]; injector payload[cap]} whose header
// carries the count (boundary_p.c:333-337 sends the count ahead of the payload; here it rides in front of it).
// One read-back per step (_finish) brings np / nm and the counts to the host.
static int upload_species_table(Engine *e) {
  const int ns = (int)e->species.size();
  if (ns > MAX_SPECIES) VH_FAIL("boundary_p: more than %d species", MAX_SPECIES);
  if (!e->sp_table_dev) {
    VH_CHECK(hipMalloc(&e->sp_table_dev, sizeof(SpeciesTable)));
    VH_CHECK(hipHostMalloc(&e->sp_table_host, sizeof(SpeciesTable)));
  }
  SpeciesTable &T = *reinterpret_cast<SpeciesTable *>(e->sp_table_host);
  T.n = ns;
  for (int k = 0; k < ns; k++) {
    Species &s = e->species[k];
    T.p[k] = s.p; T.pm[k] = s.pm; T.max_np[k] = (int)s.max_np; T.max_nm[k] = (int)s.max_nm;
    T.tag[k] = s.has_tags ? s.tag : nullptr; T.tag2[k] = s.has_tags ? s.tag2 : nullptr;
  }
  VH_CHECK(hipMemcpyAsync(e->sp_table_dev, e->sp_table_host, sizeof(SpeciesTable), hipMemcpyHostToDevice, e->stream));
  return 0;
}

int k_exchange_begin(Engine *e) {
  // the host's particle counts are current here (start of a step's exchange): put them on the device
  const int ns = (int)e->species.size();
  if (ns > MAX_SPECIES) VH_FAIL("boundary_p: more than %d species", MAX_SPECIES);
  for (int k = 0; k < ns; k++) { e->host_counters[C_NP + k] = (int)e->species[k].np; e->species[k].hist_valid = false; }
  VH_CHECK(hipMemcpyAsync(e->counters + C_NP, e->host_counters + C_NP, sizeof(int) * ns, hipMemcpyHostToDevice, e->stream));
  VH_CHECK(hipMemsetAsync(e->counters + C_SEND, 0, sizeof(int) * 6, e->stream));
  VH_CHECK(hipMemsetAsync(e->counters + C_CHARGED, 0, sizeof(int) * 3, e->stream));      // charged species, overflow flags, parked movers
  VH_CHECK(hipMemsetAsync(e->counters + C_NHOLE, 0, sizeof(int) * MAX_SPECIES, e->stream));   // (dead slots made THIS step; the host keeps the total)
  if (!e->retry_buf) VH_CHECK(hipMalloc(&e->retry_buf, sizeof(RetryMover) * RETRY_CAP));
  return upload_species_table(e);
}

int k_exchange_pack(Engine *e, void *const msg[6], const int32_t cap[6], int mover_cap, uint32_t species_mask) {
  if (!e->reflux.empty()) VH_FAIL("the device-resident exchange does not serve custom particle boundary handlers");
  if (!e->retry_buf || !e->sp_table_dev) VH_FAIL("vpic_hip_exchange_pack before vpic_hip_exchange_begin");
  if (mover_cap < 1) VH_FAIL("Bad mover capacity");
  {                                                         // no launch is wider than the largest mover list
    int64_t widest = 1;
    for (auto &s : e->species) widest = std::max(widest, s.max_nm);
    if (mover_cap > widest) mover_cap = (int)std::min<int64_t>(widest, 1 << 30);
  }
  constexpr size_t XMSG = sizeof(void *) * 6 + sizeof(int) * 6 + 8;   // one message table; XSLOTS slots used in turn
  constexpr unsigned XSLOTS = 64;                                      // (a table's upload may still be pending when later rounds fill the next ones: one per species and round of a step)
  if (!e->xmsg_dev) {
    VH_CHECK(hipMalloc(&e->xmsg_dev, XMSG * XSLOTS));
    VH_CHECK(hipHostMalloc(&e->xmsg_host, XMSG * XSLOTS));
  }
  const size_t slot = (size_t)(e->xmsg_turn++ % XSLOTS) * XMSG;
  char *xh = reinterpret_cast<char *>(e->xmsg_host) + slot, *xd = reinterpret_cast<char *>(e->xmsg_dev) + slot;
  SendTable send;
  int **hdr = reinterpret_cast<int **>(xh);
  int *caps = reinterpret_cast<int *>(hdr + 6);
  int min_cap = 1 << 30;
  for (int f = 0; f < 6; f++) {
    const int code = e->gk.pbc[f];
    // (a shared face without a message this round is closed: movers bound for it are parked, see boundary_classify_kernel)
    const bool shared = code >= 0 && code != e->gk.rank && msg[f] && cap[f] >= 1;
    hdr[f] = shared ? reinterpret_cast<int *>(msg[f]) : nullptr;
    caps[f] = shared ? cap[f] : 0;
    send.buf[f] = shared ? reinterpret_cast<vpic_particle_injector_t *>(reinterpret_cast<char *>(msg[f]) + 16) : nullptr;
    send.capf[f] = caps[f];
    if (shared && cap[f] < min_cap) min_cap = cap[f];
  }
  send.cap = min_cap; send.local = nullptr;
  VH_CHECK(hipMemcpyAsync(xd, xh, sizeof(void *) * 6 + sizeof(int) * 6, hipMemcpyHostToDevice, e->stream));
  const vpic_hip_grid_t &G = e->grid;
  RefluxK rk = {};
  for (size_t k = 0; k < e->species.size(); k++) {
    if (!(species_mask >> k & 1u)) continue;
    Species &s = e->species[k];
    const int launch = (int)std::min<int64_t>(mover_cap, s.max_nm), nb = (launch + 255) / 256;
    int *nm_dev = e->counters + C_NMS + k, *np_dev = e->counters + C_NP + k;
    hipLaunchKernelGGL(boundary_classify_kernel, dim3(nb), dim3(256), 0, e->stream, s.p, s.pm, launch, 0, (int)k,
                       e->gk, G.rdx, G.rdy, G.rdz, e->f.c[F_RHOB], send, e->counters, (int *)nullptr, (int *)nullptr,
                       rk, G.dx, G.dy, G.dz, nm_dev, np_dev, reinterpret_cast<RetryMover *>(e->retry_buf));
    hipLaunchKernelGGL(exchange_removed_kernel, dim3(1), dim3(256), 0, e->stream, nm_dev, s.pm, launch);
    VH_CHECK(hipGetLastError());
    s.partition_valid = false;
  }
  hipLaunchKernelGGL(exchange_unpark_kernel, dim3(1), dim3(256), 0, e->stream, e->counters,
                     reinterpret_cast<const RetryMover *>(e->retry_buf), (const SpeciesTable *)e->sp_table_dev);
  hipLaunchKernelGGL(exchange_header_kernel, dim3(1), dim3(64), 0, e->stream, e->counters,
                     reinterpret_cast<int *const *>(xd), reinterpret_cast<const int *>(xd + sizeof(void *) * 6));
  VH_CHECK(hipGetLastError());
  return 0;
}

int k_exchange_inject(Engine *e, const void *msg, int cap) {
  if (cap < 1 || !msg) VH_FAIL("Bad exchange message");
  const int *hdr = reinterpret_cast<const int *>(msg);
  const vpic_particle_injector_t *inj = reinterpret_cast<const vpic_particle_injector_t *>(reinterpret_cast<const char *>(msg) + 16);
  if (e->det_acc && acc_prepare_det(e)) return 1;
  hipLaunchKernelGGL(boundary_inject_kernel, dim3((cap + 255) / 256), dim3(256), 0, e->stream,
                     (const SpeciesTable *)e->sp_table_dev, inj, cap, e->gk,
                     e->det_acc ? reinterpret_cast<float *>(e->acc64) : reinterpret_cast<float *>(e->acc), e->counters, (const int64_t *)nullptr, hdr,
                     e->det_acc ? e->acc_scale : 0.0);
  VH_CHECK(hipGetLastError());
  return 0;
}

// the one read-back: counters block + the message headers (species and rounds x 6 faces x sent and received)
// Flags 1 and 2 are reported, not errors: the movers concerned are still on their lists (Species::nm) and the caller offers
// them again (vpic_hip_exchange_pack with larger messages); 4, 8 and 16 mean particles were lost.
int k_exchange_finish(Engine *e, const void *const *recv, int n_recv, int32_t *headers_out, int32_t *flags_out) {
  if (n_recv < 0 || n_recv > MAX_HEADERS) VH_FAIL("Bad message list");
  VH_CHECK(hipMemcpyAsync(e->host_counters, e->counters, sizeof(int) * C_TOTAL, hipMemcpyDeviceToHost, e->stream));
  for (int k = 0; k < n_recv; k++)
    VH_CHECK(hipMemcpyAsync(e->host_counters + HEADER_BASE + 4 * k, recv[k], sizeof(int) * 4, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  const int over = e->host_counters[C_OVER];
  VH_CHECK(hipMemsetAsync(e->counters + C_OVER, 0, sizeof(int), e->stream));
  VH_CHECK(hipMemsetAsync(e->counters + C_NHOLE, 0, sizeof(int) * MAX_SPECIES, e->stream));   // (counted once: a recovery round starts from zero)
  for (size_t k = 0; k < e->species.size(); k++) {
    Species &s = e->species[k];
    if (e->host_counters[C_CHARGED] >> k & 1) s.chargeless = false;
    const int64_t np = e->host_counters[C_NP + k], nm = e->host_counters[C_NMS + k], gone = e->host_counters[C_NHOLE + k];
    if (np != s.np || gone) s.partition_valid = false;
    s.np = np > s.max_np ? s.max_np : np; s.nm = nm > s.max_nm ? s.max_nm : nm; s.n_holes += gone;
  }
  for (int k = 0; k < n_recv; k++) for (int w = 0; w < 4; w++) headers_out[4 * k + w] = e->host_counters[HEADER_BASE + 4 * k + w];
  if (flags_out) *flags_out = over;
  if (over & 4) VH_FAIL("boundary_p: a species ran out of particle slots (reserve more: vpic_hip_species_reserve; the reference grows the array, boundary_p.c:416-432)");
  if (over & 8) VH_FAIL("boundary_p: a species ran out of mover slots (vpic_hip_species_reserve)");
  if (over & 16) VH_FAIL("boundary_p: more than %d movers found their message full in one step", RETRY_CAP);
  return 0;
}

// ---- surface emitters: src/emitter/child-langmuir.c:43-97, ccube.c, ivory.c (one law, three coefficients) ------
// One thread per (component, particle of that face).  A face emits when the normal field pulls the species
// out of it (q_m * dir * E_n > 0) and |E_n| reaches the threshold: n_emit particles share the charge
// eps0 dY dZ dt sqrt(coef |q_m E_n^3| / dX), start on the face with a half-Maxwellian normal momentum, a Maxwellian
// tangential one and a uniformly random age, leave their charge, negated, in rhob and become injector records
// (record.sp_id = -1: this face does not emit).  Random numbers: the device's counter-based stream.
struct EmitParams { int sp_id, n_emit; float q_m, ut_perp, ut_para, coef, thresh, eps0, dt, cvac, d[3], rd[3]; unsigned seed, call; const double *draws; int draws_n; };   // draws: test mode, see vpic_hip_set_emit_draws
__global__ __launch_bounds__(256)
void emit_kernel(const int *__restrict__ component, int n_component, EmitParams P, const float *__restrict__ fi,
                 float *__restrict__ rhob, GridK g, vpic_particle_injector_t *__restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_component * P.n_emit) return;
  const int c = t / P.n_emit, id = component[c], i = id >> 5, type = id & 31;
  vpic_particle_injector_t inj;
  inj.sp_id = -1;
  int axis = -1; float dir = 0;
  if (type == 12) { axis = 0; dir = 1; } else if (type == 10) { axis = 1; dir = 1; } else if (type == 4) { axis = 2; dir = 1; }
  else if (type == 14) { axis = 0; dir = -1; } else if (type == 16) { axis = 1; dir = -1; } else if (type == 22) { axis = 2; dir = -1; }
  if (axis >= 0 && i > 0 && i < g.nv - g.sz - g.sy - 1) {            // a real voxel: its 8 nodes exist
    const float en = fi[(size_t)i * 20 + 4 * axis];                      // interpolator_t: ex, ey, ez lead their groups of four
    if (P.q_m * (dir * en) > 0 && fabsf(en) >= P.thresh) {
      const int ay = (axis + 1) % 3, az = (axis + 2) % 3;
      float qp = P.eps0 * P.d[ay] * P.d[az] * P.dt * sqrtf(P.coef * fabsf(P.q_m * en * en * en) / P.d[axis]) / (float)P.n_emit;
      if (P.q_m < 0) qp = -qp;
      float pos[3], u[3], age;
      pos[axis] = -dir;
      if (P.draws && t < P.draws_n) {
        // test mode: the model's six draws per particle (child-langmuir.c:60-75: mt_drand_c x 2, mt_drandn x 3,
        // mt_drand_c0) from a table, so that every emitted particle can be compared with the reference's
        const double *d = P.draws + 6 * (size_t)t;
        pos[ay] = (float)(2 * d[0] - 1); pos[az] = (float)(2 * d[1] - 1);
        u[axis] = (float)(dir * fabs((double)P.ut_para * d[2]));
        u[ay] = (float)((double)P.ut_perp * d[3]); u[az] = (float)((double)P.ut_perp * d[4]);
        age = (float)d[5];
      } else {
        const unsigned c0 = mix32(P.seed ^ mix32(P.call * 0x9e3779b9u + 0x51ed270bu) ^ mix32((unsigned)t * 7u + 3u));
        float r[6];
        for (int k = 0; k < 6; k++) r[k] = unit_open(mix32(c0 + 0x9e3779b9u * (unsigned)(k + 1)));
        const float n0 = sqrtf(-2.f * logf(r[0])) * cosf(6.28318530717958647692f * r[1]);
        const float rad = sqrtf(-2.f * logf(r[2]));
        pos[ay] = 2.f * r[4] - 1.f; pos[az] = 2.f * r[5] - 1.f;
        u[axis] = dir * fabsf(P.ut_para * n0);
        u[ay] = P.ut_perp * rad * cosf(6.28318530717958647692f * r[3]);
        u[az] = P.ut_perp * rad * sinf(6.28318530717958647692f * r[3]);
        age = unit_open(mix32(c0 + 0x3c6ef372u));
      }
      age *= P.cvac * P.dt / sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + 1.f);
      accumulate_rhob_dev(rhob, pos[0], pos[1], pos[2], -qp, i, g, P.rd[0], P.rd[1], P.rd[2]);
      inj.dx = pos[0]; inj.dy = pos[1]; inj.dz = pos[2]; inj.i = i;
      inj.ux = u[0]; inj.uy = u[1]; inj.uz = u[2]; inj.q = qp;
      inj.dispx = u[0] * age * P.rd[0]; inj.dispy = u[1] * age * P.rd[1]; inj.dispz = u[2] * age * P.rd[2];
      inj.sp_id = P.sp_id;
    }
  }
  out[t] = inj;
}
int k_emit(Engine *e, int sp, const int32_t *host_components, int n, int n_emit, float ut_perp, float ut_para, float coef, float thresh, unsigned seed) {
  const vpic_hip_grid_t &G = e->grid;
  EmitParams P = {sp, n_emit, e->species[sp].q_m, ut_perp, ut_para, coef, thresh, G.eps0, G.dt, G.cvac,
                  {G.dx, G.dy, G.dz}, {G.rdx, G.rdy, G.rdz}, seed, ++e->reflux_calls, e->emit_draws, (int)e->emit_draws_n};
  const size_t total = (size_t)n * n_emit;
  void *buf = nullptr;
  VH_CHECK(hipMalloc(&buf, sizeof(int32_t) * (size_t)n + sizeof(vpic_particle_injector_t) * total + 16));
  int *comp = (int *)buf;
  vpic_particle_injector_t *inj = (vpic_particle_injector_t *)((char *)buf + ((sizeof(int32_t) * (size_t)n + 15) & ~(size_t)15));
  VH_CHECK(hipMemcpyAsync(comp, host_components, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, comp, n, P,
                     reinterpret_cast<const float *>(e->fi), e->f.c[F_RHOB], e->gk, inj);
  int rc = hipGetLastError() != hipSuccess;
  if (!rc) rc = k_boundary_p_inject(e, inj, (int)total);
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(buf);
  return rc;
}

// Particles the host injects part-way through a step (inject_particle with an age, misc.cxx:93-103): injector
// records and their tags in HOST memory; appended to their species and moved by the displacement given.
int k_inject_aged(Engine *e, const vpic_particle_injector_t *host_inj, const int64_t *host_tags, int n) {
  for (int k = 0; k < n; k++) {
    const int sp = host_inj[k].sp_id;
    if (sp < 0 || (size_t)sp >= e->species.size()) VH_FAIL("injector %d names species %d", k, sp);
    e->species[sp].q_max = std::max(e->species[sp].q_max, fabsf(host_inj[k].q));
    {
      const GridK &g = e->gk;
      const int v = host_inj[k].i, z = v / g.sz, y = (v - z * g.sz) / g.sy, x = v - z * g.sz - y * g.sy;
      if (v < 0 || x < 1 || x > g.nx || y < 1 || y > g.ny || z < 1 || z > g.nz) VH_FAIL("injector %d is not in an interior voxel (%d)", k, v);
    }
    if (host_tags && (host_tags[2 * k] || host_tags[2 * k + 1]) && !e->species[sp].has_tags) {
      if (ensure_tags(e, e->species[sp])) return 1;
      e->species[sp].has_tags = true;
    }
  }
  void *buf = nullptr;
  const size_t bi = sizeof(vpic_particle_injector_t) * (size_t)n, bt = host_tags ? sizeof(int64_t) * 2 * (size_t)n : 0;
  VH_CHECK(hipMalloc(&buf, bi + bt));
  VH_CHECK(hipMemcpyAsync(buf, host_inj, bi, hipMemcpyHostToDevice, e->stream));
  if (bt) VH_CHECK(hipMemcpyAsync((char *)buf + bi, host_tags, bt, hipMemcpyHostToDevice, e->stream));
  const int rc = k_boundary_p_inject(e, (const vpic_particle_injector_t *)buf, n, bt ? (const int64_t *)((char *)buf + bi) : nullptr);
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(buf);
  return rc;
}

}  // namespace vpichip

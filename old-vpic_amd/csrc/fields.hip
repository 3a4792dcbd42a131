// fields.hip -- Yee-mesh kernels on struct-of-arrays fields: load_interpolator, unload_accumulator,
// clear_jf, advance_b, advance_e, energy_f, local boundary conditions and face messages.
//
// Reference behaviour restated (same arithmetic and parenthesisation; fp contraction is off):
//   src/sf_interface/load_interpolator.cxx:72-140     src/sf_interface/unload_accumulator.cxx:30-52
//   src/field_advance/standard/advance_b.c:12-14,38-40,122-158
//   src/field_advance/standard/advance_e.c:8-25,104-108,153-329
//   src/field_advance/standard/sfa.c:188-211 (clear_jf)   energy_f.c:50-82,158-178
//   src/field_advance/standard/local.c:50-122,224-296,335-368   remote.c:61-134,416-506
//
// The reference walks "interior, then left-over planes, then exterior planes" because its
// pipelines overlap MPI; every point is still written exactly once from values no other point of
// the same call writes, so here each kernel is one launch over the box 1..n+1 with per-component
// predicates.  Voxels are laid out x-fastest: consecutive lanes read consecutive floats of each
// component array (coalesced); y/z neighbours are re-read through L1/L2.
#include "push_device.h"

namespace vpichip {

#define VOX(x, y, z) ((x) + g.sy * (y) + g.sz * (z))

// Decode a linear index over the box [1..bx]x[1..by]x[1..bz] (x fastest) into a voxel.
struct Box3 { int bx, by, bz; };
__device__ __forceinline__ bool decode(const Box3 &b, unsigned t, int &x, int &y, int &z) {
  const unsigned per_z = (unsigned)b.bx * b.by;
  if (t >= per_z * (unsigned)b.bz) return false;
  const unsigned zz = t / per_z, r = t - zz * per_z, yy = r / (unsigned)b.bx;
  x = 1 + (int)(r - yy * b.bx); y = 1 + (int)yy; z = 1 + (int)zz;
  return true;
}

// The same box walked in an order that serves the L2s (clear_unload; tried on all four kernels of every step, A/B at 256^3:
// clear_unload 437 -> 367 us, advance_b 146 -> 144, advance_e 247 -> 253, load_interpolator 372 -> 419 -- the other three
// read 4-byte components whose planes the 256 MB Infinity Cache holds anyway, and keep the array order).
// A voxel's stencil reaches one row (y) and one plane (z) away.  Dealt out in array order, a row's
// neighbours were read by OTHER XCDs (consecutive workgroups go round-robin over the eight, each with an L2 of its own):
// measured on clear_unload at 256^3, 3.3 GB fetched for 0.83 GB of accumulators.  So: every XCD gets a contiguous range of
// Y-BANDS of 8 rows, and walks a band plane by plane -- the row below was read one row ago, the plane below eight rows ago,
// both by this XCD.  Speed only: every voxel of the box is still visited exactly once.
constexpr int BAND = 8;
__host__ __device__ __forceinline__ unsigned banded_count(const Box3 &b) {          // virtual indices, rows past `by` included
  return (unsigned)((b.by + BAND - 1) / BAND) * (unsigned)b.bz * BAND * (unsigned)b.bx;
}
static inline unsigned banded_grid(const Box3 &b) { return ((banded_count(b) + 255u) / 256u + 7u) / 8u * 8u; }
__device__ __forceinline__ bool decode_banded(const Box3 &b, unsigned block, unsigned nblocks, unsigned tid, int &x, int &y, int &z) {
  const unsigned t = xcd_block(block, nblocks) * 256u + tid;
  if (t >= banded_count(b)) return false;
  const unsigned r = t / (unsigned)b.bx, yy = r % BAND, r2 = r / BAND, band = r2 / (unsigned)b.bz;
  x = 1 + (int)(t - r * b.bx); z = 1 + (int)(r2 - band * b.bz); y = 1 + (int)(band * BAND + yy);
  return y <= b.by;
}

// ---- AoS <-> SoA -------------------------------------------------------------------------------
__global__ void fields_from_aos_kernel(FieldsK f, const vpic_field_t *__restrict__ src, int nv) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const float *s = reinterpret_cast<const float *>(src + v);
#pragma unroll
  for (int c = 0; c < F_NCOMP; c++) f.c[c][v] = s[c];
  if (f.m[0]) {
    const uint16_t *m = reinterpret_cast<const uint16_t *>(s + 16);
#pragma unroll
    for (int c = 0; c < M_NCOMP; c++) f.m[c][v] = m[c];
  }
}
__global__ void fields_to_aos_kernel(FieldsK f, vpic_field_t *__restrict__ dst, int nv) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  float *d = reinterpret_cast<float *>(dst + v);
#pragma unroll
  for (int c = 0; c < F_NCOMP; c++) d[c] = f.c[c][v];
  uint16_t *m = reinterpret_cast<uint16_t *>(d + 16);
#pragma unroll
  for (int c = 0; c < M_NCOMP; c++) m[c] = f.m[0] ? f.m[c][v] : (uint16_t)0;
}

int k_fields_from_aos(Engine *e, const vpic_field_t *host) {
  const int nv = e->gk.nv;
  if (ensure_stage(e, sizeof(vpic_field_t) * (size_t)nv)) return 1;
  // materials present in the host array but a single-material engine: ids must all be 0
  VH_CHECK(hipMemcpyAsync(e->stage, host, sizeof(vpic_field_t) * (size_t)nv, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(fields_from_aos_kernel, dim3((nv + 255) / 256), dim3(256), 0, e->stream, e->f,
                     (const vpic_field_t *)e->stage, nv);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipStreamSynchronize(e->stream));
  return 0;
}
int k_fields_to_aos(Engine *e, vpic_field_t *host) {
  const int nv = e->gk.nv;
  if (ensure_stage(e, sizeof(vpic_field_t) * (size_t)nv)) return 1;
  hipLaunchKernelGGL(fields_to_aos_kernel, dim3((nv + 255) / 256), dim3(256), 0, e->stream, e->f,
                     (vpic_field_t *)e->stage, nv);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipMemcpyAsync(host, e->stage, sizeof(vpic_field_t) * (size_t)nv, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  return 0;
}

// ---- load_interpolator: load_interpolator.cxx:75-120 over interior voxels ------------------------
// A voxel's 18 coefficients are one padded 80-byte record (five 16-byte vectors).  Written by their own threads they leave a
// wavefront as five stores of 64 x 16 bytes at an 80-byte stride: every store touches forty 128-byte lines.  So the records
// go through LDS (round 3): each wavefront parks its 64 records, then writes the 320 vectors in ADDRESS order -- consecutive
// lanes, consecutive 16 bytes, whole lines.  Same values, same bits (the record's last 8 bytes are padding: zero).
__global__ __launch_bounds__(256)
void load_interpolator_kernel(FieldsK f, float4 *__restrict__ fi, GridK g) {
  __shared__ float4 s_rec[4][64 * 5];
  __shared__ int s_vox[4][64];
  int x, y, z;
  const bool inside = decode(Box3{g.nx, g.ny, g.nz}, blockIdx.x * 256u + threadIdx.x, x, y, z);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (!inside) { x = y = z = 1; }
  const int v = VOX(x, y, z), vx = v + 1, vy = v + g.sy, vz = v + g.sz;
  const int vyz = vy + g.sz, vzx = vz + 1, vxy = vx + g.sy;
  const float fourth = 0.25f, half = 0.5f;
  float w0, w1, w2, w3;
  float4 o0, o1, o2, o3;
  float2 o4;
  w0 = f.c[F_EX][v]; w1 = f.c[F_EX][vy]; w2 = f.c[F_EX][vz]; w3 = f.c[F_EX][vyz];
  o0.x = fourth * ((w3 + w0) + (w1 + w2));
  o0.y = fourth * ((w3 - w0) + (w1 - w2));
  o0.z = fourth * ((w3 - w0) - (w1 - w2));
  o0.w = fourth * ((w3 + w0) - (w1 + w2));
  w0 = f.c[F_EY][v]; w1 = f.c[F_EY][vz]; w2 = f.c[F_EY][vx]; w3 = f.c[F_EY][vzx];
  o1.x = fourth * ((w3 + w0) + (w1 + w2));
  o1.y = fourth * ((w3 - w0) + (w1 - w2));
  o1.z = fourth * ((w3 - w0) - (w1 - w2));
  o1.w = fourth * ((w3 + w0) - (w1 + w2));
  w0 = f.c[F_EZ][v]; w1 = f.c[F_EZ][vx]; w2 = f.c[F_EZ][vy]; w3 = f.c[F_EZ][vxy];
  o2.x = fourth * ((w3 + w0) + (w1 + w2));
  o2.y = fourth * ((w3 - w0) + (w1 - w2));
  o2.z = fourth * ((w3 - w0) - (w1 - w2));
  o2.w = fourth * ((w3 + w0) - (w1 + w2));
  w0 = f.c[F_CBX][v]; w1 = f.c[F_CBX][vx]; o3.x = half * (w1 + w0); o3.y = half * (w1 - w0);
  w0 = f.c[F_CBY][v]; w1 = f.c[F_CBY][vy]; o3.z = half * (w1 + w0); o3.w = half * (w1 - w0);
  w0 = f.c[F_CBZ][v]; w1 = f.c[F_CBZ][vz]; o4.x = half * (w1 + w0); o4.y = half * (w1 - w0);
  float4 *rec = &s_rec[wave][lane * 5];
  rec[0] = o0; rec[1] = o1; rec[2] = o2; rec[3] = o3; rec[4] = make_float4(o4.x, o4.y, 0.f, 0.f);
  s_vox[wave][lane] = inside ? v : -1;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int j = k * 64 + lane, owner = j / 5, part = j - owner * 5;     // vector j of the wavefront's 320, in address order
    const int vo = s_vox[wave][owner];
    if (vo >= 0) fi[(size_t)vo * 5 + part] = s_rec[wave][j];
  }
}

int k_load_interpolator(Engine *e) {
  const GridK &g = e->gk;
  const unsigned n = (unsigned)g.nx * g.ny * g.nz;
  hipLaunchKernelGGL(load_interpolator_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f,
                     reinterpret_cast<float4 *>(e->fi), g);
  VH_CHECK(hipGetLastError());
  return 0;
}

// ---- unload_accumulator: unload_accumulator.cxx:30-52 over voxels 1..n+1 -------------------------
__global__ __launch_bounds__(256)
void unload_accumulator_kernel(FieldsK f, const float *__restrict__ a, GridK g, float cx, float cy, float cz) {
  int x, y, z;
  if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z);
  const float *a0 = a + (size_t)v * 12, *ax = a0 - 12, *ay = a0 - 12 * (size_t)g.sy, *az = a0 - 12 * (size_t)g.sz;
  const float *ayz = ay - 12 * (size_t)g.sz, *azx = az - 12, *axy = ax - 12 * (size_t)g.sy;
  f.c[F_JFX][v] += cx * (a0[0] + ay[1] + az[2] + ayz[3]);
  f.c[F_JFY][v] += cy * (a0[4] + az[5] + ax[6] + azx[7]);
  f.c[F_JFZ][v] += cz * (a0[8] + ax[9] + ay[10] + axy[11]);
}

int k_unload_accumulator(Engine *e) {
  if (acc_finalize(e)) return 1;                       // (deterministic mode: the fixed-point sums become the float accumulator)
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  // unload_accumulator.cxx:30-32: double arithmetic, rounded once to float
  const float cx = (float)(0.25 * G.rdy * G.rdz / G.dt);
  const float cy = (float)(0.25 * G.rdz * G.rdx / G.dt);
  const float cz = (float)(0.25 * G.rdx * G.rdy / G.dt);
  const unsigned n = (unsigned)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
  hipLaunchKernelGGL(unload_accumulator_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f,
                     reinterpret_cast<const float *>(e->acc), g, cx, cy, cz);
  VH_CHECK(hipGetLastError());
  return 0;
}

// clear_jf + unload_accumulator in one pass (sfa.c:188-211, unload_accumulator.cxx:30-52): every voxel's jf is written
// once -- 0 + c * sum inside the box 1..n+1 (the same additions, in the same order, as the two calls), 0 outside --
// instead of a memset followed by a read-modify-write; the accumulators are read as whole 16-byte groups.
__global__ __launch_bounds__(256)
void clear_unload_kernel(FieldsK f, const float4 *__restrict__ A, GridK g, float cx, float cy, float cz) {
  int x, y, z;
  if (!decode_banded(Box3{g.nx + 2, g.ny + 2, g.nz + 2}, blockIdx.x, gridDim.x, threadIdx.x, x, y, z)) return;
  x -= 1; y -= 1; z -= 1;                                  // (every voxel, ghosts included)
  const int v = VOX(x, y, z);
  float jx = 0.f, jy = 0.f, jz = 0.f;
  if (x >= 1 && y >= 1 && z >= 1 && x <= g.nx + 1 && y <= g.ny + 1 && z <= g.nz + 1) {
    const float4 *a0 = A + 3 * (size_t)v, *ax = a0 - 3, *ay = a0 - 3 * (size_t)g.sy, *az = a0 - 3 * (size_t)g.sz;
    const float4 *ayz = ay - 3 * (size_t)g.sz, *azx = az - 3, *axy = ax - 3 * (size_t)g.sy;
    jx += cx * (a0[0].x + ay[0].y + az[0].z + ayz[0].w);
    jy += cy * (a0[1].x + az[1].y + ax[1].z + azx[1].w);
    jz += cz * (a0[2].x + ax[2].y + ay[2].z + axy[2].w);
  }
  f.c[F_JFX][v] = jx; f.c[F_JFY][v] = jy; f.c[F_JFZ][v] = jz;
}

// The same pass TILED INTO LDS with a one-cell halo (round 4).  The kernel above has every thread fetch pieces of seven
// neighbouring 48-byte records (twelve 16-byte loads: four times the bytes it needs, served by L1 / L2: 0.35 of the
// roofline at 256^3).  Here a workgroup owns a column of UT_X x UT_Y voxels and sweeps it along z: each plane's records --
// the tile's and the row y - 1 and column x - 1 before it -- are read ONCE, as consecutive 16-byte vectors of whole rows
// (a row of 65 records is 3120 contiguous bytes), parked in LDS, and every voxel takes its twelve terms from there: eight
// from this plane (its own record, x - 1, y - 1, x - 1 y - 1), four from the plane below, of which only those four floats per
// record are kept (two buffers, by z parity).  Same terms, same order of additions as unload_accumulator.cxx:36-52 after
// clear_jf: same bits.  Traffic: (65 x 9) / (64 x 8) = 1.14 records read per voxel, + 1 / UT_Z for the plane that primes a sweep.
constexpr int UT_X = 64, UT_Y = 8, UT_Z = 32, UT_REC = (UT_X + 1) * (UT_Y + 1), UT_VEC = 3 * (UT_X + 1);
__global__ __launch_bounds__(256)
void clear_unload_tiled_kernel(FieldsK f, const float4 *__restrict__ A, GridK g, float cx, float cy, float cz, int tiles_x, int tiles_y) {
  __shared__ float4 s_cur[UT_REC * 3];                     // this plane: record (rx, ry) -> s_cur[3 * (ry * (UT_X + 1) + rx) + part]
  __shared__ float4 s_low[2][UT_REC];                      // the plane below, by z parity: (jx[2], jx[3], jy[1], jy[3]) of every record
  const unsigned b = blockIdx.x;
  const int tx = (int)(b % (unsigned)tiles_x), ty = (int)((b / (unsigned)tiles_x) % (unsigned)tiles_y), tz = (int)(b / (unsigned)(tiles_x * tiles_y));
  const int x0 = tx * UT_X, y0 = ty * UT_Y, z0 = tz * UT_Z;        // the tile's first voxel (ghosts included: 0 .. n + 1)
  const int z1 = min(z0 + UT_Z, g.nz + 2);
  const int tid = threadIdx.x;
  // records this workgroup parks: x0 - 1 .. x0 + UT_X - 1, y0 - 1 .. y0 + UT_Y - 1 (clipped to the array: what lies outside
  // feeds ghost voxels only, which are written as zeros).  A plane's vectors are fetched into registers one plane AHEAD: the
  // loads of plane z + 1 are in flight while plane z is computed from LDS (fetched, parked and computed one after the other a
  // plane costs an HBM round trip: 449 us at 256^3 against 360 for the per-voxel kernel).
  constexpr int PER = ((UT_Y + 1) * UT_VEC + 255) / 256;
  float4 r[PER];
  auto fetch = [&](const int z) {
#pragma unroll
    for (int m = 0; m < PER; m++) {
      const int j = tid + 256 * m;
      const int ry = j / UT_VEC, c = j - ry * UT_VEC, rx = c / 3;
      const int x = x0 - 1 + rx, y = y0 - 1 + ry;
      r[m] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < (UT_Y + 1) * UT_VEC && z >= 0 && x >= 0 && y >= 0 && x <= g.nx + 1 && y <= g.ny + 1) r[m] = A[3 * (size_t)VOX(x, y, z) + (c - 3 * rx)];
    }
  };
  fetch(z0 - 1);
  for (int z = z0 - 1; z < z1; z++) {
    const bool prime = z < z0;                              // the plane below the sweep's first: only its four floats are needed
#pragma unroll
    for (int m = 0; m < PER; m++) {
      const int j = tid + 256 * m;
      if (j >= (UT_Y + 1) * UT_VEC) continue;
      const int ry = j / UT_VEC, c = j - ry * UT_VEC, rx = c / 3, part = c - 3 * rx, rec = ry * (UT_X + 1) + rx;
      if (!prime) s_cur[3 * rec + part] = r[m];
      // what the plane above will want of this record: jx[2], jx[3] (part 0: z, w) and jy[1], jy[3] (part 1: y, w)
      float *low = reinterpret_cast<float *>(&s_low[(z + 1) & 1][rec]);
      if (part == 0) { low[0] = r[m].z; low[1] = r[m].w; }
      else if (part == 1) { low[2] = r[m].y; low[3] = r[m].w; }
    }
    __syncthreads();
    if (z + 1 < z1) fetch(z + 1);
    if (!prime) {
      const float4 *low = s_low[z & 1];
#pragma unroll
      for (int k = 0; k < UT_X * UT_Y / 256; k++) {
        const int t = tid + 256 * k, lx = t % UT_X, ly = t / UT_X;
        const int x = x0 + lx, y = y0 + ly;
        if (x > g.nx + 1 || y > g.ny + 1) continue;
        const int r0 = (ly + 1) * (UT_X + 1) + (lx + 1), rx = r0 - 1, ry = r0 - (UT_X + 1), rxy = ry - 1;
        float jx = 0.f, jy = 0.f, jz = 0.f;
        if (x >= 1 && y >= 1 && z >= 1 && z <= g.nz + 1) {
          const float4 a00 = s_cur[3 * r0], a01 = s_cur[3 * r0 + 1], a02 = s_cur[3 * r0 + 2];
          const float4 ax1 = s_cur[3 * rx + 1], ax2 = s_cur[3 * rx + 2];
          const float4 ay0 = s_cur[3 * ry], ay2 = s_cur[3 * ry + 2];
          const float4 axy2 = s_cur[3 * rxy + 2];
          const float4 az = low[r0], ayz = low[ry], azx = low[rx];     // (jx[2], jx[3], jy[1], jy[3]) of z - 1
          jx += cx * (a00.x + ay0.y + az.x + ayz.y);
          jy += cy * (a01.x + az.z + ax1.z + azx.w);
          jz += cz * (a02.x + ax2.y + ay2.z + axy2.w);
        }
        const int v = VOX(x, y, z);
        f.c[F_JFX][v] = jx; f.c[F_JFY][v] = jy; f.c[F_JFZ][v] = jz;
      }
    }
    __syncthreads();                                        // (the next plane overwrites s_cur and the other s_low)
  }
}

int k_clear_jf_unload_accumulator(Engine *e) {
  if (acc_finalize(e)) return 1;
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  const float cx = (float)(0.25 * G.rdy * G.rdz / G.dt);
  const float cy = (float)(0.25 * G.rdz * G.rdx / G.dt);
  const float cz = (float)(0.25 * G.rdx * G.rdy / G.dt);
  const int tiles_x = (g.nx + 2 + UT_X - 1) / UT_X, tiles_y = (g.ny + 2 + UT_Y - 1) / UT_Y, tiles_z = (g.nz + 2 + UT_Z - 1) / UT_Z;
  // (the tiles pay from about a thousand workgroups on -- four per CU, each a column of 32 planes: 273 against 360 us at 256^3;
  // at 128^3 they are 255 workgroups that wait for their planes one after the other: 78 us against 45 for the per-voxel kernel)
  if (e->knobs.unload_tiled == 2 || (e->knobs.unload_tiled == 1 && tiles_x * tiles_y * tiles_z >= 1024)) {
    hipLaunchKernelGGL(clear_unload_tiled_kernel, dim3((unsigned)(tiles_x * tiles_y * tiles_z)), dim3(256), 0, e->stream, e->f,
                       reinterpret_cast<const float4 *>(e->acc), g, cx, cy, cz, tiles_x, tiles_y);
    VH_CHECK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(clear_unload_kernel, dim3(banded_grid(Box3{g.nx + 2, g.ny + 2, g.nz + 2})), dim3(256), 0, e->stream, e->f,
                     reinterpret_cast<const float4 *>(e->acc), g, cx, cy, cz);
  VH_CHECK(hipGetLastError());
  return 0;
}

int k_clear_jf(Engine *e) {
  // jfx,jfy,jfz are adjacent component arrays of one block: one memset (sfa.c:188-211)
  VH_CHECK(hipMemsetAsync(e->f.c[F_JFX], 0, sizeof(float) * 3 * (size_t)e->gk.nv, e->stream));
  return 0;
}

// ---- advance_b: advance_b.c:12-14 over 1..n+1 with per-component predicates ----------------------
__global__ __launch_bounds__(256)
void advance_b_kernel(FieldsK f, GridK g, float px, float py, float pz) {
  int x, y, z;
  if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z), vx = v + 1, vy = v + g.sy, vz = v + g.sz;
  const float ex = f.c[F_EX][v], ey = f.c[F_EY][v], ez = f.c[F_EZ][v];
  if (y <= g.ny && z <= g.nz) f.c[F_CBX][v] -= (py * (f.c[F_EZ][vy] - ez) - pz * (f.c[F_EY][vz] - ey));
  if (z <= g.nz && x <= g.nx) f.c[F_CBY][v] -= (pz * (f.c[F_EX][vz] - ex) - px * (f.c[F_EZ][vx] - ez));
  if (x <= g.nx && y <= g.ny) f.c[F_CBZ][v] -= (px * (f.c[F_EY][vx] - ey) - py * (f.c[F_EX][vy] - ex));
}

// The same stencil TILED INTO LDS with a one-cell halo (round 4; the north star's wording).  The kernel above has every thread
// read nine E values, six of them its neighbours' (served by L1 / L2: 0.53 of the roofline at 256^3).  Here a workgroup owns a
// column of FT_X x FT_Y voxels and sweeps it along z: a plane of E -- the tile's voxels and the row y + 1 and column x + 1 behind
// them -- is read ONCE, whole rows of 65 consecutive floats, into a ring of two planes in LDS (this plane with its halo, the
// plane above for the z + 1 terms), one plane AHEAD of the arithmetic through registers (see clear_unload_tiled_kernel); cB goes
// straight from memory to its thread and back.  Same terms in the same order: same bits.  Traffic per voxel: (65 x 9) / (64 x 8)
// = 1.14 x 12 bytes of E (+ 1 / FT_Z for the plane that primes a sweep), 12 + 12 of cB.
// MEASURED (rocprofv3, 256^3, profiles/r04_field_tiles_kernel_stats.txt): 297 us against 147 for the kernel above, advance_e
// 465 against 246 (128^3: 154 against 21, 226 against 37).  Component arrays of 4-byte values are not accumulator records: a
// wavefront's x + 1 neighbours are the same lines one lane on, y + 1 and z + 1 are whole lines that another wavefront of the
// chip has just read -- L1 / L2 serve them at no HBM cost, and a thread per voxel keeps tens of thousands of independent loads in
// flight where a workgroup that sweeps a column has one plane (7 KB).  The tiles paid for clear_jf + unload (a voxel wants four
// floats out of each of seven 48-byte records: 4 x the bytes through L1).  They stay as an opt-in (VPIC_HIP_FIELD_TILES=2) with
// their parity tests; the default is one thread per voxel.
constexpr int FT_X = 64, FT_Y = 8, FT_Z = 32, FT_W = FT_X + 1, FT_H = FT_Y + 1, FT_N = FT_W * FT_H, FT_PER = (3 * FT_N + 255) / 256;
// the three components C0 .. C0 + 2 of plane z, tile origin (x0, y0), with the halo row / column on the high (HI) or low side:
// into registers (fetch), from registers into one plane of the ring (park)
template <int C0, bool HI>
__device__ __forceinline__ void ft_fetch(float (&r)[FT_PER], const FieldsK &f, const GridK &g, int x0, int y0, int z, int tid) {
#pragma unroll
  for (int m = 0; m < FT_PER; m++) {
    const int j = tid + 256 * m, comp = j / FT_N, rem = j - comp * FT_N, ry = rem / FT_W, rx = rem - ry * FT_W;
    const int x = x0 + rx - (HI ? 0 : 1), y = y0 + ry - (HI ? 0 : 1);
    r[m] = 0.f;
    if (j < 3 * FT_N && z >= 0 && z <= g.nz + 1 && x <= g.nx + 1 && y <= g.ny + 1) r[m] = f.c[C0 + comp][VOX(x, y, z)];   // (x, y >= 0: the tiles begin at voxel 1)
  }
}
__device__ __forceinline__ void ft_park(float *plane, const float (&r)[FT_PER], int tid) {
#pragma unroll
  for (int m = 0; m < FT_PER; m++) { const int j = tid + 256 * m; if (j < 3 * FT_N) plane[j] = r[m]; }
}

__global__ __launch_bounds__(256)
void advance_b_tiled_kernel(FieldsK f, GridK g, float px, float py, float pz, int tiles_x, int tiles_y) {
  __shared__ float s_e[2][3 * FT_N];                       // ring of two planes: component c of (rx, ry) at [c * FT_N + ry * FT_W + rx]
  const unsigned b = blockIdx.x;
  const int tx = (int)(b % (unsigned)tiles_x), ty = (int)((b / (unsigned)tiles_x) % (unsigned)tiles_y), tz = (int)(b / (unsigned)(tiles_x * tiles_y));
  const int x0 = 1 + tx * FT_X, y0 = 1 + ty * FT_Y, z0 = 1 + tz * FT_Z;      // the box 1 .. n + 1 (advance_b.c:74-156)
  const int z1 = min(z0 + FT_Z, g.nz + 2);
  const int tid = threadIdx.x;
  float r[FT_PER];
  ft_fetch<F_EX, true>(r, f, g, x0, y0, z0, tid);
  ft_park(s_e[z0 & 1], r, tid);
  ft_fetch<F_EX, true>(r, f, g, x0, y0, z0 + 1, tid);
  for (int z = z0; z < z1; z++) {
    ft_park(s_e[(z + 1) & 1], r, tid);                     // the plane above (zeros beyond the array: only ever multiplied into terms that are not stored)
    __syncthreads();
    if (z + 1 < z1) ft_fetch<F_EX, true>(r, f, g, x0, y0, z + 2, tid);
    const float *e0 = s_e[z & 1], *e1 = s_e[(z + 1) & 1];
#pragma unroll
    for (int k = 0; k < FT_X * FT_Y / 256; k++) {
      const int t = tid + 256 * k, lx = t % FT_X, ly = t / FT_X;
      const int x = x0 + lx, y = y0 + ly;
      if (x > g.nx + 1 || y > g.ny + 1) continue;
      const int v = VOX(x, y, z), o = ly * FT_W + lx;
      const float ex = e0[o], ey = e0[FT_N + o], ez = e0[2 * FT_N + o];
      if (y <= g.ny && z <= g.nz) f.c[F_CBX][v] -= (py * (e0[2 * FT_N + o + FT_W] - ez) - pz * (e1[FT_N + o] - ey));
      if (z <= g.nz && x <= g.nx) f.c[F_CBY][v] -= (pz * (e1[o] - ex) - px * (e0[2 * FT_N + o + 1] - ez));
      if (x <= g.nx && y <= g.ny) f.c[F_CBZ][v] -= (px * (e0[FT_N + o + 1] - ey) - py * (e0[o + FT_W] - ex));
    }
    __syncthreads();                                        // (the next plane but one overwrites e0)
  }
}

// ---- plane operations (local boundary conditions and face messages) -----------------------------
// A component directed along axis ca lives, on a plane normal to `axis`:
//   edge mesh (E, tca, jf): 1..n along ca, 1..n+1 along the other axes
//   face mesh (cB)        : 1..n+1 along ca, 1..n along the other axes
// (field_advance.h:60-70; the *_EDGE_LOOP/*_FACE_LOOP macros of local.c:26-46, remote.c:17-41.)
struct PlaneBox { int lo[3], n[3]; int count; };
static PlaneBox plane_box(const GridK &g, int axis, int plane, int ca, int edge_mesh) {
  const int n[3] = {g.nx, g.ny, g.nz};
  PlaneBox b;
  for (int d = 0; d < 3; d++) {
    b.lo[d] = 1;
    b.n[d] = (d == ca) ? (edge_mesh ? n[d] : n[d] + 1) : (edge_mesh ? n[d] + 1 : n[d]);
  }
  b.lo[axis] = plane; b.n[axis] = 1;
  b.count = b.n[0] * b.n[1] * b.n[2];
  return b;
}
__device__ __forceinline__ int plane_voxel(const PlaneBox &b, const GridK &g, int t) {
  const int per_z = b.n[0] * b.n[1];
  const int zz = t / per_z, r = t - zz * per_z, yy = r / b.n[0], xx = r - yy * b.n[0];
  return VOX(b.lo[0] + xx, b.lo[1] + yy, b.lo[2] + zz);
}

enum { OP_PACK = 0, OP_UNPACK_TANG_B, OP_UNPACK_JF, OP_COPY_FROM, OP_NEG_FROM, OP_ZERO, OP_SCALE2 };
// Two component arrays (the two tangential components of a face), each over its own box; the
// message holds box 1 then box 2, each z-outer/x-inner as the reference loops run.
struct PlaneArgs { float *c1, *c2, *d1, *d2; PlaneBox b1, b2; int op, off; };

__global__ void plane_kernel(PlaneArgs A, GridK g, float *buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.b1.count + A.b2.count) return;
  const bool second = t >= A.b1.count;
  const int v = second ? plane_voxel(A.b2, g, t - A.b1.count) : plane_voxel(A.b1, g, t);
  float *c = second ? A.c2 : A.c1;
  float *d = second ? A.d2 : A.d1;
  switch (A.op) {
    case OP_PACK: buf[t] = c[v]; break;
    case OP_UNPACK_TANG_B: {  // remote.c:108-115 on a uniform mesh: rw = 1, lw = 0
      const float rw = 1.f, lw = 0.f;
      c[v] = rw * buf[t] + lw * c[v + A.off];
    } break;
    case OP_UNPACK_JF: {      // remote.c:452-466 on a uniform mesh: lw = rw = 1
      const float rw = 1.f, lw = 1.f;
      c[v] = lw * c[v] + rw * buf[t];
    } break;
    case OP_COPY_FROM: c[v] = c[v + A.off]; break;                  // local.c:76-79 (PEC ghost B)
    case OP_NEG_FROM: c[v] = -c[v + A.off]; break;                  // local.c:80-83
    case OP_ZERO: c[v] = 0.f; if (d) d[v] = 0.f; break;             // local.c:237-246, 344-347
    case OP_SCALE2: c[v] *= 2.f; break;                             // local.c:348-351
  }
}

static int launch_plane(Engine *e, const PlaneArgs &A, float *buf) {
  const int n = A.b1.count + A.b2.count;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(plane_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, A, e->gk, buf);
  VH_CHECK(hipGetLastError());
  return 0;
}

static const int N_OF[3] = {0, 1, 2};
static int n_axis(const GridK &g, int a) { return a == 0 ? g.nx : a == 1 ? g.ny : g.nz; }
static int stride_axis(const GridK &g, int a) { return a == 0 ? 1 : a == 1 ? g.sy : g.sz; }

// tangential pair of a face: components (axis+1)%3 then (axis+2)%3 of the base component
static PlaneArgs tang_pair(Engine *e, int base, int axis, int plane, int edge_mesh, int op, int off, int base2 = -1) {
  (void)N_OF;
  const GridK &g = e->gk;
  const int ca1 = (axis + 1) % 3, ca2 = (axis + 2) % 3;
  PlaneArgs A;
  A.c1 = e->f.c[base + ca1]; A.c2 = e->f.c[base + ca2];
  A.d1 = base2 >= 0 ? e->f.c[base2 + ca1] : nullptr;
  A.d2 = base2 >= 0 ? e->f.c[base2 + ca2] : nullptr;
  A.b1 = plane_box(g, axis, plane, ca1, edge_mesh);
  A.b2 = plane_box(g, axis, plane, ca2, edge_mesh);
  A.op = op; A.off = off;
  return A;
}

int k_face_count(const Engine *e, int dir) {
  const GridK &g = e->gk;
  const int a = dir % 3, nY = n_axis(g, (a + 1) % 3), nZ = n_axis(g, (a + 2) % 3);
  return nY * (nZ + 1) + nZ * (nY + 1);       // remote.c:69 without the leading cell-size float
}

// what: 0 tang_b (remote.c:83-85: plane 1 / n), 1 jf (remote.c:442-444: plane 1 / n+1)
int k_pack_face(Engine *e, int dir, float *buf, int what) {
  const GridK &g = e->gk;
  const int axis = dir % 3, n = n_axis(g, axis);
  if (what == 0) return launch_plane(e, tang_pair(e, F_CBX, axis, dir < 3 ? 1 : n, 0, OP_PACK, 0), buf);
  return launch_plane(e, tang_pair(e, F_JFX, axis, dir < 3 ? 1 : n + 1, 1, OP_PACK, 0), buf);
}
// tang_b lands on the ghost plane (remote.c:111), jf on the shared plane (remote.c:458)
int k_unpack_face(Engine *e, int dir, const float *buf, int what) {
  const GridK &g = e->gk;
  const int axis = dir % 3, n = n_axis(g, axis), st = stride_axis(g, axis);
  if (what == 0)
    return launch_plane(e, tang_pair(e, F_CBX, axis, dir < 3 ? n + 1 : 0, 0, OP_UNPACK_TANG_B, dir < 3 ? -st : st),
                        const_cast<float *>(buf));
  return launch_plane(e, tang_pair(e, F_JFX, axis, dir < 3 ? n + 1 : 1, 1, OP_UNPACK_JF, 0), const_cast<float *>(buf));
}

// Absorbing face (local.c:84-108): 2nd-order accurate 1st-order Higdon condition.  The ghost cB of
// tangential component `ca` relaxes towards the first interior value, with the two E differences of
// Faraday's law at the face added.  One launch per (face, component).
struct AbsorbArgs {
  float *cb;            // cB component being set (ca)
  const float *e_t;     // E component along the OTHER tangential axis (differenced across the face)
  const float *e_n;     // E component along the face normal (differenced along that other axis)
  PlaneBox b;
  int off;              // ghost -> first interior voxel
  int to_face;          // ghost -> face voxel (1 or n+1 along the normal)
  int st_other;         // stride of the other tangential axis
  float cdt_n, cdt_other, decay, drive, sign_t1, sign_t2, flip;
};
__global__ void absorb_tang_b_kernel(AbsorbArgs A, GridK g) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.b.count) return;
  const int v = plane_voxel(A.b, g, t), vf = v + A.to_face;
  float t1 = A.cdt_n * (A.e_t[vf + A.off] - A.e_t[vf]);
  t1 = A.flip * t1;
  float t2 = A.e_n[v + A.off + A.st_other];
  t2 = A.cdt_other * (t2 - A.e_n[v + A.off]);
  // cbY = decay*cbY + drive*cbY(in) - t1 + t2 ;  cbZ = decay*cbZ + drive*cbZ(in) + t1 - t2
  if (A.sign_t1 < 0) A.cb[v] = A.decay * A.cb[v] + A.drive * A.cb[v + A.off] - t1 + t2;
  else               A.cb[v] = A.decay * A.cb[v] + A.drive * A.cb[v + A.off] + t1 - t2;
}
static int absorb_tang_b(Engine *e, int face) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis), st = stride_axis(g, axis);
  const int ghost = hi ? n + 1 : 0, aY = (axis + 1) % 3, aZ = (axis + 2) % 3;
  const float cdt[3] = {G.cvac * G.dt * G.rdx, G.cvac * G.dt * G.rdy, G.cvac * G.dt * G.rdz};
  const float higend = (g.nx > 1 || g.ny > 1 || g.nz > 1) ? 1.03527618 : 1.;
  float drive = cdt[axis] * higend;
  const float decay = (1 - drive) / (1 + drive);
  drive = 2 * drive / (1 + drive);
  for (int k = 0; k < 2; k++) {
    const int ca = k == 0 ? aY : aZ, other = k == 0 ? aZ : aY;
    AbsorbArgs A;
    A.cb = e->f.c[F_CBX + ca]; A.e_t = e->f.c[F_EX + other]; A.e_n = e->f.c[F_EX + axis];
    A.b = plane_box(g, axis, ghost, ca, 0);
    A.off = hi ? -st : st; A.to_face = ((hi ? n + 1 : 1) - ghost) * st; A.st_other = stride_axis(g, other);
    A.cdt_n = cdt[axis]; A.cdt_other = cdt[other]; A.decay = decay; A.drive = drive;
    A.sign_t1 = k == 0 ? -1.f : 1.f; A.sign_t2 = -A.sign_t1; A.flip = hi ? -1.f : 1.f;
    if (A.b.count <= 0) continue;
    hipLaunchKernelGGL(absorb_tang_b_kernel, dim3((A.b.count + 255) / 256), dim3(256), 0, e->stream, A, g);
    VH_CHECK(hipGetLastError());
  }
  return 0;
}

// ---- faces a domain shares with itself, fused ----------------------------------------------------
// The pack / unpack pairs above are what a domain does per face with a neighbour.  When the
// neighbour is the domain itself the message never leaves the GPU, and all copies of one phase are
// independent (they read interior planes and write disjoint ghost planes), so one launch does them
// all: 12 launches -> 1 for the tangential-B ghosts of advance_e / compute_curl_b.
struct SelfCopy { float *c; PlaneBox b; int src_off, in_off; };
struct SelfCopyTable { SelfCopy job[12]; int first[13]; int n; };
__global__ void self_ghost_kernel(SelfCopyTable T, GridK g) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T.first[T.n]) return;
  int k = 0;
  while (t >= T.first[k + 1]) k++;
  const SelfCopy &J = T.job[k];
  const int v = plane_voxel(J.b, g, t - T.first[k]);
  const float rw = 1.f, lw = 0.f;                        // remote.c:108-115 on a uniform mesh
  J.c[v] = rw * J.c[v + J.src_off] + lw * J.c[v + J.in_off];
}
static int self_ghost_tang_b(Engine *e) {
  const GridK &g = e->gk;
  SelfCopyTable T;
  T.n = 0; T.first[0] = 0;
  for (int dir = 0; dir < 6; dir++) {
    if (g.fbc[dir] != g.rank) continue;
    const int axis = dir % 3, n = n_axis(g, axis), st = stride_axis(g, axis);
    const int from = dir < 3 ? 1 : n, to = dir < 3 ? n + 1 : 0;
    for (int k = 1; k <= 2; k++) {
      const int ca = (axis + k) % 3;
      SelfCopy &J = T.job[T.n];
      J.c = e->f.c[F_CBX + ca]; J.b = plane_box(g, axis, to, ca, 0);
      J.src_off = (from - to) * st; J.in_off = dir < 3 ? -st : st;
      T.first[T.n + 1] = T.first[T.n] + J.b.count;
      T.n++;
    }
  }
  const int total = T.first[T.n];
  if (total <= 0) return 0;
  hipLaunchKernelGGL(self_ghost_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, T, g);
  VH_CHECK(hipGetLastError());
  return 0;
}
// synchronize_jf for one axis shared with the domain itself (remote.c:452-466 with lw = rw = 1):
// both shared planes end with old(1) + old(n+1); 4 launches -> 1
__global__ void self_sum_planes_kernel(float *c1, float *c2, PlaneBox b1, PlaneBox b2, GridK g, int span) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b1.count + b2.count) return;
  const bool second = t >= b1.count;
  float *c = second ? c2 : c1;
  const int v = second ? plane_voxel(b2, g, t - b1.count) : plane_voxel(b1, g, t);
  const float lw = 1.f, rw = 1.f, lo = c[v], hi = c[v + span];
  c[v + span] = lw * hi + rw * lo;
  c[v] = lw * lo + rw * hi;
}

static int local_ghost_tang_b(Engine *e) {        // local.c:50-122
  const GridK &g = e->gk;
  for (int face = 0; face < 6; face++) {
    const int bc = g.fbc[face];
    if (bc >= 0) continue;
    const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis), st = stride_axis(g, axis);
    int op;
    if (bc == VPIC_ABSORB_FIELDS) { if (absorb_tang_b(e, face)) return 1; continue; }
    if (bc == VPIC_PEC_FIELDS) op = OP_COPY_FROM;
    else if (bc == VPIC_SYMMETRIC_FIELDS || bc == VPIC_PMC_FIELDS) op = OP_NEG_FROM;
    else VH_FAIL("Bad boundary condition encountered.");
    if (launch_plane(e, tang_pair(e, F_CBX, axis, hi ? n + 1 : 0, 0, op, hi ? -st : st), nullptr)) return 1;
  }
  return 0;
}
static int local_adjust_tang_e(Engine *e) {       // local.c:224-264
  const GridK &g = e->gk;
  for (int face = 0; face < 6; face++) {
    if (g.fbc[face] != VPIC_PEC_FIELDS) continue;
    const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis);
    if (launch_plane(e, tang_pair(e, F_EX, axis, hi ? n + 1 : 1, 1, OP_ZERO, 0, F_TCAX), nullptr)) return 1;
  }
  return 0;
}
static int local_adjust_norm_b(Engine *e) {       // local.c:266-296
  const GridK &g = e->gk;
  for (int face = 0; face < 6; face++) {
    if (g.fbc[face] != VPIC_SYMMETRIC_FIELDS) continue;
    const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis);
    PlaneArgs A{};
    A.c1 = e->f.c[F_CBX + axis]; A.c2 = nullptr; A.d1 = A.d2 = nullptr;
    A.b1 = plane_box(g, axis, hi ? n + 1 : 1, axis, 0);
    A.b2 = A.b1; A.b2.count = 0;
    A.op = OP_ZERO; A.off = 0;
    if (launch_plane(e, A, nullptr)) return 1;
  }
  return 0;
}
static int local_adjust_jf(Engine *e) {           // local.c:335-368
  const GridK &g = e->gk;
  for (int face = 0; face < 6; face++) {
    const int bc = g.fbc[face];
    if (bc >= 0) continue;
    const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis);
    const int op = (bc == VPIC_PEC_FIELDS) ? OP_ZERO : OP_SCALE2;
    if (launch_plane(e, tang_pair(e, F_JFX, axis, hi ? n + 1 : 1, 1, op, 0), nullptr)) return 1;
  }
  return 0;
}

// synchronize_jf (remote.c:416-506) for the faces this domain shares with itself: per axis both
// planes are packed before either is accumulated into, exactly as both sends are posted before
// either receive is unpacked.
int k_local_adjust_jf(Engine *e) { return local_adjust_jf(e); }

int k_synchronize_jf_self(Engine *e, int axis) {
  const GridK &g = e->gk;
  if (g.fbc[axis] != g.rank || g.fbc[axis + 3] != g.rank) return 0;
  const int ca1 = (axis + 1) % 3, ca2 = (axis + 2) % 3;
  const PlaneBox b1 = plane_box(g, axis, 1, ca1, 1), b2 = plane_box(g, axis, 1, ca2, 1);
  const int total = b1.count + b2.count;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(self_sum_planes_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream,
                     e->f.c[F_JFX + ca1], e->f.c[F_JFX + ca2], b1, b2, g, n_axis(g, axis) * stride_axis(g, axis));
  VH_CHECK(hipGetLastError());
  return 0;
}

int k_synchronize_jf_local(Engine *e) {
  if (local_adjust_jf(e)) return 1;
  for (int axis = 0; axis < 3; axis++)
    if (k_synchronize_jf_self(e, axis)) return 1;
  return 0;
}

int k_advance_b(Engine *e, float frac) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  const float px = (g.nx > 1) ? frac * G.cvac * G.dt * G.rdx : 0;
  const float py = (g.ny > 1) ? frac * G.cvac * G.dt * G.rdy : 0;
  const float pz = (g.nz > 1) ? frac * G.cvac * G.dt * G.rdz : 0;
  const unsigned n = (unsigned)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
  const int tiles_x = (g.nx + FT_X) / FT_X, tiles_y = (g.ny + FT_Y) / FT_Y, tiles_z = (g.nz + FT_Z) / FT_Z;      // (n + 1 voxels per axis)
  // (LDS tiles where they fill the chip -- a column of 32 planes per workgroup, four workgroups per CU -- like clear_unload)
  if (e->knobs.field_tiles == 2 || (e->knobs.field_tiles == 1 && tiles_x * tiles_y * tiles_z >= 1024))
    hipLaunchKernelGGL(advance_b_tiled_kernel, dim3((unsigned)(tiles_x * tiles_y * tiles_z)), dim3(256), 0, e->stream, e->f, g, px, py, pz, tiles_x, tiles_y);
  else
  hipLaunchKernelGGL(advance_b_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, g, px, py, pz);
  VH_CHECK(hipGetLastError());
  return local_adjust_norm_b(e);
}

// ---- advance_e: advance_e.c:8-25 over 1..n+1 with per-component predicates -----------------------
// part: 0 the whole box 1..n+1; 1 the planes x = 2..nx (need no x ghost of another domain); 2 the planes x = 1 and
// x = nx+1 (advance_e.c:155-327 makes the same split: interior first, the rest once the remote ghosts are in)
struct AdvanceEParams { float px, py, pz, damp, cj; int part; };

template <bool SINGLE_MATERIAL>
__global__ __launch_bounds__(256)
void advance_e_kernel(FieldsK f, const vpic_material_coefficient_t *__restrict__ m, GridK g, AdvanceEParams P) {
  int x, y, z;
  if (P.part == 0) { if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return; }
  else if (P.part == 1) { if (!decode(Box3{g.nx - 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return; x += 1; }
  else { if (!decode(Box3{2, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return; x = (x == 1) ? 1 : g.nx + 1; }
  const int v = VOX(x, y, z), vx = v - 1, vy = v - g.sy, vz = v - g.sz;
  const float px = P.px, py = P.py, pz = P.pz, damp = P.damp, cj = P.cj;
  const float cbx = f.c[F_CBX][v], cby = f.c[F_CBY][v], cbz = f.c[F_CBZ][v];
#define MAT(which, vox) (SINGLE_MATERIAL ? m[0] : m[f.m[which][vox]])
  if (x <= g.nx) {
    const float t = (py * (cbz * MAT(M_FMATZ, v).rmuz - f.c[F_CBZ][vy] * MAT(M_FMATZ, vy).rmuz) -
                     pz * (cby * MAT(M_FMATY, v).rmuy - f.c[F_CBY][vz] * MAT(M_FMATY, vz).rmuy)) - damp * f.c[F_TCAX][v];
    f.c[F_TCAX][v] = t;
    f.c[F_EX][v] = MAT(M_EMATX, v).decayx * f.c[F_EX][v] + MAT(M_EMATX, v).drivex * (t - cj * f.c[F_JFX][v]);
  }
  if (y <= g.ny) {
    const float t = (pz * (cbx * MAT(M_FMATX, v).rmux - f.c[F_CBX][vz] * MAT(M_FMATX, vz).rmux) -
                     px * (cbz * MAT(M_FMATZ, v).rmuz - f.c[F_CBZ][vx] * MAT(M_FMATZ, vx).rmuz)) - damp * f.c[F_TCAY][v];
    f.c[F_TCAY][v] = t;
    f.c[F_EY][v] = MAT(M_EMATY, v).decayy * f.c[F_EY][v] + MAT(M_EMATY, v).drivey * (t - cj * f.c[F_JFY][v]);
  }
  if (z <= g.nz) {
    const float t = (px * (cby * MAT(M_FMATY, v).rmuy - f.c[F_CBY][vx] * MAT(M_FMATY, vx).rmuy) -
                     py * (cbx * MAT(M_FMATX, v).rmux - f.c[F_CBX][vy] * MAT(M_FMATX, vy).rmux)) - damp * f.c[F_TCAZ][v];
    f.c[F_TCAZ][v] = t;
    f.c[F_EZ][v] = MAT(M_EMATZ, v).decayz * f.c[F_EZ][v] + MAT(M_EMATZ, v).drivez * (t - cj * f.c[F_JFZ][v]);
  }
#undef MAT
}

// advance_e TILED INTO LDS (round 4; one material, the whole box or its planes x = 2 .. nx): cB of a plane -- the tile's voxels
// and the row y - 1 and column x - 1 before them -- read once into a ring of two planes (this plane with its halo, the plane
// below for the z - 1 terms), one plane ahead of the arithmetic; tca, jf and E go straight from memory to their thread.  The
// terms and their order are advance_e_kernel<true>'s: same bits.
__global__ __launch_bounds__(256)
void advance_e_tiled_kernel(FieldsK f, const vpic_material_coefficient_t *__restrict__ m, GridK g, AdvanceEParams P, int x_lo, int x_hi, int tiles_x, int tiles_y) {
  __shared__ float s_b[2][3 * FT_N];                       // component c of (rx, ry) at [c * FT_N + ry * FT_W + rx]; (rx, ry) = voxel (x0 - 1 + rx, y0 - 1 + ry)
  const unsigned b = blockIdx.x;
  const int tx = (int)(b % (unsigned)tiles_x), ty = (int)((b / (unsigned)tiles_x) % (unsigned)tiles_y), tz = (int)(b / (unsigned)(tiles_x * tiles_y));
  const int x0 = x_lo + tx * FT_X, y0 = 1 + ty * FT_Y, z0 = 1 + tz * FT_Z;
  const int z1 = min(z0 + FT_Z, g.nz + 2);
  const int tid = threadIdx.x;
  const float px = P.px, py = P.py, pz = P.pz, damp = P.damp, cj = P.cj;
  const vpic_material_coefficient_t mc = m[0];
  float r[FT_PER];
  ft_fetch<F_CBX, false>(r, f, g, x0, y0, z0 - 1, tid);
  ft_park(s_b[(z0 - 1) & 1], r, tid);
  ft_fetch<F_CBX, false>(r, f, g, x0, y0, z0, tid);
  for (int z = z0; z < z1; z++) {
    ft_park(s_b[z & 1], r, tid);
    __syncthreads();
    if (z + 1 < z1) ft_fetch<F_CBX, false>(r, f, g, x0, y0, z + 1, tid);
    const float *b0 = s_b[z & 1], *bl = s_b[(z - 1) & 1];
#pragma unroll
    for (int k = 0; k < FT_X * FT_Y / 256; k++) {
      const int t = tid + 256 * k, lx = t % FT_X, ly = t / FT_X;
      const int x = x0 + lx, y = y0 + ly;
      if (x > x_hi || y > g.ny + 1) continue;
      const int v = VOX(x, y, z), o = (ly + 1) * FT_W + (lx + 1);
      const float cbx = b0[o], cby = b0[FT_N + o], cbz = b0[2 * FT_N + o];
      if (x <= g.nx) {
        const float t_ = (py * (cbz * mc.rmuz - b0[2 * FT_N + o - FT_W] * mc.rmuz) - pz * (cby * mc.rmuy - bl[FT_N + o] * mc.rmuy)) - damp * f.c[F_TCAX][v];
        f.c[F_TCAX][v] = t_;
        f.c[F_EX][v] = mc.decayx * f.c[F_EX][v] + mc.drivex * (t_ - cj * f.c[F_JFX][v]);
      }
      if (y <= g.ny) {
        const float t_ = (pz * (cbx * mc.rmux - bl[o] * mc.rmux) - px * (cbz * mc.rmuz - b0[2 * FT_N + o - 1] * mc.rmuz)) - damp * f.c[F_TCAY][v];
        f.c[F_TCAY][v] = t_;
        f.c[F_EY][v] = mc.decayy * f.c[F_EY][v] + mc.drivey * (t_ - cj * f.c[F_JFY][v]);
      }
      if (z <= g.nz) {
        const float t_ = (px * (cby * mc.rmuy - b0[FT_N + o - 1] * mc.rmuy) - py * (cbx * mc.rmux - b0[o - FT_W] * mc.rmux)) - damp * f.c[F_TCAZ][v];
        f.c[F_TCAZ][v] = t_;
        f.c[F_EZ][v] = mc.decayz * f.c[F_EZ][v] + mc.drivez * (t_ - cj * f.c[F_JFZ][v]);
      }
    }
    __syncthreads();
  }
}

int k_advance_e(Engine *e, int part) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  if (!e->mc) VH_FAIL("advance_e: no material coefficients set");
  AdvanceEParams P;
  P.damp = G.damp;
  P.px = (g.nx > 1) ? (1 + G.damp) * G.cvac * G.dt * G.rdx : 0;
  P.py = (g.ny > 1) ? (1 + G.damp) * G.cvac * G.dt * G.rdy : 0;
  P.pz = (g.nz > 1) ? (1 + G.damp) * G.cvac * G.dt * G.rdz : 0;
  P.cj = G.dt / G.eps0;
  P.part = part;
  // tangential-B ghosts: faces shared with this same domain (the reference sends to itself,
  // grid_comm.c:17-49), then the local boundary conditions (advance_e.c:114-115)
  if (part != 2) {
    if (self_ghost_tang_b(e)) return 1;
    if (local_ghost_tang_b(e)) return 1;
  }
  if (part == 1 && g.nx < 2) return 0;            // a slab one cell thick has no planes 2..nx (its ghosts were still filled above)
  const unsigned n = (unsigned)(part == 0 ? g.nx + 1 : part == 1 ? g.nx - 1 : 2) * (g.ny + 1) * (g.nz + 1);
  const int x_lo = part == 1 ? 2 : 1, x_hi = part == 1 ? g.nx : g.nx + 1;
  const int tiles_x = (x_hi - x_lo + FT_X) / FT_X, tiles_y = (g.ny + FT_Y) / FT_Y, tiles_z = (g.nz + FT_Z) / FT_Z;
  if (part != 2 && !e->f.m[0] && (e->knobs.field_tiles == 2 || (e->knobs.field_tiles == 1 && tiles_x * tiles_y * tiles_z >= 1024)))
    hipLaunchKernelGGL(advance_e_tiled_kernel, dim3((unsigned)(tiles_x * tiles_y * tiles_z)), dim3(256), 0, e->stream, e->f, e->mc, g, P, x_lo, x_hi, tiles_x, tiles_y);
  else
  if (e->f.m[0])
    hipLaunchKernelGGL(advance_e_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, e->mc, g, P);
  else
    hipLaunchKernelGGL(advance_e_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, e->mc, g, P);
  VH_CHECK(hipGetLastError());
  return part == 1 ? 0 : local_adjust_tang_e(e);
}

// ---- energy_f: energy_f.c:50-82 over interior voxels; double partial sums per workgroup ----------
template <bool SINGLE_MATERIAL>
__global__ __launch_bounds__(256)
void energy_f_kernel(FieldsK f, const vpic_material_coefficient_t *__restrict__ m, GridK g, double *__restrict__ partial) {
  __shared__ double s_sum[4][6];
  double en[6] = {0, 0, 0, 0, 0, 0};
  const unsigned total = (unsigned)g.nx * g.ny * g.nz;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
    int x, y, z;
    decode(Box3{g.nx, g.ny, g.nz}, t, x, y, z);
    const int v = VOX(x, y, z), vx = v + 1, vy = v + g.sy, vz = v + g.sz;
    const int vyz = vy + g.sz, vzx = vz + 1, vxy = vx + g.sy;
#define MAT(which, vox) (SINGLE_MATERIAL ? m[0] : m[f.m[which][vox]])
#define SQ(c, vox) (f.c[c][vox] * f.c[c][vox])
    en[0] += 0.25 * (MAT(M_EMATX, v).epsx * f.c[F_EX][v] * f.c[F_EX][v] + MAT(M_EMATX, vy).epsx * f.c[F_EX][vy] * f.c[F_EX][vy] +
                     MAT(M_EMATX, vz).epsx * f.c[F_EX][vz] * f.c[F_EX][vz] + MAT(M_EMATX, vyz).epsx * f.c[F_EX][vyz] * f.c[F_EX][vyz]);
    en[1] += 0.25 * (MAT(M_EMATY, v).epsy * f.c[F_EY][v] * f.c[F_EY][v] + MAT(M_EMATY, vz).epsy * f.c[F_EY][vz] * f.c[F_EY][vz] +
                     MAT(M_EMATY, vx).epsy * f.c[F_EY][vx] * f.c[F_EY][vx] + MAT(M_EMATY, vzx).epsy * f.c[F_EY][vzx] * f.c[F_EY][vzx]);
    en[2] += 0.25 * (MAT(M_EMATZ, v).epsz * f.c[F_EZ][v] * f.c[F_EZ][v] + MAT(M_EMATZ, vx).epsz * f.c[F_EZ][vx] * f.c[F_EZ][vx] +
                     MAT(M_EMATZ, vy).epsz * f.c[F_EZ][vy] * f.c[F_EZ][vy] + MAT(M_EMATZ, vxy).epsz * f.c[F_EZ][vxy] * f.c[F_EZ][vxy]);
    en[3] += 0.5 * (MAT(M_FMATX, v).rmux * f.c[F_CBX][v] * f.c[F_CBX][v] + MAT(M_FMATX, vx).rmux * f.c[F_CBX][vx] * f.c[F_CBX][vx]);
    en[4] += 0.5 * (MAT(M_FMATY, v).rmuy * f.c[F_CBY][v] * f.c[F_CBY][v] + MAT(M_FMATY, vy).rmuy * f.c[F_CBY][vy] * f.c[F_CBY][vy]);
    en[5] += 0.5 * (MAT(M_FMATZ, v).rmuz * f.c[F_CBZ][v] * f.c[F_CBZ][v] + MAT(M_FMATZ, vz).rmuz * f.c[F_CBZ][vz] * f.c[F_CBZ][vz]);
#undef SQ
#undef MAT
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double s = en[k];
    for (int off = 32; off; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    partial[blockIdx.x * 6 + k] = (s_sum[0][k] + s_sum[1][k]) + (s_sum[2][k] + s_sum[3][k]);
  }
}

int k_energy_f(Engine *e, double *en6) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  if (!e->mc) VH_FAIL("energy_f: no material coefficients set");
  const int nb = (int)(e->dsum_count / 6);
  if (e->f.m[0])
    hipLaunchKernelGGL(energy_f_kernel<false>, dim3(nb), dim3(256), 0, e->stream, e->f, e->mc, g, e->dsum);
  else
    hipLaunchKernelGGL(energy_f_kernel<true>, dim3(nb), dim3(256), 0, e->stream, e->f, e->mc, g, e->dsum);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipMemcpyAsync(e->host_dsum, e->dsum, sizeof(double) * 6 * nb, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  const double v0 = 0.5 * G.eps0 * G.dx * G.dy * G.dz;       // energy_f.c:171
  for (int k = 0; k < 6; k++) {
    double s = 0;
    for (int b = 0; b < nb; b++) s += e->host_dsum[b * 6 + k];
    en6[k] = s * v0;
  }
  return 0;
}


// =================================================================================================
// Divergence cleaning family and charge densities.  Reference restated:
//   src/field_advance/standard/sfa.c:213-234 (clear_rhof)       src/species_advance/standard/rho_p.c:23-86
//   remote.c:533-622 (synchronize_rho)      local.c:368-445 (local_adjust_rhof / rhob)
//   compute_div_e_err.c:6-11, compute_rhob.c:8-12, local.c:128-180 (normal-E ghosts), :298-330
//   compute_rms_div_e_err.c:36-159, compute_rms_div_b_err.c:36-93
//   clean_div_e.c:6-13,39-46   compute_div_b_err.c:44-48   clean_div_b.c:6-8,31-38, local.c:182-216
//   compute_curl_b.c:8-18      remote.c:298-414 (synchronize_tang_e_norm_b)
// Same structure as the kernels above: one launch over the box 1..n+1 with per-component
// predicates, plane kernels for ghosts / local adjustments / face messages.
// =================================================================================================

static PlaneBox node_box(const GridK &g, int axis, int plane) {   // X_NODE_LOOP: 1..n+1 along the other axes
  const int n[3] = {g.nx, g.ny, g.nz};
  PlaneBox b;
  for (int d = 0; d < 3; d++) { b.lo[d] = 1; b.n[d] = n[d] + 1; }
  b.lo[axis] = plane; b.n[axis] = 1;
  b.count = b.n[0] * b.n[1] * b.n[2];
  return b;
}
static PlaneBox face_box(const GridK &g, int axis, int plane) { return plane_box(g, axis, plane, axis, 0); }

enum { P1_COPY = 0, P1_ZERO, P1_SCALE2, P1_PACK_RHO, P1_UNPACK_RHO, P1_EXTRAPOLATE, P1_PACK1, P1_UNPACK1, P1_AVG1, P1_AVG2 };
// one box, up to two component arrays treated alike
struct Plane1Args { float *c, *d; PlaneBox b; int op, off; float sign, w0, w1, w2, w3; double *err; };
__global__ void plane1_kernel(Plane1Args A, GridK g, float *buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.b.count) return;
  const int v = plane_voxel(A.b, g, t);
  switch (A.op) {
    case P1_COPY: A.c[v] = A.sign * A.c[v + A.off]; if (A.d) A.d[v] = A.sign * A.d[v + A.off]; break;
    case P1_EXTRAPOLATE:                                                              // local.c:162-170
      A.c[v] = 2 * A.c[v + A.off] - A.c[v + 2 * A.off]; if (A.d) A.d[v] = 2 * A.d[v + A.off] - A.d[v + 2 * A.off]; break;
    case P1_ZERO: A.c[v] = 0.f; if (A.d) A.d[v] = 0.f; break;
    case P1_SCALE2: A.c[v] *= 2.f; break;
    case P1_PACK_RHO: buf[2 * t] = A.c[v]; buf[2 * t + 1] = A.d[v]; break;           // remote.c:553-557
    case P1_UNPACK_RHO:                                                                // remote.c:572-577
      A.c[v] = A.w0 * A.c[v] + A.w1 * buf[2 * t];
      A.d[v] = A.w2 * A.d[v] + A.w3 * buf[2 * t + 1];
      break;
    case P1_PACK1: buf[t] = A.c[v]; break;
    case P1_UNPACK1: A.c[v] = buf[t]; break;                                          // remote.c:182-183 / :256-258 with rw = 1, lw = 0
    case P1_AVG1: case P1_AVG2: {                                                     // remote.c:340-371
      double w1 = buf[A.op == P1_AVG2 ? 2 * t : t], w2 = A.c[v];
      A.c[v] = 0.5 * (w1 + w2);
      atomicAdd(A.err, (w1 - w2) * (w1 - w2));
      if (A.op == P1_AVG2) { w1 = buf[2 * t + 1]; w2 = A.d[v]; A.d[v] = 0.5 * (w1 + w2); }
    } break;
  }
}
static int launch_plane1(Engine *e, float *c, float *d, const PlaneBox &b, int op, int off, float sign, float *buf,
                         float w0 = 0, float w1 = 0, float w2 = 0, float w3 = 0) {
  if (b.count <= 0) return 0;
  Plane1Args A{c, d, b, op, off, sign, w0, w1, w2, w3, e->dsum};
  hipLaunchKernelGGL(plane1_kernel, dim3((b.count + 255) / 256), dim3(256), 0, e->stream, A, e->gk, buf);
  VH_CHECK(hipGetLastError());
  return 0;
}

int k_clear_rhof(Engine *e) {
  VH_CHECK(hipMemsetAsync(e->f.c[F_RHOF], 0, sizeof(float) * (size_t)e->gk.nv, e->stream));
  return 0;
}

// rho_p.c:43-84: the eight trilinear weights of a particle, added to the nodes of its cell.  The
// sums are float atomics: same values as the reference's, added in another order.
// DET (deterministic accumulation, engine.h): the weights are rounded to 64-bit fixed point and summed as integers in rho64;
// rho_finalize_kernel rounds the sums into rhof.
template <bool DET>
__global__ __launch_bounds__(256)
void accumulate_rho_p_kernel(float *__restrict__ rhof, ParticlesK p, int np, float r8V, int sy, int sz, double scale) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= np || p.i[idx] < 0) return;          // (a dead slot: engine.h, Species::n_holes)
  float w0, w1, w2, w3, w4, w5, w6, w7, t;
  t = p.dx[idx]; w0 = r8V * p.q[idx]; t *= w0; w1 = w0 + t; w0 -= t;
  t = p.dy[idx]; w3 = 1 + t; w2 = w0 * w3; w3 *= w1; t = 1 - t; w0 *= t; w1 *= t;
  t = p.dz[idx]; w7 = 1 + t; w4 = w0 * w7; w5 = w1 * w7; w6 = w2 * w7; w7 *= w3;
  t = 1 - t; w0 *= t; w1 *= t; w2 *= t; w3 *= t;
  if (DET) {
    unsigned long long *r = reinterpret_cast<unsigned long long *>(rhof) + p.i[idx];
    atomicAdd(r, to_fixed(w0, scale)); atomicAdd(r + 1, to_fixed(w1, scale)); atomicAdd(r + sy, to_fixed(w2, scale)); atomicAdd(r + sy + 1, to_fixed(w3, scale));
    atomicAdd(r + sz, to_fixed(w4, scale)); atomicAdd(r + sz + 1, to_fixed(w5, scale)); atomicAdd(r + sz + sy, to_fixed(w6, scale)); atomicAdd(r + sz + sy + 1, to_fixed(w7, scale));
    return;
  }
  float *r = rhof + p.i[idx];
  atomicAdd(r, w0); atomicAdd(r + 1, w1); atomicAdd(r + sy, w2); atomicAdd(r + sy + 1, w3);
  atomicAdd(r + sz, w4); atomicAdd(r + sz + 1, w5); atomicAdd(r + sz + sy, w6); atomicAdd(r + sz + sy + 1, w7);
}
__global__ __launch_bounds__(256)
void rho_finalize_kernel(float *__restrict__ rhof, unsigned long long *__restrict__ rho64, int nv, double inv_scale) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nv) return;
  const long long s = (long long)rho64[v];
  if (s) { rhof[v] += (float)((double)s * inv_scale); rho64[v] = 0; }
}
// from a cell-sorted species: one thread per voxel sums the weights of its particles, 8 atomics per
// occupied cell instead of per particle
__global__ __launch_bounds__(256)
void accumulate_rho_cells_kernel(float *__restrict__ rhof, ParticlesK p, const int *__restrict__ partition, int nv,
                                 float r8V, int sy, int sz) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nv) return;
  const int first = partition[v], last = partition[v + 1];
  if (first >= last) return;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
#pragma unroll 1
  for (int idx = first; idx < last; idx++) {
    float w0, w1, w2, w3, w4, w5, w6, w7, t;
    t = p.dx[idx]; w0 = r8V * p.q[idx]; t *= w0; w1 = w0 + t; w0 -= t;
    t = p.dy[idx]; w3 = 1 + t; w2 = w0 * w3; w3 *= w1; t = 1 - t; w0 *= t; w1 *= t;
    t = p.dz[idx]; w7 = 1 + t; w4 = w0 * w7; w5 = w1 * w7; w6 = w2 * w7; w7 *= w3;
    t = 1 - t; w0 *= t; w1 *= t; w2 *= t; w3 *= t;
    s0 += w0; s1 += w1; s2 += w2; s3 += w3; s4 += w4; s5 += w5; s6 += w6; s7 += w7;
  }
  float *r = rhof + v;
  atomicAdd(r, s0); atomicAdd(r + 1, s1); atomicAdd(r + sy, s2); atomicAdd(r + sy + 1, s3);
  atomicAdd(r + sz, s4); atomicAdd(r + sz + 1, s5); atomicAdd(r + sz + sy, s6); atomicAdd(r + sz + sy + 1, s7);
}
int k_accumulate_rho_p(Engine *e, Species &s) {
  if (s.np == 0 || s.chargeless) return 0;                                // charge-0 copies add nothing
  const vpic_hip_grid_t &G = e->grid;
  const float r8V = 0.125 * G.rdx * G.rdy * G.rdz;                       // rho_p.c:37
  if (e->det_acc) {
    // deterministic accumulation: per-particle fixed-point atomics into rho64, then one rounding into rhof (the species are
    // added to rhof one after the other, in their fixed order)
    if (acc_prepare_det(e)) return 1;
    const size_t nv = (size_t)e->gk.nv;
    if (!e->rho64) { VH_CHECK(hipMalloc(&e->rho64, sizeof(unsigned long long) * nv)); VH_CHECK(hipMemsetAsync(e->rho64, 0, sizeof(unsigned long long) * nv, e->stream)); }
    int ex = 0; (void)frexp(8.0 * (double)r8V, &ex);
    const double scale = ldexp(e->acc_scale, -ex);                       // a weight is at most 8 r8V |q|
    hipLaunchKernelGGL(accumulate_rho_p_kernel<true>, dim3((unsigned)((s.np + 255) / 256)), dim3(256), 0, e->stream,
                       reinterpret_cast<float *>(e->rho64), s.p, (int)s.np, r8V, e->gk.sy, e->gk.sz, scale);
    hipLaunchKernelGGL(rho_finalize_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, e->stream, e->f.c[F_RHOF], e->rho64, (int)nv, 1.0 / scale);
    VH_CHECK(hipGetLastError());
    return 0;
  }
  if (s.np >= 4 * (int64_t)e->gk.nv && s.nm == 0 && !e->knobs.rho_per_particle) {
    if (!s.partition_valid && k_sort_p(e, s)) return 1;
    hipLaunchKernelGGL(accumulate_rho_cells_kernel, dim3((unsigned)((e->gk.nv + 255) / 256)), dim3(256), 0, e->stream,
                       e->f.c[F_RHOF], s.p, s.partition, e->gk.nv, r8V, e->gk.sy, e->gk.sz);
    VH_CHECK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(accumulate_rho_p_kernel<false>, dim3((unsigned)((s.np + 255) / 256)), dim3(256), 0, e->stream,
                     e->f.c[F_RHOF], s.p, (int)s.np, r8V, e->gk.sy, e->gk.sz, 0.0);
  VH_CHECK(hipGetLastError());
  return 0;
}

static int local_adjust_rho(Engine *e) {            // local.c:368-445: six faces for rhof, then six for rhob
  const GridK &g = e->gk;
  for (int comp = 0; comp < 2; comp++)
    for (int face = 0; face < 6; face++) {
      const int bc = g.fbc[face];
      if (bc >= 0) continue;
      const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis);
      const PlaneBox b = node_box(g, axis, hi ? n + 1 : 1);
      if (bc == VPIC_PEC_FIELDS) { if (launch_plane1(e, e->f.c[comp ? F_RHOB : F_RHOF], nullptr, b, P1_ZERO, 0, 1.f, nullptr)) return 1; }
      else if (!comp)            { if (launch_plane1(e, e->f.c[F_RHOF], nullptr, b, P1_SCALE2, 0, 1.f, nullptr)) return 1; }
    }
  return 0;
}
int k_rho_count(const Engine *e, int dir) {
  const GridK &g = e->gk;
  const int a = dir % 3;
  return 2 * (n_axis(g, (a + 1) % 3) + 1) * (n_axis(g, (a + 2) % 3) + 1);   // remote.c:546 without the cell-size float
}
int k_pack_rho(Engine *e, int dir, float *buf) {
  const GridK &g = e->gk;
  const int axis = dir % 3;
  return launch_plane1(e, e->f.c[F_RHOF], e->f.c[F_RHOB], node_box(g, axis, dir < 3 ? 1 : n_axis(g, axis) + 1), P1_PACK_RHO, 0, 1.f, buf);
}
int k_unpack_rho(Engine *e, int dir, const float *buf) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  const int axis = dir % 3;
  const float d = axis == 0 ? G.dx : axis == 1 ? G.dy : G.dz;
  float hrw = d, hlw = hrw + d, lw, rw;               // remote.c:566-571, remote cell size == ours
  hrw /= hlw; hlw = d / hlw; lw = hlw + hlw; rw = hrw + hrw;
  return launch_plane1(e, e->f.c[F_RHOF], e->f.c[F_RHOB], node_box(g, axis, dir < 3 ? n_axis(g, axis) + 1 : 1), P1_UNPACK_RHO, 0, 1.f,
                       const_cast<float *>(buf), lw, rw, hlw, hrw);
}
int k_local_adjust_rho(Engine *e) { return local_adjust_rho(e); }
int k_synchronize_rho_self(Engine *e, int axis) {
  const GridK &g = e->gk;
  if (g.fbc[axis] != g.rank || g.fbc[axis + 3] != g.rank) return 0;
  if (k_pack_rho(e, axis, e->face_buf[0]) || k_pack_rho(e, axis + 3, e->face_buf[1])) return 1;
  if (k_unpack_rho(e, axis, e->face_buf[0]) || k_unpack_rho(e, axis + 3, e->face_buf[1])) return 1;
  return 0;
}
int k_synchronize_rho_local(Engine *e) {
  if (local_adjust_rho(e)) return 1;
  for (int axis = 0; axis < 3; axis++)
    if (k_synchronize_rho_self(e, axis)) return 1;
  return 0;
}

// normal-E ghosts of the faces this domain shares with itself (remote.c:136-207) and of its local
// faces (local.c:128-180)
static int ghost_norm_e(Engine *e) {
  const GridK &g = e->gk;
  for (int dir = 0; dir < 6; dir++) {
    if (g.fbc[dir] != g.rank) continue;
    const int axis = dir % 3, n = n_axis(g, axis), from = dir < 3 ? 1 : n, to = dir < 3 ? n + 1 : 0;
    if (launch_plane1(e, e->f.c[F_EX + axis], nullptr, node_box(g, axis, to), P1_COPY, (from - to) * stride_axis(g, axis), 1.f, nullptr)) return 1;
  }
  for (int face = 0; face < 6; face++) {
    const int bc = g.fbc[face];
    if (bc >= 0) continue;
    const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis), st = stride_axis(g, axis);
    float sign;
    if (bc == VPIC_ABSORB_FIELDS) {
      if (launch_plane1(e, e->f.c[F_EX + axis], e->f.c[F_TCAX + axis], node_box(g, axis, hi ? n + 1 : 0), P1_EXTRAPOLATE, hi ? -st : st, 1.f, nullptr)) return 1;
      continue;
    }
    if (bc == VPIC_PEC_FIELDS) sign = 1.f;
    else if (bc == VPIC_SYMMETRIC_FIELDS || bc == VPIC_PMC_FIELDS) sign = -1.f;
    else VH_FAIL("Bad boundary condition encountered.");
    if (launch_plane1(e, e->f.c[F_EX + axis], e->f.c[F_TCAX + axis], node_box(g, axis, hi ? n + 1 : 0), P1_COPY, hi ? -st : st, sign, nullptr)) return 1;
  }
  return 0;
}

template <bool SINGLE_MATERIAL, bool RHOB>
__global__ __launch_bounds__(256)
void div_e_kernel(FieldsK f, const vpic_material_coefficient_t *__restrict__ m, GridK g, float px, float py, float pz, float cj) {
  int x, y, z;
  if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z), vx = v - 1, vy = v - g.sy, vz = v - g.sz;
#define MAT(which, vox) (SINGLE_MATERIAL ? m[0] : m[f.m[which][vox]])
  const float s = px * (MAT(M_EMATX, v).epsx * f.c[F_EX][v] - MAT(M_EMATX, vx).epsx * f.c[F_EX][vx]) +
                  py * (MAT(M_EMATY, v).epsy * f.c[F_EY][v] - MAT(M_EMATY, vy).epsy * f.c[F_EY][vy]) +
                  pz * (MAT(M_EMATZ, v).epsz * f.c[F_EZ][v] - MAT(M_EMATZ, vz).epsz * f.c[F_EZ][vz]);
  if (RHOB) f.c[F_RHOB][v] = MAT(M_NMAT, v).nonconductive * (s - f.c[F_RHOF][v]);
  else      f.c[F_DIV_E_ERR][v] = MAT(M_NMAT, v).nonconductive * (s - cj * (f.c[F_RHOF][v] + f.c[F_RHOB][v]));
#undef MAT
}
static int div_e_like(Engine *e, bool rhob) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  if (!e->mc) VH_FAIL("no material coefficients set");
  const float px = (g.nx > 1) ? (rhob ? G.eps0 * G.rdx : G.rdx) : 0;
  const float py = (g.ny > 1) ? (rhob ? G.eps0 * G.rdy : G.rdy) : 0;
  const float pz = (g.nz > 1) ? (rhob ? G.eps0 * G.rdz : G.rdz) : 0;
  const float cj = 1. / G.eps0;
  if (ghost_norm_e(e)) return 1;
  const unsigned n = (unsigned)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
  const dim3 grid((n + 255) / 256), block(256);
  if (e->f.m[0]) {
    if (rhob) hipLaunchKernelGGL((div_e_kernel<false, true>), grid, block, 0, e->stream, e->f, e->mc, g, px, py, pz, cj);
    else      hipLaunchKernelGGL((div_e_kernel<false, false>), grid, block, 0, e->stream, e->f, e->mc, g, px, py, pz, cj);
  } else {
    if (rhob) hipLaunchKernelGGL((div_e_kernel<true, true>), grid, block, 0, e->stream, e->f, e->mc, g, px, py, pz, cj);
    else      hipLaunchKernelGGL((div_e_kernel<true, false>), grid, block, 0, e->stream, e->f, e->mc, g, px, py, pz, cj);
  }
  VH_CHECK(hipGetLastError());
  for (int face = 0; face < 6; face++) {             // local.c:298-330 (PEC and absorbing) / :414-445 (PEC)
    if (!(g.fbc[face] == VPIC_PEC_FIELDS || (!rhob && g.fbc[face] == VPIC_ABSORB_FIELDS))) continue;
    const int axis = face % 3, hi = face >= 3, nn = n_axis(g, axis);
    if (launch_plane1(e, e->f.c[rhob ? F_RHOB : F_DIV_E_ERR], nullptr, node_box(g, axis, hi ? nn + 1 : 1), P1_ZERO, 0, 1.f, nullptr)) return 1;
  }
  return 0;
}
int k_compute_div_e_err(Engine *e) { return div_e_like(e, false); }
int k_compute_rhob(Engine *e) { return div_e_like(e, true); }

// sums of squares, double partials per workgroup; NODE: compute_rms_div_e_err.c (float products
// inside, half/quarter/eighth-weighted double products on faces/edges/corners), else
// compute_rms_div_b_err.c (cells 1..n, float products)
template <bool NODE>
__global__ __launch_bounds__(256)
void rms_kernel(const float *__restrict__ c, GridK g, double *__restrict__ partial) {
  __shared__ double s_sum[4];
  double err = 0;
  const Box3 box = NODE ? Box3{g.nx + 1, g.ny + 1, g.nz + 1} : Box3{g.nx, g.ny, g.nz};
  const unsigned total = (unsigned)box.bx * box.by * box.bz;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
    int x, y, z;
    decode(box, t, x, y, z);
    const float v = c[VOX(x, y, z)];
    if (NODE) {
      const int nb = (x == 1 || x == g.nx + 1) + (y == 1 || y == g.ny + 1) + (z == 1 || z == g.nz + 1);
      if (nb == 0) err += v * v;
      else err += (nb == 1 ? 0.5 : nb == 2 ? 0.25 : 0.125) * (double)v * (double)v;
    } else err += v * v;
  }
  for (int off = 32; off; off >>= 1) err += __shfl_down(err, off);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = err;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}
// local2[0] = sum * dV, local2[1] = volume: what the reference hands to mp_allsum_d
static int rms_local(Engine *e, bool node, double *local2) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  const int nb = 1024;
  if (node) hipLaunchKernelGGL(rms_kernel<true>, dim3(nb), dim3(256), 0, e->stream, e->f.c[F_DIV_E_ERR], g, e->dsum);
  else      hipLaunchKernelGGL(rms_kernel<false>, dim3(nb), dim3(256), 0, e->stream, e->f.c[F_DIV_B_ERR], g, e->dsum);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipMemcpyAsync(e->host_dsum, e->dsum, sizeof(double) * nb, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  double s = 0;
  for (int b = 0; b < nb; b++) s += e->host_dsum[b];
  local2[0] = s * G.dx * G.dy * G.dz;
  local2[1] = (double)g.nx * g.ny * g.nz * G.dx * G.dy * G.dz;
  return 0;
}
int k_rms_div_e_err_local(Engine *e, double *local2) { return rms_local(e, true, local2); }
int k_rms_div_b_err_local(Engine *e, double *local2) { return rms_local(e, false, local2); }

static void marder_p(const Engine *e, float &px, float &py, float &pz) {   // clean_div_e.c:39-46
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  px = (g.nx > 1) ? G.rdx : 0; py = (g.ny > 1) ? G.rdy : 0; pz = (g.nz > 1) ? G.rdz : 0;
  const float alphadt = 0.3888889 / (px * px + py * py + pz * pz);
  px *= alphadt; py *= alphadt; pz *= alphadt;
}

template <bool SINGLE_MATERIAL>
__global__ __launch_bounds__(256)
void clean_div_e_kernel(FieldsK f, const vpic_material_coefficient_t *__restrict__ m, GridK g, float px, float py, float pz) {
  int x, y, z;
  if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z);
  const float d0 = f.c[F_DIV_E_ERR][v];
#define MAT(which, vox) (SINGLE_MATERIAL ? m[0] : m[f.m[which][vox]])
  if (x <= g.nx) f.c[F_EX][v] += MAT(M_EMATX, v).drivex * px * (f.c[F_DIV_E_ERR][v + 1] - d0);
  if (y <= g.ny) f.c[F_EY][v] += MAT(M_EMATY, v).drivey * py * (f.c[F_DIV_E_ERR][v + g.sy] - d0);
  if (z <= g.nz) f.c[F_EZ][v] += MAT(M_EMATZ, v).drivez * pz * (f.c[F_DIV_E_ERR][v + g.sz] - d0);
#undef MAT
}
int k_clean_div_e(Engine *e) {
  const GridK &g = e->gk;
  if (!e->mc) VH_FAIL("clean_div_e: no material coefficients set");
  float px, py, pz;
  marder_p(e, px, py, pz);
  const unsigned n = (unsigned)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
  if (e->f.m[0]) hipLaunchKernelGGL(clean_div_e_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, e->mc, g, px, py, pz);
  else           hipLaunchKernelGGL(clean_div_e_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, e->mc, g, px, py, pz);
  VH_CHECK(hipGetLastError());
  return local_adjust_tang_e(e);
}

__global__ __launch_bounds__(256)
void div_b_kernel(FieldsK f, GridK g, float px, float py, float pz) {
  int x, y, z;
  if (!decode(Box3{g.nx, g.ny, g.nz}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z);
  f.c[F_DIV_B_ERR][v] = px * (f.c[F_CBX][v + 1] - f.c[F_CBX][v]) + py * (f.c[F_CBY][v + g.sy] - f.c[F_CBY][v]) +
                        pz * (f.c[F_CBZ][v + g.sz] - f.c[F_CBZ][v]);
}
int k_compute_div_b_err(Engine *e) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  const float px = (g.nx > 1) ? G.rdx : 0, py = (g.ny > 1) ? G.rdy : 0, pz = (g.nz > 1) ? G.rdz : 0;
  const unsigned n = (unsigned)g.nx * g.ny * g.nz;
  hipLaunchKernelGGL(div_b_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, g, px, py, pz);
  VH_CHECK(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256)
void clean_div_b_kernel(FieldsK f, GridK g, float px, float py, float pz) {
  int x, y, z;
  if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z);
  const float d0 = f.c[F_DIV_B_ERR][v];
  if (y <= g.ny && z <= g.nz) f.c[F_CBX][v] += px * (d0 - f.c[F_DIV_B_ERR][v - 1]);
  if (z <= g.nz && x <= g.nx) f.c[F_CBY][v] += py * (d0 - f.c[F_DIV_B_ERR][v - g.sy]);
  if (x <= g.nx && y <= g.ny) f.c[F_CBZ][v] += pz * (d0 - f.c[F_DIV_B_ERR][v - g.sz]);
}
int k_clean_div_b(Engine *e) {
  const GridK &g = e->gk;
  float px, py, pz;
  marder_p(e, px, py, pz);
  // div_b_err ghosts: remote.c:209-281 for faces shared with this same domain, local.c:182-216
  for (int dir = 0; dir < 6; dir++) {
    if (g.fbc[dir] != g.rank) continue;
    const int axis = dir % 3, n = n_axis(g, axis), from = dir < 3 ? 1 : n, to = dir < 3 ? n + 1 : 0;
    if (launch_plane1(e, e->f.c[F_DIV_B_ERR], nullptr, face_box(g, axis, to), P1_COPY, (from - to) * stride_axis(g, axis), 1.f, nullptr)) return 1;
  }
  for (int face = 0; face < 6; face++) {
    const int bc = g.fbc[face];
    if (bc >= 0) continue;
    const int axis = face % 3, hi = face >= 3, n = n_axis(g, axis), st = stride_axis(g, axis);
    const PlaneBox b = face_box(g, axis, hi ? n + 1 : 0);
    int rc;
    if (bc == VPIC_PEC_FIELDS) rc = launch_plane1(e, e->f.c[F_DIV_B_ERR], nullptr, b, P1_COPY, hi ? -st : st, 1.f, nullptr);
    else if (bc == VPIC_SYMMETRIC_FIELDS || bc == VPIC_PMC_FIELDS) rc = launch_plane1(e, e->f.c[F_DIV_B_ERR], nullptr, b, P1_COPY, hi ? -st : st, -1.f, nullptr);
    else rc = launch_plane1(e, e->f.c[F_DIV_B_ERR], nullptr, b, P1_ZERO, 0, 1.f, nullptr);
    if (rc) return 1;
  }
  const unsigned n = (unsigned)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
  hipLaunchKernelGGL(clean_div_b_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, g, px, py, pz);
  VH_CHECK(hipGetLastError());
  return local_adjust_norm_b(e);
}

template <bool SINGLE_MATERIAL>
__global__ __launch_bounds__(256)
void curl_b_kernel(FieldsK f, const vpic_material_coefficient_t *__restrict__ m, GridK g, float px, float py, float pz) {
  int x, y, z;
  if (!decode(Box3{g.nx + 1, g.ny + 1, g.nz + 1}, blockIdx.x * 256u + threadIdx.x, x, y, z)) return;
  const int v = VOX(x, y, z), vx = v - 1, vy = v - g.sy, vz = v - g.sz;
  const float cbx = f.c[F_CBX][v], cby = f.c[F_CBY][v], cbz = f.c[F_CBZ][v];
#define MAT(which, vox) (SINGLE_MATERIAL ? m[0] : m[f.m[which][vox]])
  if (x <= g.nx) f.c[F_TCAX][v] = py * (cbz * MAT(M_FMATZ, v).rmuz - f.c[F_CBZ][vy] * MAT(M_FMATZ, vy).rmuz) -
                                  pz * (cby * MAT(M_FMATY, v).rmuy - f.c[F_CBY][vz] * MAT(M_FMATY, vz).rmuy);
  if (y <= g.ny) f.c[F_TCAY][v] = pz * (cbx * MAT(M_FMATX, v).rmux - f.c[F_CBX][vz] * MAT(M_FMATX, vz).rmux) -
                                  px * (cbz * MAT(M_FMATZ, v).rmuz - f.c[F_CBZ][vx] * MAT(M_FMATZ, vx).rmuz);
  if (z <= g.nz) f.c[F_TCAZ][v] = px * (cby * MAT(M_FMATY, v).rmuy - f.c[F_CBY][vx] * MAT(M_FMATY, vx).rmuy) -
                                  py * (cbx * MAT(M_FMATX, v).rmux - f.c[F_CBX][vy] * MAT(M_FMATX, vy).rmux);
#undef MAT
}
int k_compute_curl_b(Engine *e) {
  const GridK &g = e->gk;
  const vpic_hip_grid_t &G = e->grid;
  if (!e->mc) VH_FAIL("compute_curl_b: no material coefficients set");
  const float px = (g.nx > 1) ? G.cvac * G.dt * G.rdx : 0;
  const float py = (g.ny > 1) ? G.cvac * G.dt * G.rdy : 0;
  const float pz = (g.nz > 1) ? G.cvac * G.dt * G.rdz : 0;
  if (self_ghost_tang_b(e)) return 1;
  if (local_ghost_tang_b(e)) return 1;
  const unsigned n = (unsigned)(g.nx + 1) * (g.ny + 1) * (g.nz + 1);
  if (e->f.m[0]) hipLaunchKernelGGL(curl_b_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, e->mc, g, px, py, pz);
  else           hipLaunchKernelGGL(curl_b_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->f, e->mc, g, px, py, pz);
  VH_CHECK(hipGetLastError());
  return 0;
}

// ---- face messages of the family for faces shared with ANOTHER domain (uniform meshes, without
// the leading cell-size float).  kind 0: normal E, remote.c:136-207 (node plane 1 / n -> ghost
// n+1 / 0); kind 1: div_b_err, remote.c:209-281 (face plane 1 / n -> ghost); kind 2: tang E and norm
// B, remote.c:298-414 (plane 1 / n+1: cB_X over the face box, then (e_Y, tca_Y), then (e_Z, tca_Z)
// over their edge boxes; the receiver averages and adds the squared differences of cB and e to
// e->dsum[0]).
int k_msg_count(const Engine *e, int kind, int dir) {
  const GridK &g = e->gk;
  const int a = dir % 3, nY = n_axis(g, (a + 1) % 3), nZ = n_axis(g, (a + 2) % 3);
  if (kind == 0) return (nY + 1) * (nZ + 1);
  if (kind == 1) return nY * nZ;
  return nY * nZ + 2 * nY * (nZ + 1) + 2 * nZ * (nY + 1);
}
int k_pack_msg(Engine *e, int kind, int dir, float *buf) {
  const GridK &g = e->gk;
  const int axis = dir % 3, n = n_axis(g, axis);
  if (kind == 0) return launch_plane1(e, e->f.c[F_EX + axis], nullptr, node_box(g, axis, dir < 3 ? 1 : n), P1_PACK1, 0, 1.f, buf);
  if (kind == 1) return launch_plane1(e, e->f.c[F_DIV_B_ERR], nullptr, face_box(g, axis, dir < 3 ? 1 : n), P1_PACK1, 0, 1.f, buf);
  const int plane = dir < 3 ? 1 : n + 1;
  PlaneBox b = face_box(g, axis, plane);
  if (launch_plane1(e, e->f.c[F_CBX + axis], nullptr, b, P1_PACK1, 0, 1.f, buf)) return 1;
  int k = b.count;
  for (int t = 1; t <= 2; t++) {
    const int ca = (axis + t) % 3;
    b = plane_box(g, axis, plane, ca, 1);
    if (launch_plane1(e, e->f.c[F_EX + ca], e->f.c[F_TCAX + ca], b, P1_PACK_RHO, 0, 1.f, buf + k)) return 1;
    k += 2 * b.count;
  }
  return 0;
}
int k_unpack_msg(Engine *e, int kind, int dir, const float *cbuf) {
  const GridK &g = e->gk;
  float *buf = const_cast<float *>(cbuf);
  const int axis = dir % 3, n = n_axis(g, axis);
  if (kind == 0) return launch_plane1(e, e->f.c[F_EX + axis], nullptr, node_box(g, axis, dir < 3 ? n + 1 : 0), P1_UNPACK1, 0, 1.f, buf);
  if (kind == 1) return launch_plane1(e, e->f.c[F_DIV_B_ERR], nullptr, face_box(g, axis, dir < 3 ? n + 1 : 0), P1_UNPACK1, 0, 1.f, buf);
  const int plane = dir < 3 ? n + 1 : 1;
  PlaneBox b = face_box(g, axis, plane);
  if (launch_plane1(e, e->f.c[F_CBX + axis], nullptr, b, P1_AVG1, 0, 1.f, buf)) return 1;
  int k = b.count;
  for (int t = 1; t <= 2; t++) {
    const int ca = (axis + t) % 3;
    b = plane_box(g, axis, plane, ca, 1);
    if (launch_plane1(e, e->f.c[F_EX + ca], e->f.c[F_TCAX + ca], b, P1_AVG2, 0, 1.f, buf + k)) return 1;
    k += 2 * b.count;
  }
  return 0;
}
int k_err_begin(Engine *e) { VH_CHECK(hipMemsetAsync(e->dsum, 0, sizeof(double), e->stream)); return 0; }
int k_err_read(Engine *e, double *err) {
  VH_CHECK(hipMemcpyAsync(e->host_dsum, e->dsum, sizeof(double), hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  *err = e->host_dsum[0];
  return 0;
}
int k_local_adjust_tang_e_norm_b(Engine *e) { return local_adjust_tang_e(e) || local_adjust_norm_b(e); }
// one axis of synchronize_tang_e_norm_b for a domain that shares both faces of the axis with itself:
// both planes are packed before either is averaged into, as both sends precede both receives
int k_synchronize_tang_e_norm_b_self(Engine *e, int axis) {
  const GridK &g = e->gk;
  if (g.fbc[axis] != g.rank || g.fbc[axis + 3] != g.rank) return 0;
  if (k_pack_msg(e, 2, axis, e->face_buf[0]) || k_pack_msg(e, 2, axis + 3, e->face_buf[1])) return 1;
  if (k_unpack_msg(e, 2, axis, e->face_buf[0]) || k_unpack_msg(e, 2, axis + 3, e->face_buf[1])) return 1;
  return 0;
}
int k_synchronize_tang_e_norm_b_local(Engine *e, double *err_out) {
  if (k_local_adjust_tang_e_norm_b(e) || k_err_begin(e)) return 1;
  for (int axis = 0; axis < 3; axis++)
    if (k_synchronize_tang_e_norm_b_self(e, axis)) return 1;
  return k_err_read(e, err_out);
}

// ---- hydro array (sf_interface/sf_interface.c:29-36, sf_interface/hydro.c:28-200) ----------------
// hydro_t stays array-of-struct on the device (16 floats per voxel, 14 used): it is written by
// scattered atomics and read back whole.
int ensure_hydro(Engine *e) {
  if (e->hydro) return 0;
  VH_CHECK(hipMalloc(&e->hydro, sizeof(vpic_hydro_t) * (size_t)e->gk.nv));
  VH_CHECK(hipMemsetAsync(e->hydro, 0, sizeof(vpic_hydro_t) * (size_t)e->gk.nv, e->stream));
  const size_t need = 14 * (size_t)k_rho_count(e, 0) / 2, need1 = 14 * (size_t)k_rho_count(e, 1) / 2, need2 = 14 * (size_t)k_rho_count(e, 2) / 2;
  const size_t n = std::max(need, std::max(need1, need2));
  VH_CHECK(hipMalloc(&e->hydro_buf[0], sizeof(float) * n));
  VH_CHECK(hipMalloc(&e->hydro_buf[1], sizeof(float) * n));
  return 0;
}
int k_clear_hydro(Engine *e) {
  if (ensure_hydro(e)) return 1;
  VH_CHECK(hipMemsetAsync(e->hydro, 0, sizeof(vpic_hydro_t) * (size_t)e->gk.nv, e->stream));
  return 0;
}
enum { H_SCALE2 = 0, H_PACK, H_UNPACK };
__global__ void hydro_plane_kernel(float *h, PlaneBox b, GridK g, int op, float lw, float rw, float *buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.count * 14) return;
  const int node = t / 14, c = t - node * 14;
  float *m = h + (size_t)plane_voxel(b, g, node) * 16 + c;
  if (op == H_SCALE2) *m *= 2.f;                         // hydro.c:178-191
  else if (op == H_PACK) buf[t] = *m;                    // hydro.c:48-61
  else *m = lw * *m + rw * buf[t];                       // hydro.c:79-92
}
static int hydro_plane(Engine *e, const PlaneBox &b, int op, float lw, float rw, float *buf) {
  const int n = b.count * 14;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(hydro_plane_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, reinterpret_cast<float *>(e->hydro), b, e->gk, op, lw, rw, buf);
  VH_CHECK(hipGetLastError());
  return 0;
}
int k_local_adjust_hydro(Engine *e) {
  if (ensure_hydro(e)) return 1;
  const GridK &g = e->gk;
  for (int face = 0; face < 6; face++) {
    if (g.fbc[face] >= 0) continue;
    const int axis = face % 3, hi = face >= 3;
    if (hydro_plane(e, node_box(g, axis, hi ? n_axis(g, axis) + 1 : 1), H_SCALE2, 0, 0, nullptr)) return 1;
  }
  return 0;
}
int k_hydro_count(const Engine *e, int dir) { return 7 * k_rho_count(e, dir); }
int k_pack_hydro(Engine *e, int dir, float *buf) {
  if (ensure_hydro(e)) return 1;
  const int axis = dir % 3;
  return hydro_plane(e, node_box(e->gk, axis, dir < 3 ? 1 : n_axis(e->gk, axis) + 1), H_PACK, 0, 0, buf);
}
int k_unpack_hydro(Engine *e, int dir, const float *buf) {
  if (ensure_hydro(e)) return 1;
  const vpic_hip_grid_t &G = e->grid;
  const int axis = dir % 3;
  const float d = axis == 0 ? G.dx : axis == 1 ? G.dy : G.dz;
  float rw = d, lw = rw + d;                             // hydro.c:69-74, remote cell size == ours
  rw /= lw; lw = d / lw; lw += lw; rw += rw;
  return hydro_plane(e, node_box(e->gk, axis, dir < 3 ? n_axis(e->gk, axis) + 1 : 1), H_UNPACK, lw, rw, const_cast<float *>(buf));
}
int k_synchronize_hydro_self(Engine *e, int axis) {
  const GridK &g = e->gk;
  if (ensure_hydro(e)) return 1;
  if (g.fbc[axis] != g.rank || g.fbc[axis + 3] != g.rank) return 0;
  if (k_pack_hydro(e, axis, e->hydro_buf[0]) || k_pack_hydro(e, axis + 3, e->hydro_buf[1])) return 1;
  if (k_unpack_hydro(e, axis, e->hydro_buf[0]) || k_unpack_hydro(e, axis + 3, e->hydro_buf[1])) return 1;
  return 0;
}
int k_synchronize_hydro_local(Engine *e) {
  if (k_local_adjust_hydro(e)) return 1;
  for (int axis = 0; axis < 3; axis++)
    if (k_synchronize_hydro_self(e, axis)) return 1;
  return 0;
}

// ---- field_dump / hydro_dump payloads (src/vpic/dump.cxx:1116-1364, 1366-1552) ---------------------
// One thread per 32-bit output word.  The fields live as one array per component, so a banded,
// strided dump is a coalesced read of exactly the components asked for; nothing else moves.
struct DumpShape {
  int no[3], n[3], s[3];   // outputs per axis (without the two boundary entries), cells, stride
  int dim[3];              // extents of the output block
  int inner;               // INTERLEAVE_INNER offsets (no far boundary entry)
  int unit;                // all strides 1: the reference's fast branch, plain indices
  int nwords;
  int words[32];
};
__device__ __forceinline__ int dump_offset(const DumpShape &d, int a, int i) {
  if (i == 0) return 0;
  if (!d.inner && i == d.no[a] + 1) return d.n[a] + 1;
  return d.unit ? i : i * d.s[a] - 1;   // also on an axis whose own stride is 1 (dump.cxx:1262-1273)
}
__device__ __forceinline__ uint32_t field_word(const FieldsK &f, int v, int w, int nv) {
  if (w >= 20) { v++; w -= 20; if (v >= nv) return 0u; }       // the reference's index runs into the next record
  if (w < F_NCOMP) return __float_as_uint(f.c[w][v]);
  if (!f.m[0]) return 0u;
  const int c = 2 * (w - F_NCOMP);
  return (uint32_t)f.m[c][v] | ((uint32_t)f.m[c + 1][v] << 16);
}
template <int WHAT>
__global__ __launch_bounds__(256)
void dump_gather_kernel(FieldsK f, const uint32_t *__restrict__ hydro, GridK g, DumpShape d, int layout,
                        uint32_t *__restrict__ out, size_t total) {
  constexpr int W = WHAT == VPIC_HIP_DUMP_FIELDS ? 20 : 16;
  const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= total) return;
  const size_t plane = (size_t)d.dim[0] * d.dim[1], block = plane * d.dim[2];
  int w;
  size_t r;
  if (layout == VPIC_HIP_DUMP_BAND) { const int b = (int)(t / block); r = t - (size_t)b * block; w = d.words[b]; }
  else { r = t / W; w = (int)(t - r * W); }
  int v;
  if (layout == VPIC_HIP_DUMP_INTERLEAVE_INNER && d.unit) v = (int)r;
  else {
    const int k = (int)(r / plane), rr = (int)(r - (size_t)k * plane), j = rr / d.dim[0], i = rr - j * d.dim[0];
    v = dump_offset(d, 0, i) + g.sy * dump_offset(d, 1, j) + g.sz * dump_offset(d, 2, k);
  }
  out[t] = WHAT == VPIC_HIP_DUMP_FIELDS ? field_word(f, v, w, g.nv) : hydro[(size_t)v * 16 + w];
}
int k_dump_gather(Engine *e, int what, int layout, const int32_t *words, int nwords, int sx, int sy, int sz,
                  void *out, size_t out_bytes) {
  DumpShape d = {};
  d.n[0] = e->gk.nx; d.n[1] = e->gk.ny; d.n[2] = e->gk.nz; d.s[0] = sx; d.s[1] = sy; d.s[2] = sz;
  d.inner = layout == VPIC_HIP_DUMP_INTERLEAVE_INNER;
  d.unit = sx == 1 && sy == 1 && sz == 1;
  for (int a = 0; a < 3; a++) { d.no[a] = d.n[a] / d.s[a]; d.dim[a] = d.no[a] + (d.inner ? 0 : 2); }
  const int W = what == VPIC_HIP_DUMP_FIELDS ? 20 : 16;
  d.nwords = layout == VPIC_HIP_DUMP_BAND ? nwords : W;
  for (int k = 0; k < nwords && layout == VPIC_HIP_DUMP_BAND; k++) d.words[k] = words[k];
  const size_t total = (size_t)d.dim[0] * d.dim[1] * d.dim[2] * d.nwords;
  if (out_bytes != total * 4) VH_FAIL("dump buffer size does not match the request");
  if (what == VPIC_HIP_DUMP_HYDRO && ensure_hydro(e)) return 1;
  if (ensure_stage(e, total * 4)) return 1;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (what == VPIC_HIP_DUMP_FIELDS)
    hipLaunchKernelGGL(dump_gather_kernel<VPIC_HIP_DUMP_FIELDS>, grid, dim3(256), 0, e->stream, e->f, (const uint32_t *)nullptr, e->gk, d, layout, (uint32_t *)e->stage, total);
  else
    hipLaunchKernelGGL(dump_gather_kernel<VPIC_HIP_DUMP_HYDRO>, grid, dim3(256), 0, e->stream, e->f, (const uint32_t *)e->hydro, e->gk, d, layout, (uint32_t *)e->stage, total);
  VH_CHECK(hipGetLastError());
  VH_CHECK(hipMemcpyAsync(out, e->stage, total * 4, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  return 0;
}

}  // namespace vpichip

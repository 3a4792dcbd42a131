// push_device.h -- device functions shared by the push kernel (push.hip) and the injector path of
// boundary_p (particles.hip): the LDS accumulator window, the 12-term streak deposit and move_p.
#pragma once
#include "engine.h"

namespace vpichip {

#ifndef VPIC_HIP_PUSH_THREADS
#define VPIC_HIP_PUSH_THREADS 256
#endif
constexpr int PUSH_THREADS = VPIC_HIP_PUSH_THREADS;     // a workgroup of the push kernel: 4 wavefronts share one accumulator window
constexpr int FLUSH_CELLS = PUSH_THREADS / 12;          // accumulators a sweep of the flush covers (12 consecutive threads each)
// Correctly rounded 1/sqrt-free pieces of the push for operands that are KNOWN to be ordinary numbers:
// the compiler's IEEE sequences for x / y and sqrtf(x) start by rescaling operands near the ends of
// the exponent range (v_div_scale x2 ... v_div_fixup; a compare, two selects and two multiplies around
// v_sqrt_f32) -- for 1 + u.u >= 1 and the quotients formed with it those steps never do anything.
// What is left is the very same arithmetic (reciprocal refined once, quotient refined twice; root
// nudged by one ulp either way against the residual), so the results are bit-identical.
__device__ __forceinline__ float sqrt_normal(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  s = (r_dn <= 0.f) ? s_dn : s;
  s = (r_up > 0.f) ? s_up : s;
  return s;
}
__device__ __forceinline__ float div_normal(float a, float b) {
  float rcp = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, rcp, 1.0f);
  rcp = __builtin_fmaf(e, rcp, rcp);
  float q = a * rcp;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, rcp, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, rcp, q);
}

constexpr int PUSH_ITERS = 64;   // most passes a wavefront makes over its span (high-ppc decks: 512 ppc runs best at 64)
// a window base no voxel index can match: every deposit goes to the global accumulator
constexpr int NO_WINDOW = -(1 << 30);

// The accumulator window in LDS: 5 segments (own row, +y, -y, +z, -z) of WX cells x 12 sums.  Two instances:
//   Window<false>  float,  WX = 62: 27.2 KB per workgroup with the crosser queues, six workgroups per CU;
//   Window<true>   double, WX = 42: 32.5 KB, five workgroups per CU.
// Measured (tools/ubench/lds_rate.hip): ds_add_f32 costs the LDS 3 clocks per LIVE LANE (193 for a full
// wavefront), ds_add_f64 8 clocks per instruction whatever the lanes.  A main pass has two or three run tails per
// atomic instruction (9 clocks either way); a pass of cell-crossers has about twenty (60 against 8 + conflicts).
// So the double window pays for crossing-heavy (hot) plasmas -- +15 % at 0.3 cells per step -- and costs 10 % on a
// cold 32-ppc deck, where the smaller window halves the chunk a workgroup amortises its set-up and flush over:
// the host picks the instance per species from the crossing fraction the kernel counts (push.hip, k_advance_p).
constexpr int WMARGIN = 4;                // cells of the segment that precede the chunk's first cell
constexpr int NSEG = 5;                   // own row, +y, -y, +z, -z
//   Window<2>      double, the 6 x 6 x 6 cells of one 4 x 4 x 4 tile and its halo (TILE order, engine.h): 33 KB, four
//                  workgroups per CU; every cell a particle of the tile can reach in one step is in it, diagonals included.
template <int WIN> struct Window;
template <> struct Window<0> { typedef float acc_t;  static constexpr bool TILE = false; static constexpr int WX = 62, NSLOT = NSEG * WX, NSLOT_PAD = NSLOT + 1, DRAIN_BLOCK = 64; };
template <> struct Window<1> { typedef double acc_t; static constexpr bool TILE = false; static constexpr int WX = 42, NSLOT = NSEG * WX, NSLOT_PAD = NSLOT + 1, DRAIN_BLOCK = 8; };
#ifndef VPIC_HIP_TILE_DRAIN_BLOCK
#define VPIC_HIP_TILE_DRAIN_BLOCK 8
#endif
#ifndef VPIC_HIP_TILE_ACC
#define VPIC_HIP_TILE_ACC double   // (float, re-measured in round 3: -17 % at 64 ppc, -30 % at 32, -50 % hot: ds_add_f32 is paid per live lane)
#endif
template <> struct Window<2> { typedef VPIC_HIP_TILE_ACC acc_t; static constexpr bool TILE = true;  static constexpr int WX = TILE_EDGE + 2, NSLOT = WX * WX * WX, NSLOT_PAD = NSLOT + 1, DRAIN_BLOCK = VPIC_HIP_TILE_DRAIN_BLOCK; };
// the same for a species sorted by tile only (no runs of equal cells to sum: every lane adds for itself, no regrouping: +4 %)
#ifndef VPIC_HIP_UNORDERED_DRAIN_BLOCK
#define VPIC_HIP_UNORDERED_DRAIN_BLOCK 8
#endif
template <> struct Window<3> : Window<2> { static constexpr int DRAIN_BLOCK = VPIC_HIP_UNORDERED_DRAIN_BLOCK; };
// DETERMINISTIC accumulation (vpic_hip_set_accumulation): every deposit is rounded to a fixed-point number (64 bits, a
// power-of-two scale chosen from the macro-particle charge) and summed as an INTEGER -- in the lanes' own LDS atomics, in
// the window, in the global flush -- so the sums do not depend on the order of anything: array order, wavefront
// scheduling, atomics.  The reference gets there by construction (private accumulators reduced in a fixed order,
// sf_interface/reduce_accumulators.cxx:37-55).  No scan (a float scan would round in lane order) and no regrouping:
//   Window<4>  the tile window in unsigned 64-bit words; Window<5> the reference's order without a window (every
//              deposit is a global 64-bit atomic: slow, for species the tile order does not serve).
template <> struct Window<4> : Window<2> { typedef unsigned long long acc_t; static constexpr int DRAIN_BLOCK = 1; };
template <> struct Window<5> { typedef unsigned long long acc_t; static constexpr bool TILE = false; static constexpr int WX = 2, NSLOT = NSEG * WX, NSLOT_PAD = NSLOT + 1, DRAIN_BLOCK = 1; };
//   Window<6>  Window<4> for a species sorted by tile only: no runs of equal cells to sum, every lane adds its own deposits
//              (round 4 gave Window<4> integer run sums; a hot species keeps what rounds 2-3 did)
template <> struct Window<6> : Window<4> {};
template <class W> struct is_det { static constexpr bool value = false; };
template <> struct is_det<Window<6>> { static constexpr bool value = true; };
template <> struct is_det<Window<4>> { static constexpr bool value = true; };
template <> struct is_det<Window<5>> { static constexpr bool value = true; };
struct TileDiv { unsigned mul_sy, sh_sy, mul_sz, sh_sz; double scale; };    // magic_div of the voxel strides (engine.h); fixed-point scale (deterministic mode)

// float -> fixed point, round to nearest: adding 1.5 x 2^52 leaves round(x * scale) in the low bits of the double's
// significand (two's complement relative to the constant); |x * scale| < 2^51.
__device__ __forceinline__ unsigned long long to_fixed(float x, double scale) {
  const double d = __builtin_fma((double)x, scale, 6755399441055744.0);
  return (unsigned long long)__double_as_longlong(d) - 0x4338000000000000ull;
}
#ifndef VPIC_HIP_MAIN_BLOCK
#define VPIC_HIP_MAIN_BLOCK 64
#endif
#ifndef VPIC_HIP_TILE_MAIN_BLOCK
#define VPIC_HIP_TILE_MAIN_BLOCK 16
#endif
constexpr int TILE_MAIN_BLOCK = VPIC_HIP_TILE_MAIN_BLOCK;   // the same for the tile window (double-precision atomics cost the same whatever the number of run tails)
constexpr int MAIN_BLOCK = VPIC_HIP_MAIN_BLOCK;     // lanes over which the segmented scan sums a run before the LDS atomics (main pass)
constexpr int MAX_GROUP_ITERS = 6;
constexpr int MIN_GROUP = 3;



// ---- accumulator window ------------------------------------------------------------------------
template <int WX>
__device__ __forceinline__ int window_slot(int key, int wbase, int sy, int sz) {
  unsigned o;
  o = (unsigned)(key - wbase);        if (o < (unsigned)WX) return (int)o;
  o = (unsigned)(key - wbase - sy);   if (o < (unsigned)WX) return WX + (int)o;
  o = (unsigned)(key - wbase + sy);   if (o < (unsigned)WX) return 2 * WX + (int)o;
  o = (unsigned)(key - wbase - sz);   if (o < (unsigned)WX) return 3 * WX + (int)o;
  o = (unsigned)(key - wbase + sz);   if (o < (unsigned)WX) return 4 * WX + (int)o;
  return -1;
}

// slot of voxel `key` in the workgroup's window, or -1.  Row windows: wbase = first voxel of the own-row segment.  Tile
// window: wbase = voxel of the block's first cell (one cell before the tile on every axis); key - wbase is taken apart
// into (lx, ly, lz) with the strides' magic numbers, so slot -> voxel (the flush) is the exact inverse of voxel -> slot.
template <class W>
__device__ __forceinline__ int slot_of(int key, int wbase, int sy, int sz, const TileDiv &td) {
  if constexpr (W::TILE) {
    const unsigned rel = (unsigned)(key - wbase);
    const unsigned lz = __umulhi(rel, td.mul_sz) >> td.sh_sz, r2 = rel - lz * (unsigned)sz;
    const unsigned ly = __umulhi(r2, td.mul_sy) >> td.sh_sy, lx = r2 - ly * (unsigned)sy;
    const bool in = wbase != NO_WINDOW && (int)rel >= 0 && lz < (unsigned)W::WX && ly < (unsigned)W::WX && lx < (unsigned)W::WX;
    return in ? (int)(lx + (unsigned)W::WX * (ly + (unsigned)W::WX * lz)) : -1;
  } else {
    return window_slot<W::WX>(key, wbase, sy, sz);
  }
}

template <bool USE_LDS = true, class W = Window<false>>
__device__ __forceinline__ void deposit12(typename W::acc_t *s_acc, float *g_acc, int key, int slot, const float *v, double det_scale = 0) {
  typedef typename W::acc_t acc_t;
  constexpr int NSLOT_PAD = W::NSLOT_PAD;
  if constexpr (is_det<W>::value) {
    unsigned long long *g64 = reinterpret_cast<unsigned long long *>(g_acc) + (size_t)key * 12;
    if (USE_LDS && slot >= 0) {
#pragma unroll
      for (int k = 0; k < 12; k++) atomicAdd(&s_acc[k * NSLOT_PAD + slot], to_fixed(v[k], det_scale));   // ds_add_u64
      asm volatile("" ::: "memory");
    } else {
#pragma unroll
      for (int k = 0; k < 12; k++) atomicAdd(&g64[k], to_fixed(v[k], det_scale));                          // global_atomic_add_x2
    }
    return;
  }
  if (det_scale != 0) {                              // a caller without a window (the injection path) on a deterministic engine
    unsigned long long *g64 = reinterpret_cast<unsigned long long *>(g_acc) + (size_t)key * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) atomicAdd(&g64[k], to_fixed(v[k], det_scale));
    return;
  }
  if (USE_LDS && slot >= 0) {
#pragma unroll
    for (int k = 0; k < 12; k++) atomicAdd(&s_acc[k * NSLOT_PAD + slot], (acc_t)v[k]);   // ds_add_f32 / ds_add_f64
    // keeps the compiler from merging the last atomic of the two branches into one FLAT atomic on a selected
    // pointer: a pending FLAT operation turns every later s_waitcnt of the push loop into vmcnt(0)
    asm volatile("" ::: "memory");
  } else {
    float *a = g_acc + (size_t)key * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) atomicAdd(&a[k], v[k]);                          // global_atomic_add_f32
  }
}

// The 12 quarter-face contributions of one straight streak: midpoint (dx,dy,dz), half
// displacement (ux,uy,uz) in cell units, charge q.  advance_p.cxx:131-162 / move_p.c:76-100.
__device__ __forceinline__ void streak12(float *a, float q, float dx, float dy, float dz,
                                         float ux, float uy, float uz, float v5) {
  const float one = 1.f;
  float v0, v1, v2, v3, v4;
#define ACCUMULATE_J(X, Y, Z, off)     \
  v4 = q * u##X;                       \
  v1 = v4 * d##Y;                      \
  v0 = v4 - v1;                        \
  v1 += v4;                            \
  v4 = one + d##Z;                     \
  v2 = v0 * v4;                        \
  v3 = v1 * v4;                        \
  v4 = one - d##Z;                     \
  v0 *= v4;                            \
  v1 *= v4;                            \
  v0 += v5;                            \
  v1 -= v5;                            \
  v2 -= v5;                            \
  v3 += v5;                            \
  a[off + 0] = v0; a[off + 1] = v1; a[off + 2] = v2; a[off + 3] = v3
  ACCUMULATE_J(x, y, z, 0);
  ACCUMULATE_J(y, z, x, 4);
  ACCUMULATE_J(z, x, y, 8);
#undef ACCUMULATE_J
}

// move_p.c:34-134 for one lane.  Returns 1 when the particle stopped on a face this domain
// cannot handle (absorbing face or another domain's), with the remaining displacement in disp.
template <bool USE_LDS = true, class W = Window<false>>
__device__ __forceinline__ int move_p_lane(float &pdx, float &pdy, float &pdz, int &pi,
                                           float &pux, float &puy, float &puz, const float q,
                                           float &dispx, float &dispy, float &dispz,
                                           typename W::acc_t *s_acc, float *g_acc, int wbase, const GridK &g, const bool no_deposit = false,
                                           const double det_scale = 0) {
  for (;;) {
    float s_midx = pdx, s_midy = pdy, s_midz = pdz;
    float s_dispx = dispx, s_dispy = dispy, s_dispz = dispz;
    const float s_dir0 = (s_dispx > 0) ? 1.f : -1.f;
    const float s_dir1 = (s_dispy > 0) ? 1.f : -1.f;
    const float s_dir2 = (s_dispz > 0) ? 1.f : -1.f;
    const float big = (float)3.4e38;
    float v0 = (s_dispx == 0) ? big : (s_dir0 - s_midx) / s_dispx;
    float v1 = (s_dispy == 0) ? big : (s_dir1 - s_midy) / s_dispy;
    float v2 = (s_dispz == 0) ? big : (s_dir2 - s_midz) / s_dispz;
    float v3 = 2.f;
    int type = 3;
    if (v0 < v3) { v3 = v0; type = 0; }
    if (v1 < v3) { v3 = v1; type = 1; }
    if (v2 < v3) { v3 = v2; type = 2; }
    v3 *= 0.5f;

    s_dispx *= v3; s_dispy *= v3; s_dispz *= v3;
    s_midx += s_dispx; s_midy += s_dispy; s_midz += s_dispz;

    // move_p.c:76: the 1/3 is a double constant there, so the last multiply is done in double
    const float v5 = (float)((double)(q * s_dispx * s_dispy * s_dispz) * (1. / 3.));
    float a[12];
    streak12(a, q, s_midx, s_midy, s_midz, s_dispx, s_dispy, s_dispz, v5);
    if (!no_deposit) deposit12<USE_LDS, W>(s_acc, g_acc, pi, USE_LDS ? window_slot<W::WX>(pi, wbase, g.sy, g.sz) : -1, a, det_scale);

    dispx -= s_dispx; dispy -= s_dispy; dispz -= s_dispz;
    pdx += s_dispx + s_dispx; pdy += s_dispy + s_dispy; pdz += s_dispz + s_dispz;

    if (type == 3) return 0;

    // neighbor[6*i + face] of move_p.c:123, generated from the per-face codes (ops.c:74-97)
    const float dir = (type == 0) ? s_dir0 : (type == 1) ? s_dir1 : s_dir2;
    const int up = dir > 0;
    const int cz = pi / g.sz, rem = pi - cz * g.sz, cy = rem / g.sy, cx = rem - cy * g.sy;
    const int c = (type == 0) ? cx : (type == 1) ? cy : cz;
    const int n = (type == 0) ? g.nx : (type == 1) ? g.ny : g.nz;
    const int stride = (type == 0) ? 1 : (type == 1) ? g.sy : g.sz;
    const int at_edge = up ? (c == n) : (c == 1);
    const int code = pbc_of(g, (up ? 3 : 0) + type);
    if (at_edge && code != g.rank) {
      if (type == 0) pdx = dir; else if (type == 1) pdy = dir; else pdz = dir;
      if (code != VPIC_REFLECT_PARTICLES) return 1;
      if (type == 0) { pux = -pux; dispx = -dispx; }
      else if (type == 1) { puy = -puy; dispy = -dispy; }
      else { puz = -puz; dispz = -dispz; }
    } else {
      if (at_edge) pi += (up ? -(n - 1) : (n - 1)) * stride;   // periodic onto this same domain
      else         pi += up ? stride : -stride;
      if (type == 0) pdx = -dir; else if (type == 1) pdy = -dir; else pdz = -dir;
    }
  }
}

}  // namespace vpichip

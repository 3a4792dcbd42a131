// push.hip -- advance_p: relativistic Boris push + charge-conserving current deposition, and the
// cell-crossing path (move_p), for gfx950.
//
// Reference behaviour restated (arithmetic and operation order kept so that, compiled with
// -ffp-contract=off, every particle comes out bit-identical to the scalar CPU pipeline):
//   src/species_advance/standard/advance_p.cxx:68-177   per-particle push + in-cell deposit
//   src/species_advance/standard/move_p.c:34-134        streak splitting across cell faces
//   src/species_advance/standard/advance_p.cxx:399-472  host wrapper (constants, mover list)
//
// MI355X design (not the reference's pipeline structure).  Measured facts that shaped it (tools/ubench/lds_rate.hip):
// an LDS float atomic costs 3 clocks per LIVE LANE, a double-precision one 8 clocks per instruction whatever the lanes
// (plus 2 per lane that collides on an address), so the deposition must reach LDS already summed over the lanes that
// share a cell; a wavefront that executes the cell-crossing loop with a few live lanes costs as much as one with 64;
// and the kernel is bound by VALU issue (90 % busy), not by HBM.  Hence:
//   * particles are struct-of-arrays and grouped by 4x4x4-cell TILE, cell by cell within a tile (the engine's sort
//     order, engine.h; the reference's order by voxel is served by row windows, see Window<> in push_device.h);
//     a 256-thread workgroup owns one tile's particles, each wavefront a quarter of them, 64 per pass;
//   * every pass a wavefront regroups its lanes by cell (ballot per distinct cell, rank by mbcnt, ds_permute): a few
//     steps after a sort the particles of neighbouring cells interleave in the array, and the regrouping turns them
//     back into a handful of runs of equal cells.  The permutation stays inside a 256-byte window of each array, so
//     loads and stores remain coalesced;
//   * the 12 accumulator components are summed over each run with a segmented DPP scan in registers (v_fmac_f32 with
//     a DPP source per component and step) and only the last lane of a run issues LDS atomics;
//   * the accumulators of the tile and its halo (6x6x6 cells: every cell a particle of the tile can reach in a step,
//     and for several steps after the sort) live in LDS in double precision and are flushed once per workgroup with
//     coalesced global float atomics; a deposit outside the window goes to the global accumulator directly;
//   * cell-crossers are queued per wavefront in LDS (36 bytes each) and finished 64 at a time in a wave-synchronous,
//     branch-free restatement of the move_p loop whose per-segment deposits go through the same scan;
//   * the accumulator array is a single copy: there is no per-pipeline replica to reduce.
// No MFMA: there is no dense contraction on this path.  HBM traffic by design: 32 B read + 24 B written per particle
// (i and q are not rewritten for in-cell particles).
//
// Build note: -fno-slp-vectorize is required.  With the SLP vectorizer on, hipcc (ROCm 7.2) packs
// pairs of these fp32 operations into v_pk_mul_f32/v_pk_add_f32 and the kernel returns wrong
// momenta/positions on gfx950 (caught by the golden-vector tests).
#include "push_device.h"
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <vector>
#include <algorithm>

namespace vpichip {

// What every pass of the push loop reads stays in the kernel argument (scalar registers for the
// whole kernel).  What only the cell-crossing path reads (DrainParams, engine.h) lives in device
// memory and is fetched where it is used: kept in the argument it would occupy ~25 scalar
// registers across the hot loop, and the loop then runs out of them (the compiler spills scalars
// into lanes of a VGPR and pays a v_readlane plus wait states at every use).
struct PushParams {
  float qdt_2mc, cdt_dx, cdt_dy, cdt_dz;
  int np;
  int iters;    // passes of 64 particles per wavefront: 256*iters particles per workgroup, chosen so that a chunk spans <= ~64 cells
  int sy, sz;   // voxel strides of the grid
  int idx_base;        // index of this launch's first particle in the species (a species beyond 2^30 particles is pushed in segments)
  unsigned *crossed;   // device counter: particles that left their cell in this launch
  // TILE order only (Window<2>; engine.h): where every cell of every tile began at the last sort, the tile grid, the
  // particles sorted then (those behind are pushed by extra workgroups without a window), the strides' magic numbers
  const int *tpart, *ttail; int ntx, nty, ntiles, n_sorted; unsigned mul_sy, sh_sy, mul_sz, sh_sz;   // ttail: the appended particles' ranges by tile, relative to n_sorted (null: not regrouped)
  // ... and which tiles this launch pushes (vpic_hip_advance_p_phase: the tiles on the domain's shared faces first, the
  // others behind the start of the exchange): workgroup b < n_launch takes tile tile_list[b] (null: tile b), the next
  // tail_chunks workgroups the appended particles that were not regrouped
  const int *tile_list; int n_launch, tail_chunks;
  int nx, ny, nz;      // cells per axis (TILE: where a window may lie, see the window's place in advance_p_kernel)
  int follow;          // TILE: the window follows the tile's particles: 1 always, 0 never, -1 when the device's flag says so -- word 2 of
                       // `crossed`, set behind the first launch whose deposits began to miss the windows, cleared by the sort (in stream order)
  double acc_scale;    // deterministic accumulation (Window<4>, Window<5>): the fixed-point scale
  int *hist; int ntz;  // HIST instances: the next sort's counts by tile-order key (Species::hist, engine.h)
  // SORT instance (Species::fuse_pending, engine.h): the particles leave for the second buffer, each to the next free place
  // of its cell (the cell it was in BEFORE this push) in the new order
  ParticlesK out; int *next;
  int stage;           // STAGE instances: 1 = the positions of a pass wait in registers for its cell-crossers (see Staged below); 0 = stored at once
#ifdef VPIC_HIP_ABLATION
  int ablate;   // timing experiments only (builds with -DVPIC_HIP_ABLATION: VPIC_HIP_ABLATE; tools/ablate.sh): 1 no in-cell deposit, 2 no mover path, 4 no interpolator gather, 8 no flush, 16 no lane regrouping, 32 no mover deposit, 64 no drain, 128 no in-cell stores
#endif
};

// ---- segmented wavefront scan with DPP ---------------------------------------------------------
// Lanes holding consecutive particles of the same cell form a run.  An inclusive segmented scan
// (Kogge-Stone inside each row of 16 lanes with row_shr, then row_bcast:15 / row_bcast:31 across
// rows) leaves each run's total in its last lane.  One step for one value is a single
// v_fmac_f32 with a DPP source: v += dpp(v) * flag, flag in {0,1}; fma(t,1,v) rounds once, like
// the add it stands for.  The 12 accumulator components are independent, so issuing the same step
// for all 12 back to back also covers the DPP read-after-write wait states.
#define SEG_STEP(v, flag, CTRL)                                                                   \
  asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL : "+v"(v) : "v"(flag))
#define SEG_STEP12(a, flag, CTRL)                                                                 \
  do {                                                                                            \
    asm volatile("s_nop 1");                                                                      \
    SEG_STEP(a[0], flag, CTRL); SEG_STEP(a[1], flag, CTRL); SEG_STEP(a[2], flag, CTRL);           \
    SEG_STEP(a[3], flag, CTRL); SEG_STEP(a[4], flag, CTRL); SEG_STEP(a[5], flag, CTRL);           \
    SEG_STEP(a[6], flag, CTRL); SEG_STEP(a[7], flag, CTRL); SEG_STEP(a[8], flag, CTRL);           \
    SEG_STEP(a[9], flag, CTRL); SEG_STEP(a[10], flag, CTRL); SEG_STEP(a[11], flag, CTRL);         \
  } while (0)

// Particle arrays are addressed as (uniform base pointer in SGPRs) + (32-bit byte offset in one
// VGPR): every access then uses the scalar-base form of global_load/global_store and the eight
// arrays share one offset register instead of eight 64-bit address computations.
__device__ __forceinline__ float ldf(const float *base, unsigned off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off); }
__device__ __forceinline__ int ldi(const int *base, unsigned off) { return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(base) + off); }
__device__ __forceinline__ void stf(float *base, unsigned off, float v) { *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) = v; }
__device__ __forceinline__ void sti(int *base, unsigned off, int v) { *reinterpret_cast<int *>(reinterpret_cast<char *>(base) + off) = v; }
// NON-TEMPORAL stores (round 4) for what the launch writes once and never touches again: the momenta of the main pass and the
// crossers' late stores (after them the line is complete).  Their lines then leave L2 first, and the position lines that the late
// stores will hit stay longer.  Same-box A/B at 256^3 x 64 ppc, three script runs: momenta +0.3-0.7 %, with the late stores
// +0.5-1.3 % on the launch (16.41 -> 16.33, 16.50 -> 16.30, 16.35 -> 16.16 ms).  Measured and NOT adopted: non-temporal particle
// LOADS (-3 %: 17.25 against 16.75 ms), non-temporal position stores of the main pass (level), non-temporal stores of the launch
// that sorts as it pushes (45 ms against 26: its runs end inside sectors, and the pieces no longer meet in L2), the momenta of
// the hot instances whose positions wait in registers (level: 1.41-1.43 ms either way on the configs[3] slab).
__device__ __forceinline__ void stf_nt(float *base, unsigned off, float v) { __builtin_nontemporal_store(v, reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off)); }
__device__ __forceinline__ void sti_nt(int *base, unsigned off, int v) { __builtin_nontemporal_store(v, reinterpret_cast<int *>(reinterpret_cast<char *>(base) + off)); }


// A wave-uniform value parked in a VECTOR register.  The pass loop needs more scalars than the 102 SGPRs hold; what the
// compiler then spills into lanes of a VGPR comes back through a v_readlane at every use (16 of them per pass for the
// strides, their magic numbers and np - 1: 5 % of the pass's vector instructions).  A value that only ever feeds vector
// instructions costs nothing as a VGPR operand: the opaque move keeps the compiler from turning it back into a scalar.
__device__ __forceinline__ int in_vgpr(int s) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }

__device__ __forceinline__ int mbcnt64(unsigned long long m) {      // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

#ifndef VPIC_HIP_MAX_KEYS
#define VPIC_HIP_MAX_KEYS 12
#endif
constexpr int MAX_KEYS = VPIC_HIP_MAX_KEYS;     // distinct cells per wavefront that get a group of their own

// Destination lane of each lane such that equal keys become consecutive (stable, groups in order
// of first appearance; keys beyond MAX_KEYS distinct ones keep their relative order at the end).
// Scalar work per distinct key plus two VALU ops.  All 64 lanes must call.
__device__ __forceinline__ int group_lanes_by_key(int key, int lane) {
  unsigned long long todo = ~0ull;
  int dest = lane, offset = 0;
  for (int it = 0; todo && it < MAX_KEYS; ++it) {
    const int lead = __ffsll((long long)todo) - 1;
    const int k0 = __builtin_amdgcn_readlane(key, lead);
    const bool mine = (key == k0);
    const unsigned long long m = __ballot(mine);
    if (mine) dest = offset + mbcnt64(m);
    offset += __popcll(m);
    todo &= ~m;
  }
  if ((todo >> lane) & 1ull) dest = offset + mbcnt64(todo);
  return dest;
}

// Sum a[0..11] over each run of equal keys; the last lane of each run with key >= 0 adds the run's
// totals to accumulator `key`.  All 64 lanes must call.  BLOCK < 64 additionally ends every run at
// the multiples of BLOCK lanes: log2(BLOCK) scan steps instead of 6, at the price of one more
// group of 12 LDS atomics for every block boundary that falls inside a run.
//
// A run whose cell lies outside the window has to go to the global accumulator: twelve float atomics.  Issued on the
// spot they sit in the wavefront's in-order vector-memory queue in front of the next pass's loads, and a pass that waits
// for its interpolators waits for them too (measured: 1 % of the particles outside the window cost the launch 10 %).
// So such totals are parked in a small per-wavefront list in LDS and flushed together -- MISS_CAP runs x 12 values by one
// atomic instruction -- when the list is full and when the wavefront is done.
constexpr int MISS_CAP = 5;                                   // 5 x 12 = 60 lanes of one atomic instruction
struct MissList { int key[MISS_CAP]; float v[MISS_CAP][12]; int total; };   // 264 bytes per wavefront; total: runs that missed the window so far (the host reads the launch's sum: see PushParams::follow)

__device__ __forceinline__ void flush_misses(MissList *ml, int &n_miss, float *g_acc, int lane) {
  if (n_miss > 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int entry = lane / 12, k = lane - entry * 12;
    if (entry < n_miss) atomicAdd(g_acc + (size_t)ml->key[entry] * 12 + k, ml->v[entry][k]);
    if (lane == 0) ml->total += n_miss;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    n_miss = 0;
  }
}

// HIST: where a particle ends the step is counted for the sort that follows: two 16-bit counters per LDS word, one per window
// cell (a cell of a tile whose fullest tile holds fewer than 2^15 particles cannot overflow one); outside the window straight
// into the global table.
struct HistK { int *hist; unsigned *s_cnt; TileK tk; };
template <class W>
__device__ __forceinline__ void hist_count(const HistK &h, int key, int wbase, int sy, int sz, const TileDiv &td) {
  const int slot = slot_of<W>(key, wbase, sy, sz, td);
  if (slot >= 0) atomicAdd(&h.s_cnt[slot >> 1], 1u << ((slot & 1) << 4));
  else atomicAdd(&h.hist[sort_key<true>(key, h.tk)], 1);
}

template <class W>
__device__ __forceinline__ void deposit_run(const bool tail, const float (&a)[12], int key, int lane, typename W::acc_t *s_acc, float *g_acc,
                                            int wbase, int sy, int sz, const TileDiv &td, MissList *ml, int &n_miss, const int pre_slot = -2) {
  const int slot = pre_slot != -2 ? pre_slot : slot_of<W>(key, wbase, sy, sz, td);   // (pre_slot: the caller has it already)
  if constexpr (is_det<W>::value) {                 // (integer sums: a miss goes straight to the global words, in any order)
    if (tail) deposit12<true, W>(s_acc, g_acc, key, slot, a, td.scale);
    return;
  }
  if (tail && slot >= 0) deposit12<true, W>(s_acc, g_acc, key, slot, a);
  const bool miss = tail && slot < 0;
  const unsigned long long mm = __ballot(miss);
  if (mm) {                                                    // wave-uniform, rare
    const int cnt = __popcll(mm);
    if (cnt > MISS_CAP) {                                      // more than the list holds: on the spot
      if (miss) deposit12<true, W>(s_acc, g_acc, key, -1, a);
      if (lane == 0) ml->total += cnt;
      } else {
      if (n_miss + cnt > MISS_CAP) flush_misses(ml, n_miss, g_acc, lane);
      if (miss) {
        const int e = n_miss + mbcnt64(mm);
        ml->key[e] = key;
#pragma unroll
        for (int k = 0; k < 12; k++) ml->v[e][k] = a[k];
      }
      n_miss += cnt;
    }
  }
}

template <int BLOCK, class W>
__device__ __forceinline__ void run_deposit(float (&a)[12], int key, int lane, typename W::acc_t *s_acc, float *g_acc,
                                            int wbase, int sy, int sz, const TileDiv &td, MissList *ml, int &n_miss,
                                            const HistK *hk = nullptr, const bool counted = false, const int given_slot = -2) {
  static_assert(BLOCK == 1 || BLOCK == 2 || BLOCK == 4 || BLOCK == 8 || BLOCK == 16 || BLOCK == 64, "scan width");
  if (BLOCK == 1) {                                                // no scan at all: every lane adds for itself
    deposit_run<W>(key >= 0, a, key, lane, s_acc, g_acc, wbase, sy, sz, td, ml, n_miss);
    return;
  }
  const int prev = __builtin_amdgcn_update_dpp(-2, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const unsigned long long block_heads = BLOCK == 64 ? 0ull : BLOCK == 16 ? 0x0001000100010001ull : BLOCK == 8 ? 0x0101010101010101ull
                                       : BLOCK == 4 ? 0x1111111111111111ull : 0x5555555555555555ull;
  const unsigned long long heads = __ballot(prev != key) | block_heads;   // lane 0 reads old = -2: always a head
  const unsigned long long below = heads & ((2ull << lane) - 1ull);
  const int d = lane - (63 - __clzll((long long)below));           // distance from the run's first lane
  const float f1 = d >= 1 ? 1.f : 0.f;
  SEG_STEP12(a, f1, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  if (BLOCK >= 4) {
    const float f2 = d >= 2 ? 1.f : 0.f;
    SEG_STEP12(a, f2, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  }
  if (BLOCK >= 8) {
    const float f4 = d >= 4 ? 1.f : 0.f;
    SEG_STEP12(a, f4, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  }
  if (BLOCK >= 16) {
    const float f8 = d >= 8 ? 1.f : 0.f;
    SEG_STEP12(a, f8, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  }
  if (BLOCK == 64) {
    const int r = lane & 15;
    const float f16 = d > r ? 1.f : 0.f;                           // run began before this row
    const float f32 = d > (lane & 31) ? 1.f : 0.f;                 // run began before lane 32 (rows 2,3)
    SEG_STEP12(a, f16, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    SEG_STEP12(a, f32, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  }
  asm volatile("s_nop 1");
  const bool tail = (lane == 63) || ((heads >> ((lane + 1) & 63)) & 1ull);
  if (hk) {
    // HIST: the run's last lane also counts the run's particles that stay in the cell (`counted` lanes) for the sort that
    // follows -- one LDS atomic per run on a word of its own instead of one per lane on a few words
    const int slot = slot_of<W>(key, wbase, sy, sz, td);
    const unsigned long long inc = __ballot(counted);
    const unsigned long long run = ((2ull << lane) - 1ull) & ~((1ull << (lane - d)) - 1ull);
    const int n = __popcll(inc & run);
    if (tail && key >= 0 && n) {
      if (slot >= 0) atomicAdd(&hk->s_cnt[slot >> 1], (unsigned)n << ((slot & 1) << 4));
      else atomicAdd(&hk->hist[sort_key<true>(key, hk->tk)], n);
    }
    deposit_run<W>(tail && key >= 0, a, key, lane, s_acc, g_acc, wbase, sy, sz, td, ml, n_miss, slot);
    return;
  }
  deposit_run<W>(tail && key >= 0, a, key, lane, s_acc, g_acc, wbase, sy, sz, td, ml, n_miss, given_slot);   // (given_slot: the caller has the key's window slot already)
}

// DETERMINISTIC accumulation, the main pass of the tile instance (round 4).  Rounds 2-3 had every lane add its twelve deposits to
// the window by itself (64-bit fixed point, ds_add_u64): on a cell-sorted species the 64 lanes of a pass hit two or three
// cells -- twenty-fold collisions on every one of the twelve atomics, a launch twice as long as the float mode's.  Integer sums
// do not care about their order, so the float mode's structure carries over: the lanes regrouped by cell, a segmented scan over
// each run of equal cells, the run's last lane adding its total.  The scan runs on 32-BIT fixed point (a third of the
// instructions of a 64-bit one): a deposit is rounded to 2^-26 of the largest one a particle of the reference charge makes
// (4.2 q_ref -- the 64-bit conversion keeps 2^-37; the float mode's own sums carry 2^-24 of their size), a run of sixteen stays
// below 2^30.  A pass in which some deposit is too large for that (a macro-particle of more than 1.9 reference charges) adds lane
// by lane as before: the same integers either way, so the sums do not depend on which path a pass took.
constexpr int DET_SHIFT = 11;                                    // 32-bit sums are 2^11 coarser than the window's 64-bit words
template <int BLOCK, class W>
__device__ __forceinline__ void run_deposit_fixed(const float (&a)[12], int key, int lane, typename W::acc_t *s_acc, float *g_acc,
                                                  int wbase, int sy, int sz, const TileDiv &td, MissList *ml, int &n_miss) {
  static_assert(BLOCK == 16, "runs of at most 16 lanes: 16 x 2^26 < 2^31");
  const float s32 = (float)(td.scale * (1.0 / (1 << DET_SHIFT)));          // a power of two: the product below is exact
  float big = fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))), fmaxf(fmaxf(fabsf(a[4]), fabsf(a[5])), fmaxf(fabsf(a[6]), fabsf(a[7]))));
  big = fmaxf(big, fmaxf(fmaxf(fabsf(a[8]), fabsf(a[9])), fmaxf(fabsf(a[10]), fabsf(a[11]))));
  if (__ballot(key >= 0 && !(big * s32 < 67108864.f))) {       // (wave-uniform, rare; NaNs take this path too)
    // every lane for itself, rounded the same way: round(a * 2^k) * 2^11
    if (key >= 0) {
      const int slot = slot_of<W>(key, wbase, sy, sz, td);
      unsigned long long *g64 = reinterpret_cast<unsigned long long *>(g_acc) + (size_t)key * 12;
#pragma unroll
      for (int k = 0; k < 12; k++) {
        const unsigned long long v = (unsigned long long)((long long)__builtin_rint((double)a[k] * (double)s32) << DET_SHIFT);
        if (slot >= 0) atomicAdd(&s_acc[k * W::NSLOT_PAD + slot], v); else atomicAdd(&g64[k], v);
      }
    }
    asm volatile("" ::: "memory");
    return;
  }
  int v[12];
#pragma unroll
  for (int k = 0; k < 12; k++) v[k] = (int)__builtin_rintf(a[k] * s32);
  const int prev = __builtin_amdgcn_update_dpp(-2, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const unsigned long long heads = __ballot(prev != key) | 0x0001000100010001ull;    // (rows of 16 lanes)
  const unsigned long long below = heads & ((2ull << lane) - 1ull);
  const int d = lane - (63 - __clzll((long long)below));           // distance from the run's first lane
#pragma unroll
  for (int step = 1; step < 16; step <<= 1) {
    const bool on = d >= step;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const int t = step == 1 ? __builtin_amdgcn_update_dpp(0, v[k], 0x111 /* row_shr:1 */, 0xf, 0xf, true)
                  : step == 2 ? __builtin_amdgcn_update_dpp(0, v[k], 0x112, 0xf, 0xf, true)
                  : step == 4 ? __builtin_amdgcn_update_dpp(0, v[k], 0x114, 0xf, 0xf, true)
                              : __builtin_amdgcn_update_dpp(0, v[k], 0x118, 0xf, 0xf, true);
      v[k] += on ? t : 0;
    }
  }
  const bool tail = key >= 0 && ((lane == 63) || ((heads >> ((lane + 1) & 63)) & 1ull));
  if (tail) {
    const int slot = slot_of<W>(key, wbase, sy, sz, td);
    unsigned long long *g64 = reinterpret_cast<unsigned long long *>(g_acc) + (size_t)key * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const unsigned long long w = (unsigned long long)((long long)v[k] << DET_SHIFT);
      if (slot >= 0) atomicAdd(&s_acc[k * W::NSLOT_PAD + slot], w); else atomicAdd(&g64[k], w);
    }
  }
  asm volatile("" ::: "memory");
}

constexpr int WAVES = PUSH_THREADS / 64;

#ifndef VPIC_HIP_MQW
#define VPIC_HIP_MQW 64
#endif
static_assert(VPIC_HIP_MQW >= 64, "a pass can yield 64 cell-crossers: the queue must hold them once it has been drained");
constexpr int MQW = VPIC_HIP_MQW;                    // per-wavefront queue of cell-crossers: a pass that would overflow it drains first
                                             // (any crosser fraction is safe).  Exactly one wavefront's worth (round 3, A/B on one
                                             // box: 64 is 4-5 % faster than 72 or 96 -- the queue is drained whole, nothing is left
                                             // behind to be moved to its front -- at 32 and at 64 particles per cell)
// A queued cell-crosser carries the state its move needs -- position and voxel, remaining displacement and particle
// index, charge: 36 bytes -- so finishing it needs no second trip to HBM for the particle arrays (re-reading them cost
// about one extra read of the whole species per step).  Its momentum stays where the pass that queued it stored it: only
// a reflection touches it, by negating the component in place (rare).  Which components a reflection has flipped so far
// rides in bits 28-30 of the voxel word (voxel indices are below 2^28: 12 nv < 2^31, engine.hip).
struct WaveQueue { float4 pos_i[MQW]; float4 disp_idx[MQW]; float q[MQW]; };   // (dx,dy,dz,i) (dispx,dispy,dispz,idx); ints bit-cast
constexpr int FLIP_SHIFT = 28;

// Finish n_mq queued cell-crossers of this wavefront (move_p.c:34-134): 64 at a time, one lane
// each, every pass of the loop body executed by the whole wavefront so that the deposits of a
// pass can be summed per cell before they touch LDS.
//
// max_round bounds the rounds of one call (n_mq <= 64 then).  A round is a full pass of the loop body (the face search:
// the crosser reaches its first face and hops) followed by a pass WITHOUT the search for the lanes whose remaining
// displacement ends inside the new cell -- most of them; a pass costs the same with three live lanes as with 64, so
// crossers still on their way after max_round rounds (corner cutters) go back to the front of the queue,
// mq[0..return value), and ride along with the next batch.
// FAST arithmetic (advance_p_kernel<.., FAST = true>): contracted multiply-adds, v_rsq_f32 / v_rcp_f32 (1 ulp) in
// place of the correctly rounded sqrt and divide sequences -- the choice the reference's own V4 pipelines make
// (src/util/v4/v4_sse.hxx:914-939: rsqrt / rcp estimates refined once).  Results agree with the scalar pipeline
// to a few ulp per step (tests/test_gpu_kernels.py states and checks the bound); the exact instance is the default.
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// the 12 streak terms with contracted multiply-adds (same formula as streak12, push_device.h)
__device__ __forceinline__ void streak12_fast(float *a, float q, float dx, float dy, float dz,
                                              float ux, float uy, float uz) {
  const float xp = 1.f + dx, xm = 1.f - dx, yp = 1.f + dy, ym = 1.f - dy, zp = 1.f + dz, zm = 1.f - dz;
  const float qx = q * ux, qy = q * uy, qz = q * uz;
  const float v5 = (qx * uy) * (uz * (1.f / 3.f));
  float v0, v1;
  v0 = __builtin_fmaf(-qx, dy, qx); v1 = __builtin_fmaf(qx, dy, qx);
  a[0] = __builtin_fmaf(v0, zm, v5); a[1] = __builtin_fmaf(v1, zm, -v5); a[2] = __builtin_fmaf(v0, zp, -v5); a[3] = __builtin_fmaf(v1, zp, v5);
  v0 = __builtin_fmaf(-qy, dz, qy); v1 = __builtin_fmaf(qy, dz, qy);
  a[4] = __builtin_fmaf(v0, xm, v5); a[5] = __builtin_fmaf(v1, xm, -v5); a[6] = __builtin_fmaf(v0, xp, -v5); a[7] = __builtin_fmaf(v1, xp, v5);
  v0 = __builtin_fmaf(-qz, dx, qz); v1 = __builtin_fmaf(qz, dx, qz);
  a[8] = __builtin_fmaf(v0, ym, v5); a[9] = __builtin_fmaf(v1, ym, -v5); a[10] = __builtin_fmaf(v0, yp, -v5); a[11] = __builtin_fmaf(v1, yp, v5);
}

// ---- positions that wait for their cell-crossers (round 4) ------------------------------------------------------------
// A cell-crosser's final position and cell are known only when its move is done, passes after the one that pushed it; stored
// then, they are four scattered 4-byte stores into lines the pass wrote long before -- a 64-byte write request each for
// 4 bytes.  Measured on one launch (tools/ablate_once.py): 20 % of a hot charged species' launch, 28 % of a charge-0 copy's,
// 8 % of a cold beam's.  Where half the particles cross (the hot species: sorted by tile only, pushed without regrouping, so
// lane l of a pass holds particle base + l) the queue fills every other pass anyway: the positions of up to STAGE_CAP passes
// stay in registers, the batch is finished, and the passes' positions are stored ONCE, whole spans, the crossers' final values
// among them; only the crossers' new cells remain scattered stores.  A crosser that is not done by then (it cuts a corner:
// a few per cent) falls back to the late stores.  What is wave-uniform about a waiting pass lives in LDS (the kernel is out
// of scalar registers).
constexpr int STAGE_CAP = 2;
// rounds of a batch before its passes are released (what is not done by then stores late): a second round finishes the corner
// cutters too -- worth its pass where a round is cheap (charge-0 copies: no deposits; 1.25 -> 1.12 ms per launch on the
// configs[3] slab), not where it carries two scans and 24 LDS atomics (charged: 1.72 -> 1.76)
template <bool CHARGELESS> struct StageRounds { static constexpr int value = CHARGELESS ? 2 : 1; };
struct StagePark { int base[STAGE_CAP], qb[STAGE_CAP]; unsigned cm_lo[STAGE_CAP], cm_hi[STAGE_CAP], act_lo[STAGE_CAP], act_hi[STAGE_CAP]; };
constexpr int STAGED_BIT = 31;                                   // of the queue entry's voxel word: the crosser's pass is waiting for it
struct Straggler { float4 pos_i, disp_idx; float q; bool live; unsigned long long again; };   // a crosser still on its way after a batch's rounds

template <bool FAST, class W, bool HIST = false, bool STAGE = false>
__device__ __forceinline__ int drain_wave(const ParticlesK &p, WaveQueue *mq, const int n_mq,
                                          const int lane, typename W::acc_t *s_acc, float *g_acc, const int wbase,
                                          const DrainParams *dp, const int ablate, const int max_round, const int idx_base,
                                          MissList *ml, int &n_miss, const double det_scale = 0, const HistK *hk = nullptr, Straggler *sg = nullptr) {
  if (ablate & 64) return 0;
  // fetched here with scalar loads the compiler cannot hoist out of the push loop (see PushParams);
  // a few dozen cycles per call.  As opaque scalars the per-axis values below also stay select
  // chains (as fields of one struct the compiler turns such selects into indexed vector loads).
  typedef int int8v __attribute__((ext_vector_type(8)));
  typedef int int4v __attribute__((ext_vector_type(4)));
  int8v d0, d1; int4v d2;
  asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dwordx4 %2, %3, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(d0), "=&s"(d1), "=&s"(d2) : "s"(dp));
  static_assert(offsetof(DrainParams, pbc) == 0x20 && offsetof(DrainParams, pm) == 0x40 && sizeof(DrainParams) == 0x50, "DrainParams layout");
  const int gnx = d0[0], gny = d0[1], gnz = d0[2], gsy = d0[3], gsz = d0[4], grank = d0[5], max_nm = d0[6];
  const unsigned mul_sz = (unsigned)d0[7], mul_sy = (unsigned)d1[6], sh_sz = (unsigned)d1[7] >> 8, sh_sy = (unsigned)d1[7] & 255u;
  const int pb0 = d1[0], pb1 = d1[1], pb2 = d1[2], pb3 = d1[3], pb4 = d1[4], pb5 = d1[5];
  const TileDiv td = {mul_sy, sh_sy, mul_sz, sh_sz, det_scale};
  // global address space stated: a generic pointer would make these FLAT instructions, and a pending FLAT operation
  // turns every later s_waitcnt of the loop into vmcnt(0) (FLAT returns out of order)
  typedef __attribute__((address_space(1))) vpic_particle_mover_t *global_mover_ptr;
  typedef __attribute__((address_space(1))) int *global_int_ptr;
  const global_mover_ptr pm = (global_mover_ptr)(((unsigned long long)(unsigned)d2[1] << 32) | (unsigned)d2[0]);
  const global_int_ptr nm_counter = (global_int_ptr)(((unsigned long long)(unsigned)d2[3] << 32) | (unsigned)d2[2]);
  int n_again = 0;
  for (int base = 0; base < n_mq; base += 64) {
    const int k = base + lane;
    bool live = k < n_mq;
    const int kq = live ? k : 0;
    const float4 c0 = mq->pos_i[kq], c2 = mq->disp_idx[kq];
    const float q = mq->q[kq];
    vpic_particle_mover_t m; m.dispx = c2.x; m.dispy = c2.y; m.dispz = c2.z; m.i = __float_as_int(c2.w);
    const int idx = m.i;
    const unsigned o4 = (unsigned)idx << 2;
    float dx = c0.x, dy = c0.y, dz = c0.z;
    int flips = (__float_as_int(c0.w) >> FLIP_SHIFT) & 7;          // momentum components negated by reflections so far (x 1, y 2, z 4)
    const bool staged = STAGE && live && (__float_as_int(c0.w) >> STAGED_BIT) & 1;   // its pass has not stored its positions yet: the result goes to the queue slot
    int pi = live ? (__float_as_int(c0.w) & ((1 << FLIP_SHIFT) - 1)) : -1, cx = 0, cy = 0, cz = 0;
    if (live) {   // voxel -> (x,y,z) by multiplication with the precomputed reciprocals (DrainParams)
      cz = (int)(__umulhi((unsigned)pi, mul_sz) >> sh_sz); const int rem = pi - cz * gsz;
      cy = (int)(__umulhi((unsigned)rem, mul_sy) >> sh_sy); cx = rem - cy * gsy;
    }
    const bool mine = live;
    bool stuck = false;
    // per-face facts as wave-uniform booleans: does the face stop a particle (it belongs to no cell
    // of this domain), and if so does it reflect
    const bool w0 = pb0 != grank, w1 = pb1 != grank, w2 = pb2 != grank, w3 = pb3 != grank, w4 = pb4 != grank, w5 = pb5 != grank;
    const bool r0 = pb0 == VPIC_REFLECT_PARTICLES, r1 = pb1 == VPIC_REFLECT_PARTICLES, r2 = pb2 == VPIC_REFLECT_PARTICLES,
               r3 = pb3 == VPIC_REFLECT_PARTICLES, r4 = pb4 == VPIC_REFLECT_PARTICLES, r5 = pb5 == VPIC_REFLECT_PARTICLES;
    for (int round = 0; round < max_round && __ballot(live); round++) {
      // One pass of the move_p.c:34-134 loop body.  Everything up to the deposit is computed by all
      // 64 lanes without a branch (a lane that is done deposits under key -1, i.e. nowhere); only
      // the state update is predicated.
      const bool up0 = m.dispx > 0, up1 = m.dispy > 0, up2 = m.dispz > 0;
      const float s_dir0 = up0 ? 1.f : -1.f, s_dir1 = up1 ? 1.f : -1.f, s_dir2 = up2 ? 1.f : -1.f;
      const float big = (float)3.4e38;
      float t0, t1, t2;
      if (FAST) {   // |disp| below 2^-100 cannot reach a face this step (and its reciprocal would overflow)
        t0 = (fabsf(m.dispx) < 7.9e-31f) ? big : (s_dir0 - dx) * fast_rcp(m.dispx);
        t1 = (fabsf(m.dispy) < 7.9e-31f) ? big : (s_dir1 - dy) * fast_rcp(m.dispy);
        t2 = (fabsf(m.dispz) < 7.9e-31f) ? big : (s_dir2 - dz) * fast_rcp(m.dispz);
      } else {
        // A quotient that is USED is below 2 with a numerator that is 0 or at least 2^-24 (positions are in [-1, 1]): both
        // operands and the result are ordinary numbers and the unscaled sequence rounds like the IEEE one.  A quotient
        // that overflows or is not a number here compares false with `< 2` exactly as the reference's huge one does.
        t0 = (m.dispx == 0) ? big : div_normal(s_dir0 - dx, m.dispx);
        t1 = (m.dispy == 0) ? big : div_normal(s_dir1 - dy, m.dispy);
        t2 = (m.dispz == 0) ? big : div_normal(s_dir2 - dz, m.dispz);
      }
      float v3 = 2.f;
      const bool lt0 = t0 < v3; v3 = lt0 ? t0 : v3;
      const bool lt1 = t1 < v3; v3 = lt1 ? t1 : v3;
      const bool lt2 = t2 < v3; v3 = lt2 ? t2 : v3;
      const bool ty2 = lt2, ty1 = lt1 && !lt2, ty0 = lt0 && !lt1 && !lt2;      // the face hit first, if any
      const bool moved_on = lt0 || lt1 || lt2;
      v3 *= 0.5f;
      const float s_dispx = m.dispx * v3, s_dispy = m.dispy * v3, s_dispz = m.dispz * v3;
      const float s_midx = dx + s_dispx, s_midy = dy + s_dispy, s_midz = dz + s_dispz;
      float a[12];
      streak12_fast(a, q, s_midx, s_midy, s_midz, s_dispx, s_dispy, s_dispz);   // (contracted in both modes: see the main pass)
      const int key = live ? pi : -1;
      // neighbor[6*i + face] of move_p.c:123, generated from the per-face codes (ops.c:74-97)
      const bool e0 = up0 ? (cx == gnx) : (cx == 1), e1 = up1 ? (cy == gny) : (cy == 1), e2 = up2 ? (cz == gnz) : (cz == 1);
      const bool hi = ty0 ? up0 : ty1 ? up1 : up2;
      const bool edge = ty0 ? e0 : ty1 ? e1 : e2;
      const bool wcode = hi ? (ty0 ? w3 : ty1 ? w4 : w5) : (ty0 ? w0 : ty1 ? w1 : w2);
      const bool rcode = hi ? (ty0 ? r3 : ty1 ? r4 : r5) : (ty0 ? r0 : ty1 ? r1 : r2);
      const bool wall = live && moved_on && edge && wcode;
      const bool refl = wall && rcode, stop = wall && !rcode;
      const bool hop = live && moved_on && !wall;
      if (live) {
        m.dispx -= s_dispx; m.dispy -= s_dispy; m.dispz -= s_dispz;
        dx += s_dispx + s_dispx; dy += s_dispy + s_dispy; dz += s_dispz + s_dispz;
        const float dir = ty0 ? s_dir0 : ty1 ? s_dir1 : s_dir2;
        const float face = hop ? -dir : dir;             // on the far side of the face after a hop, on it at a wall
        dx = ty0 ? face : dx; dy = ty1 ? face : dy; dz = ty2 ? face : dz;
        const int sgn = hi ? 1 : -1;
        const int nm1 = ty0 ? gnx - 1 : ty1 ? gny - 1 : gnz - 1;
        const int stride = ty0 ? 1 : ty1 ? gsy : gsz;
        const int dc = hop ? (edge ? -sgn * nm1 : sgn) : 0;   // wrap onto this same domain, or the next cell
        pi += dc * stride;
        cx += ty0 ? dc : 0; cy += ty1 ? dc : 0; cz += ty2 ? dc : 0;
      }
      if (__ballot(refl)) {                              // move_p.c:126-128
        if (refl && ty0) { flips ^= 1; m.dispx = -m.dispx; }
        if (refl && ty1) { flips ^= 2; m.dispy = -m.dispy; }
        if (refl && ty2) { flips ^= 4; m.dispz = -m.dispz; }
      }
      stuck = stuck || stop;
      live = hop || refl;
      if (ablate & 32) {}
      else run_deposit<W::DRAIN_BLOCK, W>(a, key, lane, s_acc, g_acc, wbase, gsy, gsz, td, ml, n_miss);
      // (Tried in round 3 and dropped, A/B on one box: a pass body without coordinates and face codes for tiles that keep
      // clear of the domain's edges -- 3 % SLOWER at 64 and at 32 particles per cell: the window test ahead of the update and
      // the second copy of the update cost more than the ~60 instructions saved.)
      // ---- the segment behind a face usually ends inside the new cell: a pass without the face search -----------------
      // move_p.c:49-63 finds no face when none of (s_dir - r) / disp is below 2: then f = 1, the segment is the whole
      // remaining displacement and the particle is done (type 3).  Whether a correctly rounded quotient n / d is below 2
      // can be told without dividing: RN(n / d) < 2  <=>  n / d < 2 - 2^-24 (the midpoint below 2 rounds to 2, whose
      // significand is even), i.e. n < K d for d > 0 and n > K d for d < 0 with K = 2 - 2^-24 -- and K d is exact in double
      // precision (25 x 24 significant bits).  Lanes for which no face is hit take this pass; a lane that WOULD hit
      // another face (a particle cutting a corner: a few per cent) waits for the next round's full pass.
      {
        // (round 3, late: the test is made in float against 2 d, which is exact, instead of in double against K d: a lane with
        // K d <= n < 2 d -- a quotient within 2^-24 of 2: one crosser in ten million -- is taken for a hit and waits for the next
        // round's full pass, which finds none.  Never the other way round: n < K d implies n < 2 d.)
        const float n0 = ((m.dispx > 0) ? 1.f : -1.f) - dx, n1 = ((m.dispy > 0) ? 1.f : -1.f) - dy, n2 = ((m.dispz > 0) ? 1.f : -1.f) - dz;
        const float d0 = m.dispx + m.dispx, d1 = m.dispy + m.dispy, d2 = m.dispz + m.dispz;
        const bool h0 = (m.dispx > 0) ? (n0 < d0) : (m.dispx < 0) ? (n0 > d0) : false;
        const bool h1 = (m.dispy > 0) ? (n1 < d1) : (m.dispy < 0) ? (n1 > d1) : false;
        const bool h2 = (m.dispz > 0) ? (n2 < d2) : (m.dispz < 0) ? (n2 > d2) : false;
        const bool fin = live && !(h0 || h1 || h2);
        if (__ballot(fin)) {
          float b[12];
          // f = 1: the segment IS the remaining displacement (disp * 1.0f), its midpoint r + disp (move_p.c:65-74)
          streak12_fast(b, q, dx + m.dispx, dy + m.dispy, dz + m.dispz, m.dispx, m.dispy, m.dispz);
          const int fkey = fin ? pi : -1;
          if (fin) {                                        // move_p.c:103-110 with s_disp = disp; type == 3: done
            dx += m.dispx + m.dispx; dy += m.dispy + m.dispy; dz += m.dispz + m.dispz;
            live = false;
          }
          if (ablate & 32) {}
          else run_deposit<W::DRAIN_BLOCK, W>(b, fkey, lane, s_acc, g_acc, wbase, gsy, gsz, td, ml, n_miss);
        }
      }
    }
    const unsigned long long again = __ballot(live);
    if (STAGE) {
      // the passes that wait read their crossers' results from the queue slots: the stragglers go back into the queue only
      // after that (the caller does it), as late-storing entries
      sg->live = live; sg->again = again; sg->q = q;
      sg->pos_i = make_float4(dx, dy, dz, __int_as_float(pi | (flips << FLIP_SHIFT)));
      sg->disp_idx = make_float4(m.dispx, m.dispy, m.dispz, __int_as_float(idx));
      if (staged) mq->pos_i[k] = make_float4(dx, dy, dz, __int_as_float(live ? -1 : pi));     // (-1: not done, the pass stores the old position)
    } else
    if (live) {                                         // not there yet: back into the queue (max_pass reached)
      const int d = mbcnt64(again);
      mq->pos_i[d] = make_float4(dx, dy, dz, __int_as_float(pi | (flips << FLIP_SHIFT)));
      mq->disp_idx[d] = make_float4(m.dispx, m.dispy, m.dispz, __int_as_float(idx));
      mq->q[d] = q;
    }
    n_again = __popcll(again);
    if (mine && !live) {
#ifdef VPIC_HIP_ABLATION   // 256: no stores of the crossers' final positions; 512: one of the four only (what a float4 position record would issue)
      if (ablate & 256) {} else if (ablate & 512) { stf(p.dx, o4, dx + dy + dz + __int_as_float(pi)); } else
#endif
      if (!staged) { stf_nt(p.dx, o4, dx); stf_nt(p.dy, o4, dy); stf_nt(p.dz, o4, dz); sti_nt(p.i, o4, pi); }
      if (HIST) hist_count<W>(*hk, pi, wbase, gsy, gsz, td);       // (a particle stopped on a face still sits in the array, in cell pi)
      if (flips) {   // the momenta are where the pass that queued the particle stored them (this wavefront, earlier): wait, then negate in place
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (flips & 1) stf(p.ux, o4, -ldf(p.ux, o4));
        if (flips & 2) stf(p.uy, o4, -ldf(p.uy, o4));
        if (flips & 4) stf(p.uz, o4, -ldf(p.uz, o4));
      }
      if (stuck) {
        const int gs = __hip_atomic_fetch_add(nm_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gs < max_nm) { pm[gs].dispx = m.dispx; pm[gs].dispy = m.dispy; pm[gs].dispz = m.dispz; pm[gs].i = m.i + idx_base; }
      }
    }
  }
  return n_again;
}

// ---- the push kernel ---------------------------------------------------------------------------------
// Vector memory of the pass loop.  A wavefront's vector-memory operations complete in issue order and one counter
// (vmcnt) covers loads, stores and atomics alike, so what is waited for must be issued BEFORE what may stay in
// flight.  A pass issues, in this order: its interpolator gather (needed at once; waited for with vmcnt(8)), the
// particle data of the NEXT pass (needed a whole pass later), and -- after the arithmetic -- the six particle
// stores.  Every pass issues the same loads and stores on every path (no branch around them, or the compiler's
// wait counts fall back to vmcnt(0)): the last pass of a wavefront re-reads its own particles, and lanes beyond the
// end of the array read particle np-1 and store into the padding behind max_np (alloc_particles).  No FLAT
// instruction may be pending in the loop for the same reason (see drain_wave and deposit12).
//
// What was measured and dropped in round 2 (tools/ubench/lds_rate.hip, profiles/r02_*): loads as inline assembly
// with hand-placed waits (the register allocator may copy a register whose load is still in flight); a transposed
// deposit (one LDS atomic instruction for the 12 x 5 totals of five runs: ds_add_f32 costs 3 clocks per live LANE,
// so nothing is gained); two or four consecutive particles per lane with register accumulation and no scan (the
// same-address conflicts of the LDS atomics and the smaller chunks cost more than the scan saves: -30 %).
// CHARGELESS: every particle of the species has q == 0 (tracer copies, decks/trecon-part/tracer.cxx:64-70):
// all deposits are additions of zero, so the accumulator window, the cell regrouping that serves it and
// the flush are compiled out; particle states come out bit-identical to the full kernel's.
constexpr int TAIL_CHUNK = 1024;   // TILE order: particles appended since the sort are pushed 1024 to a workgroup, without a window

template <bool CHARGELESS = false, bool FAST = false, int WIN = 0, bool HIST = false, bool SORT = false>
#ifndef VPIC_HIP_PUSH_VGPRS
#define VPIC_HIP_PUSH_VGPRS 80   // (set when six workgroups per CU were in reach; the LDS allows five, which 102 would still fit: 96 / 100 measured level)
#endif
__global__ __launch_bounds__(PUSH_THREADS) __attribute__((amdgpu_num_vgpr(VPIC_HIP_PUSH_VGPRS)))
void advance_p_kernel(ParticlesK p, const float4 *__restrict__ fi, float *__restrict__ g_acc,
                      const DrainParams *__restrict__ dp, const PushParams P) {
  typedef Window<WIN> W;
  typedef typename W::acc_t acc_t;
  constexpr bool TILE = W::TILE;
  constexpr bool DET = is_det<W>::value;         // deterministic accumulation: fixed-point sums, every lane adds for itself
  constexpr bool UNORDERED = WIN == 3 || WIN == 5 || WIN == 6;   // sorted by tile only (or no window at all): no regrouping, no scan in the main pass
  constexpr int WX = W::WX, NSLOT_PAD = W::NSLOT_PAD;
  __shared__ acc_t s_acc[12 * NSLOT_PAD];
  __shared__ WaveQueue s_mq[WAVES];
  __shared__ MissList s_miss[WAVES];
  __shared__ int s_wbase;
  __shared__ unsigned s_cnt[HIST ? (NSLOT_PAD + 1) / 2 : SORT ? NSLOT_PAD : 1];   // (SORT: the next free place of every window cell in the new order)
  // STAGE: the positions of a pass wait in registers for the pass's cell-crossers (see Staged above) -- the instances that never
  // regroup their lanes: species sorted by tile only and charge-0 copies
  constexpr bool STAGE = (WIN == 3 || CHARGELESS) && !HIST && !SORT;
  __shared__ StagePark s_park[STAGE ? WAVES : 1];
  static_assert(!HIST || WIN == 2, "the histogram of the next sort is taken in tile order, by cell");
  static_assert(!SORT || (WIN == 2 && !HIST && !CHARGELESS), "the sort inside the push: tile order by cell, its counts taken by the push before");

#ifdef VPIC_HIP_ABLATION   // (every instance honours the bits in such a build, the tile kernels included)
  const int ablate = P.ablate | (CHARGELESS ? (1 | 8 | 16 | 32) : 0);
#else
  constexpr int ablate = CHARGELESS ? (1 | 8 | 16 | 32) : 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // stated wave-uniform: the pass loop and its exit become scalar control flow
  const unsigned chunk = xcd_block(blockIdx.x, gridDim.x);
  // the workgroup's particles [first, last) and each wavefront's share of them (wave_span, a multiple of 64)
  int first, last, wave_span = 0, tile_base = NO_WINDOW;
  int first2 = 0, last2 = 0;               // TILE: the tile's share of the particles appended since the sort (k_tail_sort), pushed after its own
  int bx0 = 0, by0 = 0, bz0 = 0;           // TILE: the coordinates of tile_base
  if (TILE) {
    if (chunk < (unsigned)P.n_launch) {
      const unsigned tile = P.tile_list ? (unsigned)P.tile_list[chunk] : chunk;
      // one tile: what the sort put there (clipped to the array; dead slots carry index -1 and sit out the pass) ...
      first = min(P.tpart[tile * TILE_CELLS], P.np);
      last = min(P.tpart[tile * TILE_CELLS + TILE_CELLS], P.np);
      if (P.ttail) {   // ... and its share of the particles appended since, regrouped by tile before this launch
        first2 = min(P.n_sorted + P.ttail[tile * TILE_CELLS], P.np);
        last2 = min(P.n_sorted + P.ttail[tile * TILE_CELLS + TILE_CELLS], P.np);
      }
      const unsigned txy = tile % (unsigned)(P.ntx * P.nty), tz = tile / (unsigned)(P.ntx * P.nty);
      const unsigned tx = txy % (unsigned)P.ntx, ty = txy / (unsigned)P.ntx;
      tile_base = TILE_EDGE * ((int)tx + P.sy * (int)ty + P.sz * (int)tz);   // voxel one cell before the tile on every axis
      bx0 = TILE_EDGE * (int)tx; by0 = TILE_EDGE * (int)ty; bz0 = TILE_EDGE * (int)tz;
    } else {                                // appended particles that were not regrouped: no window, every deposit goes to the global accumulator
      if (chunk - (unsigned)P.n_launch >= (unsigned)P.tail_chunks) return;   // (the grid is rounded up to a multiple of 8)
      first = min(P.n_sorted, P.np) + (int)(chunk - (unsigned)P.n_launch) * TAIL_CHUNK;
      last = min(first + TAIL_CHUNK, P.np);
    }
    if (first >= last && first2 >= last2) return;
  } else {
    wave_span = 64 * P.iters;
    if ((long long)chunk * (WAVES * wave_span) >= P.np) return;   // whole workgroup leaves together
    first = (int)chunk * (WAVES * wave_span);
    last = P.np;
  }

#ifndef VPIC_HIP_FOLLOW
#define VPIC_HIP_FOLLOW 1
#endif
  constexpr int FOLLOW_REACH = 3;          // (cells: 27 steps of the two-stream decks' beams; past that the deck's instability has heated them anyway)
  // TILE: the window FOLLOWS the tile's particles.  Between sorts a beam drifts out of the halo as one (0.11 cells per step on
  // the two-stream decks: from the ninth step on every deposit of the leading cells missed the window -- twelve global atomics
  // each -- and at sort_interval = 20 the launch averaged 26 ms instead of 16.5).  64 particles sampled evenly across the tile's
  // range say where the tile's particles ARE; of the window positions up to FOLLOW_REACH cells off the tile's own, per axis, the one that
  // holds most of the sample wins, ties going to the one with most of it away from the window's rim.  (A window is 6 cells wide,
  // a tile's particles spread over 4-5: there is always a position that holds them all while they move together.)
  // (Tried: the first pass's particle loads issued here, before the window is cleared and the workgroup's barrier, instead of
  // behind them: -0.5 % at 64 ppc, -1.6 % at 32 -- five workgroups per CU cover a workgroup's start already, and the eight
  // values held across the barrier cost more than they save.)
  int sample = -1;
  bool follow = false;
  if (TILE && VPIC_HIP_FOLLOW && tile_base != NO_WINDOW) {
    unsigned flag = 0;                                      // (a SCALAR load: the word was written by a kernel before this one)
    if (P.follow < 0) asm volatile("s_load_dword %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=s"(flag) : "s"(P.crossed));
    follow = P.follow > 0 || flag == 1u;
  }
  if (follow && wave == 0 && last > first)
    sample = p.i[first + (int)(((long long)lane * (last - first)) >> 6)];            // (in flight while the window is cleared)
  if (!CHARGELESS)
    for (int k = tid; k < 12 * NSLOT_PAD; k += PUSH_THREADS) s_acc[k] = 0;
  if (HIST)
    for (int k = tid; k < (NSLOT_PAD + 1) / 2; k += PUSH_THREADS) s_cnt[k] = 0;
  if (SORT)
    for (int k = tid; k < NSLOT_PAD; k += PUSH_THREADS) s_cnt[k] = 0;
  if (TILE) {
    if (follow) {
      if (wave == 0) {
        const bool have = sample >= 0;
        const int cz = (int)(__umulhi((unsigned)max(sample, 0), P.mul_sz) >> P.sh_sz), rem = max(sample, 0) - cz * P.sz;
        const int cy = (int)(__umulhi((unsigned)rem, P.mul_sy) >> P.sh_sy), cx = rem - cy * P.sy;
        int best[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const int c = a == 0 ? cx : a == 1 ? cy : cz, b0 = a == 0 ? bx0 : a == 1 ? by0 : bz0, n = a == 0 ? P.nx : a == 1 ? P.ny : P.nz;
          const int hi = max(n - 4, 0);                                       // (cells 0 .. n + 1 exist: the last window begins at n - 4)
          int pick = min(max(b0, 0), hi), top = -1;
#pragma unroll
          for (int t = 0; t < 2 * FOLLOW_REACH + 1; t++) {
            const int d = (t & 1) ? -((t + 1) >> 1) : (t >> 1);                      // 0, -1, +1, -2, +2, ...: the nearest position wins the ties
            const int b = min(max(b0 + d, 0), hi);
            const int cover = __popcll(__ballot(have && c >= b && c <= b + 5)), inner = __popcll(__ballot(have && c > b && c < b + 5));
            const int score = 2 * cover + inner;
            if (score > top) { top = score; pick = b; }
          }
          best[a] = pick;
        }
        if (lane == 0) s_wbase = best[0] + P.sy * best[1] + P.sz * best[2];
      }
    } else if (tid == 0) s_wbase = tile_base;
  } else if (DET) {
    if (tid == 0) s_wbase = NO_WINDOW;                 // Window<5>: no window, every deposit is a global 64-bit atomic
  } else if (!CHARGELESS && wave == 0) {
    // Centre the window on the median cell of 64 particles sampled evenly across the chunk.
    // (Stragglers -- particles that crossed into another row or plane, or wrapped around the
    // periodic box, since the last sort -- sit far from the chunk's cells and must not drag the
    // window with them; the median ignores them.)
    const int chunk_n = min(WAVES * wave_span, P.np - first);
    const int sidx = first + (int)(((long long)lane * chunk_n) >> 6);
    const int k0 = p.i[sidx];
    int rank = 0;                                      // sample keys below mine (ties by lane)
    for (int l = 0; l < 64; l++) {
      const int kl = __builtin_amdgcn_readlane(k0, l);
      rank += (kl < k0 || (kl == k0 && l < lane)) ? 1 : 0;
    }
    const unsigned long long is_med = __ballot(rank == 31);
    const int m = __builtin_amdgcn_readlane(k0, __ffsll((long long)is_med) - 1) - WX / 2 + WMARGIN;
    if (lane == 0) s_wbase = m - WMARGIN;
  }
  __syncthreads();
  const int wbase = s_wbase;
  const int gsy = P.sy, gsz = P.sz;
  WaveQueue *mq = &s_mq[wave];
  MissList *ml = &s_miss[wave];
  if (lane == 0) ml->total = 0;
  int n_miss = 0;                                      // wave-uniform
  int n_mq = 0, n_crossed = 0;                         // wave-uniform

  TileDiv td = {0u, 0u, 0u, 0u, 0.0};
  if (TILE) { td.mul_sy = P.mul_sy; td.sh_sy = P.sh_sy; td.mul_sz = P.mul_sz; td.sh_sz = P.sh_sz; }
  if (DET) td.scale = P.acc_scale;
  // the main pass's copies of what its slot arithmetic and its loads read every pass: in vector registers (in_vgpr)
  TileDiv vtd = td;
  int vsy = gsy, vsz = gsz;
  if (TILE && !DET) {
    vtd.mul_sy = (unsigned)in_vgpr((int)P.mul_sy); vtd.sh_sy = (unsigned)in_vgpr((int)P.sh_sy);
    vtd.mul_sz = (unsigned)in_vgpr((int)P.mul_sz); vtd.sh_sz = (unsigned)in_vgpr((int)P.sh_sz);
    vsy = in_vgpr(gsy); vsz = in_vgpr(gsz);
  }
  const int vnp1 = TILE ? in_vgpr(P.np - 1) : P.np - 1;
  HistK hk;
  if (HIST || SORT) {
    hk.hist = P.hist; hk.s_cnt = s_cnt;
    hk.tk.sy = P.sy; hk.tk.sz = P.sz; hk.tk.ntx = P.ntx; hk.tk.nty = P.nty; hk.tk.ntz = P.ntz; hk.tk.ntiles = P.ntiles;
    hk.tk.mul_sy = P.mul_sy; hk.tk.sh_sy = P.sh_sy; hk.tk.mul_sz = P.mul_sz; hk.tk.sh_sz = P.sh_sz;
  }

  // STAGE: the waiting passes' positions (slot 0, 1) and how many wait
  float st0x = 0.f, st0y = 0.f, st0z = 0.f, st1x = 0.f, st1y = 0.f, st1z = 0.f;
  int st0i = 0, st1i = 0;                              // ... and cells: a waiting pass rewrites its cells as a whole span too (see drain_release)
  int n_pend = 0;                                      // wave-uniform
  const bool stage = STAGE && P.stage != 0;            // wave-uniform (the host switches it on for species whose queue fills every other pass)
  StagePark *park = &s_park[STAGE ? wave : 0];
  // Finish what is queued and store the positions of the passes that wait: ONE round normally (see drain_wave), every round it
  // takes when `all`.  Returns the queue's new length (the stragglers).
  auto drain_release = [&](const bool all) -> int {
    Straggler sg; sg.live = false; sg.again = 0ull;
    int n_back = 0;
    if (n_mq > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      n_back = drain_wave<FAST, W, HIST, STAGE>(p, mq, n_mq, lane, s_acc, g_acc, wbase, dp, ablate, all ? (1 << 30) : StageRounds<CHARGELESS>::value, P.idx_base, ml, n_miss, td.scale, &hk, &sg);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
#pragma unroll
    for (int j = 0; j < STAGE_CAP; j++) {
      if (j >= n_pend) break;                          // wave-uniform
      const int pb = __builtin_amdgcn_readfirstlane(park->base[j]), qb = __builtin_amdgcn_readfirstlane(park->qb[j]);
      const unsigned long long cmj = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)park->cm_hi[j]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)park->cm_lo[j]);
      const unsigned long long actj = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)park->act_hi[j]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)park->act_lo[j]);
      float fx = j ? st1x : st0x, fy = j ? st1y : st0y, fz = j ? st1z : st0z;
      int fi_ = j ? st1i : st0i;
      const bool cr = (cmj >> lane) & 1ull;
      if (cr) {
        const float4 r = mq->pos_i[qb + mbcnt64(cmj)];
        const int newi = __float_as_int(r.w);
        if (newi >= 0) { fx = r.x; fy = r.y; fz = r.z; fi_ = newi; }
      }
      const unsigned o4 = (unsigned)(pb + lane) << 2;
      // the cells as a whole span as well: where half the particles change cell every sector of the span holds a new value, and a
      // whole sector is one write where 4 bytes of it are a read-modify-write of the memory's own
      if ((actj >> lane) & 1ull) { stf(p.dx, o4, fx); stf(p.dy, o4, fy); stf(p.dz, o4, fz); sti(p.i, o4, fi_); }
    }
    n_pend = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (the results were read: the slots may be written again)
    if (sg.live) {                                     // the stragglers: back to the front of the queue, late-storing from now on
      const int d = mbcnt64(sg.again);
      mq->pos_i[d] = sg.pos_i; mq->disp_idx[d] = sg.disp_idx; mq->q[d] = sg.q;
    }
    return n_back;
  };
  const float one = 1.f, one_third = 1. / 3., two_fifteenths = 2. / 15.;
  const float qdt_2mc = P.qdt_2mc, cdt_dx = P.cdt_dx, cdt_dy = P.cdt_dy, cdt_dz = P.cdt_dz;

  // SORT, first the places: the workgroup counts its particles by window cell (one LDS atomic per run of equal cells), reserves
  // each occupied cell's share of the cell's range in the new order with ONE returning atomic on the cell's cursor, and keeps
  // the reserved ranges' next free places in LDS -- the pass loop then hands out places with LDS atomics only.  (Particles whose
  // cell lies outside the window -- strays two cells from their tile, appended particles without a window -- go to the global
  // cursor from the pass loop, a run at a time.)
  if (SORT) {
    typedef __attribute__((address_space(1))) int *global_int_ptr;
#pragma unroll 1
    for (int seg = 0; seg < 2; seg++) {
      const int sfirst = seg ? first2 : first, slast = seg ? last2 : last;
      constexpr int AHEAD = 8;                               // keys in flight per lane: the loop is all load latency otherwise
#pragma unroll 1
      for (int base = sfirst + wave * 64; base < slast; base += AHEAD * PUSH_THREADS) {     // (wave-uniform bounds)
        int keys[AHEAD];
        const int *const cell = p.i;
#pragma unroll
        for (int j = 0; j < AHEAD; j++) { const int at = base + j * PUSH_THREADS + lane; keys[j] = at < slast ? ldi(cell, (unsigned)at << 2) : -1; }
#pragma unroll
        for (int j = 0; j < AHEAD; j++) {
          const int key = keys[j];
          const int slot = key >= 0 ? slot_of<W>(key, wbase, vsy, vsz, vtd) : -1;
          const int kprev = __builtin_amdgcn_update_dpp(-2, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
          const unsigned long long heads = __ballot(kprev != key);
          const unsigned long long above = heads & ~((2ull << lane) - 1ull);
          const int run_end = above ? __ffsll((long long)above) - 1 : 64;
          if (((heads >> lane) & 1ull) && slot >= 0) atomicAdd(&s_cnt[slot], (unsigned)(run_end - lane));
        }
      }
    }
    __syncthreads();
    for (int t = tid; t < W::NSLOT; t += PUSH_THREADS) {
      const unsigned c = s_cnt[t];
      if (c) {
        const int lx = t % WX, lyz = t / WX, ly = lyz % WX, lz = lyz / WX;
        s_cnt[t] = (unsigned)__hip_atomic_fetch_add((global_int_ptr)P.next + sort_key<true>(wbase + lx + gsy * ly + gsz * lz, hk.tk), (int)c,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }

  // software pipeline: the raw (array-order) particle data of the next pass is in flight while
  // this pass computes
#pragma unroll 1
  for (int seg = 0; seg < (TILE ? 2 : 1); seg++) {     // TILE: the tile's own particles, then its appended ones
  const int sfirst = seg ? first2 : first, slast = seg ? last2 : last;
  if (TILE && sfirst >= slast) continue;
  const int sspan = TILE ? ((slast - sfirst + 64 * WAVES - 1) / (64 * WAVES)) * 64 : wave_span;   // a wavefront's share, a multiple of 64
  const int wave_passes = TILE ? sspan >> 6 : P.iters;
  // (SORT, tried: the four wavefronts taking turns along the tile's range, so that what they write to one cell's range is
  // written within a pass or two -- the write traffic stays at 87 GB per launch for 43 GB of particles: every run of a pass
  // begins and ends inside a 32-byte sector, and the sectors reach HBM as they are.  The staged scatter of the sort proper
  // writes 44 GB; what this kernel saves is that sort's 39 GB of reads.)
  constexpr int PASS_STRIDE = 64;
  const int wave_first = sfirst + wave * sspan;
  const int wave_last = TILE ? min(slast, wave_first + sspan) : slast;     // TILE: the next wavefront's (or tile's) particles begin here
  float r_dx, r_dy, r_dz, r_ux, r_uy, r_uz, r_q;
  int r_key;
  {
    const unsigned k4 = (unsigned)min(wave_first + lane, P.np - 1) << 2;
    r_key = ldi(p.i, k4); r_dx = ldf(p.dx, k4); r_dy = ldf(p.dy, k4); r_dz = ldf(p.dz, k4);
    r_ux = ldf(p.ux, k4); r_uy = ldf(p.uy, k4); r_uz = ldf(p.uz, k4); r_q = ldf(p.q, k4);
  }

#pragma unroll 1
  for (int it = 0; it < wave_passes; it++) {
    const int base = wave_first + it * PASS_STRIDE;
    if (base >= wave_last) break;                      // wave-uniform
    int idx = base + lane, key = (base + lane < wave_last) ? r_key : -1;      // lanes beyond the end hold a particle that is not theirs (np-1 at the end of the array)
    float dx = r_dx, dy = r_dy, dz = r_dz, ux = r_ux, uy = r_uy, uz = r_uz, q = r_q;
    // regroup the 64 particles by cell: lane `dest` takes over the particle this lane loaded
    // (skipped when the cells already ascend along the lanes, the usual case right after a sort)
    const int kk = key < 0 ? 0x7fffffff : key;
    const int kprev = __builtin_amdgcn_update_dpp((int)0x80000000, kk, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    if (!UNORDERED && !(ablate & 16) && __ballot(kk < kprev)) {
      const int dest = group_lanes_by_key(key, lane);
      if (__ballot(dest != lane)) {
        const int a4 = dest << 2;
        idx = base + __builtin_amdgcn_ds_permute(a4, lane);
        key = __builtin_amdgcn_ds_permute(a4, key);
        dx = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(dx)));
        dy = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(dy)));
        dz = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(dz)));
        ux = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(ux)));
        uy = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(uy)));
        uz = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(uz)));
        q = __int_as_float(__builtin_amdgcn_ds_permute(a4, __float_as_int(q)));
      }
    }
    // SORT: where this particle goes in the new order.  The first lane of every run of equal cells reserves the run's places
    // with one returning atomic on the cell's cursor (issued every pass, ahead of the gather: it is back before the stores).
    int dst = idx, sort_slot = -2, sort_b = 0, sort_head = 0;
    if (SORT) {
      typedef __attribute__((address_space(1))) int *global_int_ptr;
      const int slot = key >= 0 ? slot_of<W>(key, wbase, vsy, vsz, vtd) : -1;
      sort_slot = slot;                                                            // (the deposit below takes it from here)
      const int kprev2 = __builtin_amdgcn_update_dpp(-2, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
      const unsigned long long heads = __ballot(kprev2 != key);                    // (lane 0 reads -2: always a head)
      const unsigned long long upto = (2ull << lane) - 1ull;
      const int head = 63 - __clzll((long long)(heads & upto));
      const unsigned long long above = heads & ~upto;
      const int run_end = above ? __ffsll((long long)above) - 1 : 64;
      const bool reserves = lane == head && key >= 0;
      int b = 0;
      if (reserves && slot >= 0) b = (int)atomicAdd(&s_cnt[slot], (unsigned)(run_end - lane));
      if (__ballot(reserves && slot < 0)) {                                        // rare: outside the window
        if (reserves && slot < 0)
          b = __hip_atomic_fetch_add((global_int_ptr)P.next + sort_key<true>(key, hk.tk), run_end - lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // (the place itself is worked out with the stores: a run outside the window asked the GLOBAL cursor, and an answer read here
      // would be waited for here -- two cells of thermal spread between sorts put such a run into three passes of four -- while
      // behind the gather it has come back with the gather's data: vector memory returns in order)
      sort_b = b; sort_head = head;
    }
    // Memory pipeline of the pass (see above): gather, then the next pass's particles.
    const float4 *f = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(fi) + (unsigned)((ablate & 4) ? 0 : max(key, 0)) * 80u);
    const float4 fe_x = f[0], fe_y = f[1], fe_z = f[2], fb0 = f[3];
    const float2 fb1 = *reinterpret_cast<const float2 *>(f + 4);
    __builtin_amdgcn_sched_barrier(0);
    {
      const int k = base + ((it + 1 < wave_passes) ? PASS_STRIDE : 0) + lane;
      const unsigned k4 = (unsigned)min(k, vnp1) << 2;
      r_key = ldi(p.i, k4); r_dx = ldf(p.dx, k4); r_dy = ldf(p.dy, k4); r_dz = ldf(p.dz, k4);
      r_ux = ldf(p.ux, k4); r_uy = ldf(p.uy, k4); r_uz = ldf(p.uz, k4); r_q = ldf(p.q, k4);
    }
    __builtin_amdgcn_sched_barrier(0);
    // Branch-free pass body: every lane computes (a lane past the end of the array holds the data of particle
    // np-1 and reads the interpolator of voxel 0), only the queueing is predicated.  A lane that leaves its cell
    // deposits with charge 0, so it does not break its cell's run.
    const bool active = key >= 0;
    float a[12];
    bool crosser;
    float sux, suy, suz;                          // the momenta as stored (advance_p.cxx:106-108)
    float spx = 0.f, spy = 0.f, spz = 0.f;        // STAGE: the positions as they would be stored now
    {
      const unsigned o4 = (unsigned)idx << 2;
      float v0, v1, v2, v3, v4, v5;
      if (FAST) {
        const float hax = qdt_2mc * __builtin_fmaf(dz, __builtin_fmaf(dy, fe_x.w, fe_x.z), __builtin_fmaf(dy, fe_x.y, fe_x.x));
        const float hay = qdt_2mc * __builtin_fmaf(dx, __builtin_fmaf(dz, fe_y.w, fe_y.z), __builtin_fmaf(dz, fe_y.y, fe_y.x));
        const float haz = qdt_2mc * __builtin_fmaf(dy, __builtin_fmaf(dx, fe_z.w, fe_z.z), __builtin_fmaf(dx, fe_z.y, fe_z.x));
        const float cbx = __builtin_fmaf(dx, fb0.y, fb0.x), cby = __builtin_fmaf(dy, fb0.w, fb0.z), cbz = __builtin_fmaf(dz, fb1.y, fb1.x);
        ux += hax; uy += hay; uz += haz;
        v0 = qdt_2mc * fast_rsq(__builtin_fmaf(ux, ux, __builtin_fmaf(uy, uy, __builtin_fmaf(uz, uz, one))));
        v1 = __builtin_fmaf(cbx, cbx, __builtin_fmaf(cby, cby, cbz * cbz));
        v2 = (v0 * v0) * v1;
        v3 = v0 * __builtin_fmaf(v2, __builtin_fmaf(v2, two_fifteenths, one_third), one);
        v4 = (v3 + v3) * fast_rcp(__builtin_fmaf(v1, v3 * v3, one));
        v0 = __builtin_fmaf(v3, __builtin_fmaf(uy, cbz, -(uz * cby)), ux);
        v1 = __builtin_fmaf(v3, __builtin_fmaf(uz, cbx, -(ux * cbz)), uy);
        v2 = __builtin_fmaf(v3, __builtin_fmaf(ux, cby, -(uy * cbx)), uz);
        ux = __builtin_fmaf(v4, __builtin_fmaf(v1, cbz, -(v2 * cby)), ux);
        uy = __builtin_fmaf(v4, __builtin_fmaf(v2, cbx, -(v0 * cbz)), uy);
        uz = __builtin_fmaf(v4, __builtin_fmaf(v0, cby, -(v1 * cbx)), uz);
        ux += hax; uy += hay; uz += haz;
      } else {
        // advance_p.cxx:74-82
        const float hax = qdt_2mc * ((fe_x.x + dy * fe_x.y) + dz * (fe_x.z + dy * fe_x.w));
        const float hay = qdt_2mc * ((fe_y.x + dz * fe_y.y) + dx * (fe_y.z + dz * fe_y.w));
        const float haz = qdt_2mc * ((fe_z.x + dx * fe_z.y) + dy * (fe_z.z + dx * fe_z.w));
        const float cbx = fb0.x + dx * fb0.y;
        const float cby = fb0.z + dy * fb0.w;
        const float cbz = fb1.x + dz * fb1.y;
        // advance_p.cxx:87-105
        ux += hax; uy += hay; uz += haz;
        v0 = div_normal(qdt_2mc, sqrt_normal(one + (ux * ux + (uy * uy + uz * uz))));
        v1 = cbx * cbx + (cby * cby + cbz * cbz);
        v2 = (v0 * v0) * v1;
        v3 = v0 * (one + v2 * (one_third + v2 * two_fifteenths));
        v4 = div_normal(v3, one + v1 * (v3 * v3));
        v4 += v4;
        v0 = ux + v3 * (uy * cbz - uz * cby);
        v1 = uy + v3 * (uz * cbx - ux * cbz);
        v2 = uz + v3 * (ux * cby - uy * cbx);
        ux += v4 * (v1 * cbz - v2 * cby);
        uy += v4 * (v2 * cbx - v0 * cbz);
        uz += v4 * (v0 * cby - v1 * cbx);
        ux += hax; uy += hay; uz += haz;
      }
      sux = ux; suy = uy; suz = uz;
      // advance_p.cxx:109-122
      if (FAST) v0 = fast_rsq(__builtin_fmaf(ux, ux, __builtin_fmaf(uy, uy, __builtin_fmaf(uz, uz, one))));
      else v0 = div_normal(one, sqrt_normal(one + (ux * ux + (uy * uy + uz * uz))));
      ux *= cdt_dx; uy *= cdt_dy; uz *= cdt_dz;
      ux *= v0; uy *= v0; uz *= v0;                             // half displacement in cell units (what a mover carries)
      v0 = dx + ux; v1 = dy + uy; v2 = dz + uz;
      v3 = v0 + ux; v4 = v1 + uy; v5 = v2 + uz;
      const bool incell = (ablate & 2) || fmaxf(fmaxf(fabsf(v3), fabsf(v4)), fabsf(v5)) <= one;   // advance_p.cxx:124-125
      crosser = active && !incell;
      // a crosser keeps its position until drain_wave has finished its move (advance_p.cxx:166-175); storing the
      // old value here instead of skipping the lane keeps every store a whole 256-byte span (no partial lines)
      // TILE: a wavefront's share ends inside the array, where the next lanes' slots hold a neighbour's particles: lanes
      // without a particle are masked out of the stores (no branch: the six stores stay below the skip threshold)
      if (SORT) {
        dst = __shfl(sort_b, sort_head) + (lane - sort_head);
        if (active) {
          const unsigned d4 = (unsigned)dst << 2;
          stf(P.out.ux, d4, sux); stf(P.out.uy, d4, suy); stf(P.out.uz, d4, suz);
          stf(P.out.dx, d4, incell ? v3 : dx); stf(P.out.dy, d4, incell ? v4 : dy); stf(P.out.dz, d4, incell ? v5 : dz);
          sti(P.out.i, d4, key); stf(P.out.q, d4, q);
        }
      } else
      if (STAGE && stage) {
        // the momenta are final; the positions wait for the pass's crossers (drain_release stores them)
        if (!TILE || active) { stf(p.ux, o4, sux); stf(p.uy, o4, suy); stf(p.uz, o4, suz); }
        // (which slot the pass waits in is decided when it is parked: a pass that would overflow the queue releases the waiting ones first)
        spx = incell ? v3 : dx; spy = incell ? v4 : dy; spz = incell ? v5 : dz;
      } else
      if ((!TILE || active) && !(ablate & 128)) {
        stf_nt(p.ux, o4, sux); stf_nt(p.uy, o4, suy); stf_nt(p.uz, o4, suz);
        stf(p.dx, o4, incell ? v3 : dx); stf(p.dy, o4, incell ? v4 : dy); stf(p.dz, o4, incell ? v5 : dz);
      }
      if (!CHARGELESS && !(ablate & 1)) {
        const float qd = (incell && active) ? q : 0.f;
        // The 12 deposit terms with contracted multiply-adds in BOTH arithmetic modes: they are summed in an order of the
        // machine's choosing anyway (the accumulators agree with the reference's to 2e-6 of the largest entry, not bit for
        // bit), each term is within an ulp of the reference's, and no particle state depends on them (+1.3 % measured).
        streak12_fast(a, qd, v0, v1, v2, ux, uy, uz);
      } else {
#pragma unroll
        for (int k = 0; k < 12; k++) a[k] = 0.f;
      }
    }
    // in-cell deposits: a crosser lane carries zeros, so it does not break its cell's run
    // (without a scan a lane that leaves its cell has nothing to add: its zeros would only collide with its neighbours' sums)
    if constexpr (WIN == 4) run_deposit_fixed<16, W>(a, key, lane, s_acc, g_acc, wbase, gsy, gsz, td, ml, n_miss);   // (integer run sums: see above)
    else
    if (!CHARGELESS) run_deposit<UNORDERED ? 1 : TILE ? TILE_MAIN_BLOCK : MAIN_BLOCK, W>(a, (UNORDERED && crosser) ? -1 : key, lane, s_acc, g_acc, wbase, vsy, vsz, vtd, ml, n_miss,
                                                                                            HIST ? &hk : nullptr, active && !crosser, SORT ? sort_slot : -2);
    // (HIST: the particles that stay in their cell were counted with their run's deposit; a crosser is counted when its move is done)
    // queue this pass's cell-crossers in lane (= cell) order; no atomics, the wavefront is in step.
    // phase 0 (rare: the pass would overflow the queue) drains what is queued first; phase 1
    // enqueues and drains one full wavefront of crossers when there is one.
    {
      const unsigned long long cm = __ballot(crosser);
      const int cnt = __popcll(cm);
      n_crossed += cnt;
      if constexpr (STAGE) {
        // (one call site of drain_release: the instruction cache holds one copy of the crossers' path per instance)
        int phase = (n_mq + cnt > MQW) ? 0 : 1, attempt = 0;      // 0: a pass that would overflow the queue has it finished first
#pragma unroll 1
        for (;;) {
          if (phase == 1) {
            if (crosser) {
              const int d = n_mq + mbcnt64(cm);
              mq->pos_i[d] = make_float4(dx, dy, dz, __int_as_float(stage ? (key | (1 << STAGED_BIT)) : key));
              mq->disp_idx[d] = make_float4(ux, uy, uz, __int_as_float(idx));
              mq->q[d] = q;
            }
            if (stage) {
              if (n_pend == 0) { st0x = spx; st0y = spy; st0z = spz; st0i = key; } else { st1x = spx; st1y = spy; st1z = spz; st1i = key; }
              const unsigned long long act = __ballot(!TILE || active);
              if (lane == 0) {
                park->base[n_pend] = base; park->qb[n_pend] = n_mq;
                park->cm_lo[n_pend] = (unsigned)cm; park->cm_hi[n_pend] = (unsigned)(cm >> 32);
                park->act_lo[n_pend] = (unsigned)act; park->act_hi[n_pend] = (unsigned)(act >> 32);
              }
              n_pend++;
            }
            n_mq += cnt;
            if (n_mq < 64 && n_pend < STAGE_CAP) break;
          }
          n_mq = drain_release(phase == 0 && attempt > 0);       // one round; everything when one round did not make room
          if (phase == 1) break;
          attempt++;
          if (n_mq + cnt <= MQW) phase = 1;
        }
      } else {
      int attempt = 0;
#pragma unroll 1
      for (int phase = (n_mq + cnt > MQW) ? 0 : 1; phase < 2;) {
        if (phase == 1) {
          if (crosser) {                               // (ux, uy, uz hold the half displacement here)
            const int d = n_mq + mbcnt64(cm);
            mq->pos_i[d] = make_float4(dx, dy, dz, __int_as_float(key));
            mq->disp_idx[d] = make_float4(ux, uy, uz, __int_as_float(SORT ? dst : idx));   // (SORT: its place in the new order)
            mq->q[d] = q;
          }
          n_mq += cnt;
          if (n_mq < 64) break;
        }
        const int n_now = min(n_mq, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // One round per batch (a full pass and a final-segment pass, see drain_wave): most crossers need two segments, and
        // another full pass for the two or three lanes that cut a corner costs what a full pass costs -- those ride with the
        // next batch.  Only when that does not make room for this pass's crossers (phase 0, second attempt) is the batch
        // finished whatever it takes.
        const int cap = (phase == 0 && attempt > 0) ? (1 << 30) : 1;
        const int n_back = drain_wave<FAST, W, HIST>(SORT ? P.out : p, mq, n_now, lane, s_acc, g_acc, wbase, dp, ablate, cap, P.idx_base, ml, n_miss, td.scale, &hk);
            const int n_left = n_mq - n_now;               // move what stayed behind to the front, after the stragglers
        const int src = lane < n_left ? 64 + lane : 0;
        const float4 t0 = mq->pos_i[src], t2 = mq->disp_idx[src];
        const float t1 = mq->q[src];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < n_left) { const int d = n_back + lane; mq->pos_i[d] = t0; mq->disp_idx[d] = t2; mq->q[d] = t1; }
        n_mq = n_back + n_left;
        if (phase == 0) { attempt++; if (n_mq + cnt <= MQW) phase = 1; }
        else phase = 2;
      }
      }
    }
  }
  }   // seg
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // queue writes before the reads below
  if constexpr (STAGE) n_mq = drain_release(true);         // (every crosser finishes: nothing is left in the queue)
  else
  drain_wave<FAST, W, HIST>(SORT ? P.out : p, mq, n_mq, lane, s_acc, g_acc, wbase, dp, ablate, 1 << 30, P.idx_base, ml, n_miss, td.scale, &hk);
  if (!CHARGELESS && !DET) flush_misses(ml, n_miss, g_acc, lane);

  // how many particles left their cell (the host picks the window instance and the sort policy from it)
  // (256 shards on cache lines of their own: one word takes ~90 atomics per microsecond, a launch has 1e5 wavefronts)
  if (lane == 0 && n_crossed) atomicAdd(P.crossed + (blockIdx.x & 255u) * 16u, (unsigned)n_crossed);
  if (!CHARGELESS && lane == 0 && ml->total) atomicAdd(P.crossed + (blockIdx.x & 255u) * 16u + 1u, (unsigned)ml->total);   // (runs that missed the window)

  // ---- flush the window: consecutive lanes -> consecutive floats of consecutive accumulators --
  __syncthreads();
  if (ablate & 8) return;
  // thread -> (component k, first cell c0), fixed for the whole flush: 12 consecutive lanes cover the
  // 12 floats of one accumulator, FLUSH_CELLS (21) accumulators per sweep of the (first 252 threads of the) workgroup
  if (HIST && wbase != NO_WINDOW) {
    for (int slot = tid; slot < W::NSLOT; slot += PUSH_THREADS) {
      const unsigned n = (s_cnt[slot >> 1] >> ((slot & 1) << 4)) & 0xffffu;
      const int lx = slot % WX, lyz = slot / WX, ly = lyz % WX, lz = lyz / WX;
      if (n) atomicAdd(&P.hist[sort_key<true>(wbase + lx + gsy * ly + gsz * lz, hk.tk)], (int)n);
    }
  }
  const int k = tid % 12, c0 = tid / 12;
  if (DET) {
    if (TILE && wbase != NO_WINDOW && tid < 12 * FLUSH_CELLS) {
      unsigned long long *g64 = reinterpret_cast<unsigned long long *>(g_acc);
      for (int cell = c0; cell < W::NSLOT; cell += FLUSH_CELLS) {
        const unsigned long long v = (unsigned long long)s_acc[k * NSLOT_PAD + cell];
        const int lx = cell % WX, lyz = cell / WX, ly = lyz % WX, lz = lyz / WX;
        if (v) atomicAdd(g64 + (size_t)(wbase + lx + gsy * ly + gsz * lz) * 12 + k, v);
      }
    }
  } else if (TILE) {
    if (wbase != NO_WINDOW && tid < 12 * FLUSH_CELLS) {
      for (int cell = c0; cell < W::NSLOT; cell += FLUSH_CELLS) {
        const float v = (float)s_acc[k * NSLOT_PAD + cell];
        const int lx = cell % WX, lyz = cell / WX, ly = lyz % WX, lz = lyz / WX;
        if (v != 0.f) atomicAdd(g_acc + (size_t)(wbase + lx + gsy * ly + gsz * lz) * 12 + k, v);
      }
    }
  } else if (tid < 12 * FLUSH_CELLS) {
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
      const int seg_base = wbase + ((s == 0) ? 0 : (s == 1) ? gsy : (s == 2) ? -gsy : (s == 3) ? gsz : -gsz);
      const acc_t *src = s_acc + k * NSLOT_PAD + s * WX;
      float *dst = g_acc + (size_t)seg_base * 12 + k;
      for (int cell = c0; cell < WX; cell += FLUSH_CELLS) {
        const float v = (float)src[cell];                  // the workgroup's total
        if (v != 0.f) atomicAdd(dst + cell * 12, v);
      }
    }
  }
}

// hands a device counter to the host through mapped pinned memory and clears it (a copy engine transfer behind
// the kernel would cost the stream a queue switch: ~1 ms per launch measured)
__global__ __launch_bounds__(256) void publish_counter_kernel(unsigned *__restrict__ host_word, unsigned *__restrict__ dev_shards, unsigned cycle, unsigned long long follow_from) {
  __shared__ unsigned s_sum[4];
  unsigned v = dev_shards[threadIdx.x * 16];
  dev_shards[threadIdx.x * 16] = 0;
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) host_word[0] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
  __syncthreads();
  v = dev_shards[threadIdx.x * 16 + 1];                  // runs that missed the window -> host_word[3] (Species::follow)
  dev_shards[threadIdx.x * 16 + 1] = 0;
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned missed = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    host_word[3] = missed; host_word[4] = cycle;           // (cycle: which sort the launch came after; for the record)
    // from the next launch on the windows follow their particles (PushParams::follow) -- decided HERE, in stream order: the host
    // enqueues many steps ahead of the device and would act on a count that is several launches old
    // 0 -> 1 on the first launch that missed; 1 -> 2 (given up until the next sort) when a launch that followed still missed four
    // times as much: the species' particles do not move together (a hot plasma spreads every way: nothing to follow, and
    // the sampling would cost its launches 2 %)
    const unsigned state = dev_shards[2];
    if (state == 0u && (unsigned long long)missed > follow_from) dev_shards[2] = 1u;
    else if (state == 1u && (unsigned long long)missed > 4ull * follow_from) dev_shards[2] = 2u;
  }
}

// SORT: every cursor must have ended where the next key begins, or the counts the places were laid out from did not describe
// the array (the species' next push fails loudly on a count that is not zero)
__global__ __launch_bounds__(256) void fuse_check_kernel(const int *__restrict__ cursor, const int *__restrict__ starts, int n1, unsigned *__restrict__ bad) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const bool wrong = k + 1 < n1 && cursor[k] != starts[k + 1];
  const unsigned long long m = __ballot(wrong);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(bad, (unsigned)__popcll(m));
}

__global__ void fuse_clear_kernel(unsigned *__restrict__ w) { *w = 0; }
__global__ void fuse_publish_kernel(unsigned *__restrict__ host_word, const unsigned *__restrict__ dev_word) { *host_word = *dev_word; }

// the same check behind a sort whose scatter worked from counts it did not take itself (k_sort_p, `counted`): the species' next
// push fails loudly when they did not describe the array
int k_sort_check(Engine *e, Species &s, const int *starts, int n1) {
  unsigned *word = reinterpret_cast<unsigned *>(e->counters + 201);
  hipLaunchKernelGGL(fuse_clear_kernel, dim3(1), dim3(1), 0, e->stream, word);
  hipLaunchKernelGGL(fuse_check_kernel, dim3((n1 + 255) / 256), dim3(256), 0, e->stream, (const int *)e->sort_next, starts, n1, word);
  hipLaunchKernelGGL(fuse_publish_kernel, dim3(1), dim3(1), 0, e->stream, s.crossed_host_dev + 2, (const unsigned *)word);
  VH_CHECK(hipGetLastError());
  return 0;
}

// ---- host side -------------------------------------------------------------------------------
// (a species pushed in two launches -- vpic_hip_advance_p_phase -- books its particles with the first and adds the second
// one's time to it: particles < 0 marks the continuation)
static int begin_profile(Engine *e, int64_t particles, int kind = 0, int species = -1) {
  if (!e->profile) return -1;
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    e->ev_pool.push_back({a, b});
    e->ev_particles.push_back(0); e->ev_kind.push_back(0); e->ev_species.push_back(-1);
  }
  const int k = (int)e->ev_used++;
  e->ev_particles[k] = particles; e->ev_kind[k] = (char)kind; e->ev_species[k] = species;
  (void)hipEventRecord(e->ev_pool[k].first, e->stream);
  return k;
}

// The kernel addresses a particle by a 32-bit BYTE offset from the array bases (push.hip, ldf/stf): one launch covers at
// most 2^30 particles.  A larger species is pushed in segments, each a launch of its own on rebased array pointers;
// segments start on a workgroup chunk boundary, so the chunks -- and with them every result -- are those of one launch.
extern "C" int vpic_hip_push_plan(int64_t np, int iters, int64_t *start, int32_t *count, uint32_t *grid, int max_segments) {
  if (np < 0 || iters < 1 || iters > PUSH_ITERS || max_segments < 1) return -1;
  const int64_t per_chunk = (int64_t)PUSH_THREADS * iters;
  // a launch of c particles addresses up to particle c + 62 (the lanes of a last, partly filled pass): (c + 63) * 4 < 2^32
  int64_t seg = (((int64_t)1 << 30) - 64) / per_chunk * per_chunk;
  if (np <= ((int64_t)1 << 30) && (np % 64 == 0 || np <= ((int64_t)1 << 30) - 64)) seg = (np + per_chunk - 1) / per_chunk * per_chunk;   // fits one launch
  int n = 0;
  for (int64_t at = 0; at < np; at += seg, n++) {
    if (n >= max_segments) return -1;
    const int64_t c = np - at < seg ? np - at : seg;
    start[n] = at; count[n] = (int32_t)c;
    const uint32_t chunks = (uint32_t)((c + per_chunk - 1) / per_chunk);
    grid[n] = (chunks + 7u) & ~7u;
  }
  return n;
}

// The tiles that touch a face shared with another domain, and the others (Engine::tile_list): built once per engine.
static int ensure_tile_lists(Engine *e) {
  if (e->tile_list[0] || e->tile_list_n[0] < 0) return 0;
  const TileK tk = make_tile_k(e->gk);
  bool shared[6]; bool any = false;
  for (int f = 0; f < 6; f++) { shared[f] = e->gk.pbc[f] >= 0 && e->gk.pbc[f] != e->gk.rank; any = any || shared[f]; }
  if (!any) { e->tile_list_n[0] = -1; return 0; }          // nothing to split
  std::vector<int> list[2];
  for (int tz = 0; tz < tk.ntz; tz++) for (int ty = 0; ty < tk.nty; ty++) for (int tx = 0; tx < tk.ntx; tx++) {
    const bool edge = (shared[0] && tx == 0) || (shared[3] && tx == tk.ntx - 1) || (shared[1] && ty == 0) || (shared[4] && ty == tk.nty - 1) ||
                      (shared[2] && tz == 0) || (shared[5] && tz == tk.ntz - 1);
    list[edge ? 0 : 1].push_back((tz * tk.nty + ty) * tk.ntx + tx);
  }
  for (int k = 0; k < 2; k++) {
    VH_CHECK(hipMalloc(&e->tile_list[k], sizeof(int) * std::max<size_t>(list[k].size(), 1)));
    if (!list[k].empty()) VH_CHECK(hipMemcpy(e->tile_list[k], list[k].data(), sizeof(int) * list[k].size(), hipMemcpyHostToDevice));
    e->tile_list_n[k] = (int)list[k].size();
  }
  return 0;
}

// phase 0: the whole species in one launch.  phase 1 / 2 (vpic_hip_advance_p_phase, tile order): first the tiles on the
// faces this domain shares with others, together with the particles appended since the sort -- every particle that can
// leave the domain in this step, up to stragglers that drifted into a boundary cell from an interior tile's range since
// the sort -- then, behind whatever the caller puts between the two calls (the packing of the species' boundary movers
// and the start of their exchange), the other tiles.  A species that is not in tile order is pushed whole by phase 1.
int k_advance_p(Engine *e, Species &s, bool async, int phase) {
  const vpic_hip_grid_t &g = e->grid;
  PushParams P;
  // advance_p.cxx:425-428: double for qdt_2mc, float for the cdt_d*
  P.qdt_2mc = (float)(0.5 * s.q_m * g.dt / g.cvac);
  P.cdt_dx = g.cvac * g.dt * g.rdx;
  P.cdt_dy = g.cvac * g.dt * g.rdy;
  P.cdt_dz = g.cvac * g.dt * g.rdz;
  P.np = (int)s.np;
  P.sy = e->gk.sy; P.sz = e->gk.sz;
#ifdef VPIC_HIP_ABLATION
  P.ablate = e->knobs.ablate;
  const int ablating = 0;
#else
  const int ablating = 0;
#endif
  if (s.crossed_host[2]) VH_FAIL("advance_p: %u keys of the species' last sort did not receive the particles the push before it had counted for them", s.crossed_host[2]);
  if (phase == 2 && !s.phase_pending) return 0;          // phase 1 pushed everything
  if (phase != 2) {
    VH_CHECK(hipMemsetAsync(s.nm_dev, 0, sizeof(int), e->stream));   // (phase 2 appends to what the exchange left on the list)
    s.nm = 0;
  }
  s.phase_pending = false;
  P.crossed = s.crossed_dev;
  if (s.np > 0) {
    // particles per cell decide how many 64-particle passes a wavefront makes: a workgroup's chunk
    // should span a little less than the LDS window (measured, tools/iters_sweep.sh: 32 ppc best at 6
    // passes, 64 ppc at 12, 512 ppc at 64; one pass too many and the chunk overflows the window)
    const double ppc = (double)s.np / ((double)e->gk.nx * e->gk.ny * e->gk.nz);
    // Which window: the crossing fraction of this species' previous launch (a pinned word the device wrote behind
    // that launch; a stale value only delays the switch) with hysteresis; VPIC_HIP_WINDOW=wide|narrow overrides.
    if (phase != 2) {
      const double frac = s.np_pushed_last > 0 ? (double)*s.crossed_host / (double)s.np_pushed_last : 0.0;
      s.cross_frac = frac;
      if (frac > 0.30) s.wide_window = true; else if (frac < 0.20) s.wide_window = false;
      if (e->knobs.window == 'w') s.wide_window = true; else if (e->knobs.window == 'n') s.wide_window = false;
      if (ablating) s.wide_window = false;
      s.np_pushed_last = s.np;
    }
    const int wx = s.wide_window ? Window<true>::WX : Window<false>::WX;
    int it = (int)(0.9 * (wx - 2 * WMARGIN) * ppc / PUSH_THREADS);
    P.iters = it < 1 ? 1 : it > PUSH_ITERS ? PUSH_ITERS : it;
    if (e->knobs.iters > 0) P.iters = e->knobs.iters;   // tuning experiments
    int64_t seg_start[4]; int32_t seg_count[4]; uint32_t seg_grid[4];
    const int n_seg = vpic_hip_push_plan(s.np, P.iters, seg_start, seg_count, seg_grid, 4);
    if (n_seg < 1) VH_FAIL("advance_p: cannot plan %lld particles", (long long)s.np);
    // TILE order (the last sort grouped the species by tile, engine.h): one workgroup per tile plus the appended particles
    if (e->time_kernels && phase != 2) { if (!s.ev[0]) for (int i = 0; i < 4; i++) VH_CHECK(hipEventCreate(&s.ev[i])); (void)hipEventRecord(s.ev[0], e->stream); }   // (the regrouping of appended particles below counts as push time for the sort policy)
    // A tile is one workgroup's work.  When the fullest tile alone would take several times what the whole launch takes
    // if balanced (1280 workgroups run at a time: 256 CUs x 5), the species is too clumped for tiles: this launch falls
    // back to the row windows and the next sort to the reference's order.  (The count is the last tile sort's, read from
    // pinned memory without waiting: a stale value only delays the switch.)
    // (phase 2 keeps what phase 1 decided: the word is written by the sort's kernels while the host runs ahead of them, and a
    // flip between the two launches of one push would leave the interior tiles unpushed with the boundary movers on the wire)
    if (phase != 2 && s.tile_valid && (double)s.crossed_host[1] * 1280.0 > 4.0 * (double)s.np && s.crossed_host[1] > 65536u) s.tile_unbalanced = true;
    const bool tiled = phase == 2 ? true : (s.tile_valid && !s.tile_unbalanced && !s.chargeless && !ablating && n_seg == 1);   // (phase 2 only runs behind a phase 1 that split the tiles: phase_pending)
    P.tpart = s.tpart; P.ttail = nullptr; P.n_sorted = (int)s.n_sorted;
    P.tile_list = nullptr; P.n_launch = 0; P.tail_chunks = 0;
    P.nx = e->gk.nx; P.ny = e->gk.ny; P.nz = e->gk.nz;
    // the window follows the tile's particles from the launch after the one whose deposits began to miss the windows (16 runs
    // per tile: publish_counter_kernel) until the next sort: sampling costs a workgroup a dependent load at its
    // start (1-3 % of the launch).  VPIC_HIP_FOLLOW=0|1 overrides.
    P.follow = e->knobs.follow;
    // the positions of a pass wait for its crossers (STAGE instances: species sorted by tile only, charge-0 copies) when the
    // queue fills every other pass anyway -- from a third of the particles crossing per step on (a colder species would pay
    // for half-empty batches: two drains where one did); VPIC_HIP_STAGE=0|1 overrides
    P.stage = e->knobs.stage >= 0 ? e->knobs.stage : (s.cross_frac > 0.33 ? 1 : 0);
    // the sort inside the push (Species::fuse_pending, set by k_sort_p for this very call)
    bool fuse = s.fuse_pending;
    s.fuse_pending = false;
    if (fuse && !(tiled && !s.coarse_sorted && phase == 0 && !e->det_acc && !s.hist_request && s.hist_valid && s.aux.dx && s.tpart2 &&
                  s.np == s.n_sorted && !e->time_kernels)) {
      // not after all (a tile turned out overfull, the next step sorts too and this push must count for it, ...): sort the ordinary way, then push
      if (k_sort_p(e, s, true, false)) return 1;
      return k_advance_p(e, s, async, phase);
    }
    if (phase && tiled && ensure_tile_lists(e)) return 1;
    const bool split = phase && tiled && e->tile_list_n[0] > 0 && e->tile_list_n[1] > 0;   // (a domain whose tiles all lie on shared faces has nothing to push later)
    if (phase == 2 && !split) VH_FAIL("advance_p: phase 2 without phase 1");
    if (tiled) {
      const TileK tk = make_tile_k(e->gk);
      P.ntx = tk.ntx; P.nty = tk.nty; P.ntiles = tk.ntiles;
      P.mul_sy = tk.mul_sy; P.sh_sy = tk.sh_sy; P.mul_sz = tk.mul_sz; P.sh_sz = tk.sh_sz;
      const int64_t behind = s.np > s.n_sorted ? s.np - s.n_sorted : 0;
      const bool regroup_tail = behind >= e->knobs.tail_sort_min && behind > 0 && !e->knobs.no_tail_sort;   // a handful costs less pushed as it is (tests lower the threshold)
      if (phase != 2 && regroup_tail && k_tail_sort(e, s)) return 1;
      if (phase != 2) s.tail_regrouped = regroup_tail && s.tail_sorted;
      P.ttail = s.tail_regrouped ? s.ttail : nullptr;
      P.n_launch = split ? e->tile_list_n[phase - 1] : tk.ntiles;
      P.tile_list = split ? e->tile_list[phase - 1] : nullptr;
      P.tail_chunks = (P.ttail || phase == 2) ? 0 : (int)((behind + TAIL_CHUNK - 1) / TAIL_CHUNK);   // (appended particles that were not regrouped go with the first launch)
      seg_grid[0] = (uint32_t)(((int64_t)P.n_launch + P.tail_chunks + 7) / 8 * 8);
      s.phase_pending = split && phase == 1;
    }
    const int ev = begin_profile(e, phase == 2 ? -1 : s.np, fuse ? 1 : 0, (int)(&s - e->species.data()));
    // deterministic accumulation: the kernels add into the engine's 64-bit fixed-point accumulator (engine.hip, acc_finalize)
    // the histogram of the next sort (Species::hist): tile order by cell, one launch, float sums, no tile anywhere near 2^15 particles
    if (fuse) {                                            // where every key begins in the new order, and the cursors, from the counts of the push before
      const TileK tk = make_tile_k(e->gk);
      if (k_sort_scan(e, s.hist, s.tpart2, tk.ntiles * TILE_CELLS + 1)) return 1;
      P.out = s.aux; P.next = e->sort_next;
    } else { P.out = ParticlesK{}; P.next = nullptr; }
    const bool hist = !fuse && s.hist_request && tiled && !s.coarse_sorted && phase == 0 && !(e->det_acc && !s.chargeless) &&
                      (uint64_t)s.crossed_host[1] + (uint64_t)(s.np > s.n_sorted ? s.np - s.n_sorted : 0) < 30000u;   // (16-bit counters per window cell: the fullest tile AND whatever share of the appended particles its workgroup takes)
    s.hist_request = false; s.hist_valid = false;
    if (hist) {
      const TileK tk = make_tile_k(e->gk);
      const int64_t n1 = (int64_t)tk.ntiles * TILE_CELLS + 1;
      if (s.hist_count < n1) { if (s.hist) VH_CHECK(hipFree(s.hist)); s.hist = nullptr; VH_CHECK(hipMalloc(&s.hist, sizeof(int) * n1)); s.hist_count = n1; }
      VH_CHECK(hipMemsetAsync(s.hist, 0, sizeof(int) * n1, e->stream));
      P.hist = s.hist; P.ntz = tk.ntz;
    } else { P.hist = nullptr; P.ntz = fuse ? make_tile_k(e->gk).ntz : 0; }
    const bool det = e->det_acc && !s.chargeless;
    if (det && acc_prepare_det(e)) return 1;
    P.acc_scale = e->acc_scale;
    float *const g_acc = det ? reinterpret_cast<float *>(e->acc64) : reinterpret_cast<float *>(e->acc);
#define PUSH_LAUNCH(...) hipLaunchKernelGGL((advance_p_kernel<__VA_ARGS__>), dim3(grid), dim3(PUSH_THREADS), 0, e->stream, \
                                            ps, reinterpret_cast<const float4 *>(e->fi), g_acc, s.drain_k, P)
    for (int g = 0; g < n_seg; g++) {
      const int64_t at = seg_start[g];
      const unsigned grid = seg_grid[g];
      ParticlesK ps = s.p;
      ps.dx += at; ps.dy += at; ps.dz += at; ps.i += at; ps.ux += at; ps.uy += at; ps.uz += at; ps.q += at;
      P.np = seg_count[g]; P.idx_base = (int)at;
      if (s.chargeless) { if (e->push_fast) PUSH_LAUNCH(true, true); else PUSH_LAUNCH(true, false); }
      else if (det && tiled && s.coarse_sorted) { if (e->push_fast) PUSH_LAUNCH(false, true, 6); else PUSH_LAUNCH(false, false, 6); }
      else if (det && tiled) { if (e->push_fast) PUSH_LAUNCH(false, true, 4); else PUSH_LAUNCH(false, false, 4); }
      else if (det) { if (e->push_fast) PUSH_LAUNCH(false, true, 5); else PUSH_LAUNCH(false, false, 5); }
      else if (tiled && s.coarse_sorted) { if (e->push_fast) PUSH_LAUNCH(false, true, 3); else PUSH_LAUNCH(false, false, 3); }
      else if (tiled && fuse) { if (e->push_fast) PUSH_LAUNCH(false, true, 2, false, true); else PUSH_LAUNCH(false, false, 2, false, true); }
      else if (tiled && hist) { if (e->push_fast) PUSH_LAUNCH(false, true, 2, true); else PUSH_LAUNCH(false, false, 2, true); }
      else if (tiled) { if (e->push_fast) PUSH_LAUNCH(false, true, 2); else PUSH_LAUNCH(false, false, 2); }
      else if (s.wide_window) { if (e->push_fast) PUSH_LAUNCH(false, true, 1); else PUSH_LAUNCH(false, false, 1); }
      else { if (e->push_fast) PUSH_LAUNCH(false, true, 0); else PUSH_LAUNCH(false, false, 0); }
    }
#undef PUSH_LAUNCH
    if (ev >= 0) (void)hipEventRecord(e->ev_pool[ev].second, e->stream);
    if (hist) s.hist_valid = true;
    if (fuse) {
      const TileK tk = make_tile_k(e->gk);
      const int n1 = tk.ntiles * TILE_CELLS + 1;
      if (k_sort_check(e, s, s.tpart2, n1)) return 1;      // (scratch word 201 of the counter block, like the sort's 200)
      std::swap(s.tpart, s.tpart2); std::swap(s.tpart_count, s.tpart2_count);
      if (k_sort_finish(e, s, true, false)) return 1;     // (swaps the buffers: the sorted particles are the species now)
    }
    if (!s.phase_pending) {       // (the counts of a split push add up in the device's shards)
      // (the windows follow from 16 missed runs per tile on: a miss is twelve global atomics, following costs a tile's workgroup a
      // dependent load at its start -- the two meet there, measured at 32 and 64 particles per cell)
      // (a launch that sorted as it pushed deposited through the windows of the OLD order: its misses say nothing about the new
      // one -- they are published under the old cycle's number, which vpic_hip_step's early sort ignores, and switch nothing on)
      hipLaunchKernelGGL(publish_counter_kernel, dim3(1), dim3(256), 0, e->stream, s.crossed_host_dev, s.crossed_dev, (unsigned)(s.n_cycle - (fuse ? 1 : 0)),
                         fuse ? ~0ull : (unsigned long long)e->knobs.follow_from * (unsigned long long)make_tile_k(e->gk).ntiles);
      if (e->time_kernels) { (void)hipEventRecord(s.ev[1], e->stream); s.push_timed = true; }
    }
    VH_CHECK(hipGetLastError());
  }
  // Movers can only be left behind on an absorbing face or one that belongs to another domain
  // (move_p.c:124-128); without such a face the count is known to be zero.
  if (!e->can_strand || async) { s.partition_valid = false; return 0; }   // async: the count stays on the device
  // the mover count decides what boundary_p does next: read it back
  VH_CHECK(hipMemcpyAsync(e->host_counters, s.nm_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
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
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < np; idx += (long long)gridDim.x * 256) {   // (a species may hold close to 2^31 particles)
    if (p.i[idx] < 0) continue;                          // a dead slot (engine.h, Species::n_holes)
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

// ---- center_p / uncenter_p: species_advance/standard/center_p.cxx:9-71, uncenter_p.cxx:5-71 ------
// (uncenter_p runs once at initialisation, center_p before particle dumps.)
namespace vpichip {
template <bool UNCENTER>
__global__ __launch_bounds__(256)
void center_p_kernel(ParticlesK p, const float4 *__restrict__ fi, float args_qdt_2mc, int np) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= np || p.i[idx] < 0) return;
  const float qdt_2mc = UNCENTER ? -args_qdt_2mc : args_qdt_2mc;
  const float qdt_4mc = UNCENTER ? (float)(-0.5 * args_qdt_2mc) : (float)(0.5 * args_qdt_2mc);
  const float one = 1.f, one_third = 1. / 3., two_fifteenths = 2. / 15.;
  const float dx = p.dx[idx], dy = p.dy[idx], dz = p.dz[idx];
  const float4 *f = fi + (size_t)p.i[idx] * 5;
  const float4 fe_x = f[0], fe_y = f[1], fe_z = f[2], fb0 = f[3];
  const float2 fb1 = *reinterpret_cast<const float2 *>(f + 4);
  const float hax = qdt_2mc * ((fe_x.x + dy * fe_x.y) + dz * (fe_x.z + dy * fe_x.w));
  const float hay = qdt_2mc * ((fe_y.x + dz * fe_y.y) + dx * (fe_y.z + dz * fe_y.w));
  const float haz = qdt_2mc * ((fe_z.x + dx * fe_z.y) + dy * (fe_z.z + dx * fe_z.w));
  const float cbx = fb0.x + dx * fb0.y, cby = fb0.z + dy * fb0.w, cbz = fb1.x + dz * fb1.y;
  float ux = p.ux[idx], uy = p.uy[idx], uz = p.uz[idx], v0, v1, v2, v3, v4;
  if (!UNCENTER) { ux += hax; uy += hay; uz += haz; }
  v0 = qdt_4mc / sqrtf(one + (ux * ux + (uy * uy + uz * uz)));
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
  if (UNCENTER) { ux += hax; uy += hay; uz += haz; }
  p.ux[idx] = ux; p.uy[idx] = uy; p.uz[idx] = uz;
}

int k_center_p(Engine *e, Species &s, bool uncenter) {
  if (s.np == 0) return 0;
  const vpic_hip_grid_t &g = e->grid;
  const float qdt_2mc = (float)(0.5 * s.q_m * g.dt / g.cvac);       // uncenter_p.cxx:172
  const unsigned nb = (unsigned)((s.np + 255) / 256);
  if (uncenter)
    hipLaunchKernelGGL(center_p_kernel<true>, dim3(nb), dim3(256), 0, e->stream, s.p, reinterpret_cast<const float4 *>(e->fi), qdt_2mc, (int)s.np);
  else
    hipLaunchKernelGGL(center_p_kernel<false>, dim3(nb), dim3(256), 0, e->stream, s.p, reinterpret_cast<const float4 *>(e->fi), qdt_2mc, (int)s.np);
  VH_CHECK(hipGetLastError());
  return 0;
}
}  // namespace vpichip

// ---- hydro moments: species_advance/standard/hydro_p.c:24-176 ----------------
// The particle is time-centred as in center_p (half E kick, half Boris rotation -- with the
// reference's double-precision pieces kept: sqrt in double, the series factor in double), then its
// 14 moments are spread trilinearly over the 8 nodes of its cell.  Sums are float atomics.
namespace vpichip {
__global__ __launch_bounds__(256)
void accumulate_hydro_p_kernel(float *__restrict__ h0, ParticlesK p, const float4 *__restrict__ fi, int np,
                               float qdt_2mc, float qdt_4mc2, float c, float r8V, float mc_q, int sy, int sz) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= np || p.i[idx] < 0) return;
  float dx = p.dx[idx], dy = p.dy[idx], dz = p.dz[idx];
  const int ii = p.i[idx];
  float ux = p.ux[idx], uy = p.uy[idx], uz = p.uz[idx];
  const float q = p.q[idx];
  const float4 *f = fi + (size_t)ii * 5;
  const float4 fe_x = f[0], fe_y = f[1], fe_z = f[2], fb0 = f[3];
  const float2 fb1 = *reinterpret_cast<const float2 *>(f + 4);
  float vx, vy, vz, ke_mc, w0, w1, w2, w3, w4, w5, w6, w7;
  ux += qdt_2mc * ((fe_x.x + dy * fe_x.y) + dz * (fe_x.z + dy * fe_x.w));
  uy += qdt_2mc * ((fe_y.x + dz * fe_y.y) + dx * (fe_y.z + dz * fe_y.w));
  uz += qdt_2mc * ((fe_z.x + dx * fe_z.y) + dy * (fe_z.z + dx * fe_z.w));
  w5 = fb0.x + dx * fb0.y; w6 = fb0.z + dy * fb0.w; w7 = fb1.x + dz * fb1.y;
  ke_mc = ux * ux + uy * uy + uz * uz;
  vz = (float)sqrt((double)(1.f + ke_mc));                                   // hydro_p.c:86
  ke_mc *= c / (vz + 1.f);
  vz = c / vz;
  w0 = qdt_4mc2 * vz;
  w1 = w5 * w5 + w6 * w6 + w7 * w7;
  w2 = w0 * w0 * w1;
  w3 = (float)((double)w0 * (1. + (1. / 3.) * (double)w2 * (1. + 0.4 * (double)w2)));   // hydro_p.c:92
  w4 = w3 / (1.f + w1 * w3 * w3); w4 += w4;
  w0 = ux + w3 * (uy * w7 - uz * w6);
  w1 = uy + w3 * (uz * w5 - ux * w7);
  w2 = uz + w3 * (ux * w6 - uy * w5);
  ux += w4 * (w1 * w7 - w2 * w6);
  uy += w4 * (w2 * w5 - w0 * w7);
  uz += w4 * (w0 * w6 - w1 * w5);
  vx = ux * vz; vy = uy * vz; vz *= uz;
  w0 = r8V * q; dx *= w0; w1 = w0 + dx; w0 -= dx;
  w3 = 1.f + dy; w2 = w0 * w3; w3 *= w1; dy = 1.f - dy; w0 *= dy; w1 *= dy;
  w7 = 1.f + dz; w4 = w0 * w7; w5 = w1 * w7; w6 = w2 * w7; w7 *= w3;
  dz = 1.f - dz; w0 *= dz; w1 *= dz; w2 *= dz; w3 *= dz;
  float *h = h0 + (size_t)ii * 16;
#define ACCUM_HYDRO(hh, wn) do {                                                                 \
    float *m = (hh); float w = (wn);                                                             \
    atomicAdd(m + 0, w * vx); atomicAdd(m + 1, w * vy); atomicAdd(m + 2, w * vz); atomicAdd(m + 3, w); \
    w *= mc_q; const float ax = w * ux, ay = w * uy, az = w * uz;                                \
    atomicAdd(m + 4, ax); atomicAdd(m + 5, ay); atomicAdd(m + 6, az); atomicAdd(m + 7, w * ke_mc); \
    atomicAdd(m + 8, ax * vx); atomicAdd(m + 9, ay * vy); atomicAdd(m + 10, az * vz);             \
    atomicAdd(m + 11, ay * vz); atomicAdd(m + 12, az * vx); atomicAdd(m + 13, ax * vy);           \
  } while (0)
  ACCUM_HYDRO(h, w0);
  ACCUM_HYDRO(h + 16, w1);
  ACCUM_HYDRO(h + 16 * (size_t)sy, w2);
  ACCUM_HYDRO(h + 16 * (size_t)(sy + 1), w3);
  ACCUM_HYDRO(h + 16 * (size_t)sz, w4);
  ACCUM_HYDRO(h + 16 * (size_t)(sz + 1), w5);
  ACCUM_HYDRO(h + 16 * (size_t)(sz + sy), w6);
  ACCUM_HYDRO(h + 16 * (size_t)(sz + sy + 1), w7);
#undef ACCUM_HYDRO
}

// The same sums from a cell-sorted species: one thread per voxel walks its particles (partition[v] ..
// partition[v+1]) and keeps the 8 x 14 node contributions of the cell in registers; 112 atomics per
// occupied CELL instead of per particle (32 ppc: 32 x fewer), and the interpolator is read once.
// Every particle contributes the very same 112 products as in the kernel above.
__global__ __launch_bounds__(256)
void accumulate_hydro_cells_kernel(float *__restrict__ h0, ParticlesK p, const float4 *__restrict__ fi,
                                   const int *__restrict__ partition, int nv,
                                   float qdt_2mc, float qdt_4mc2, float c, float r8V, float mc_q, int sy, int sz) {
  const int ii = blockIdx.x * 256 + threadIdx.x;
  if (ii >= nv) return;
  const int first = partition[ii], last = partition[ii + 1];
  if (first >= last) return;
  const float4 *f = fi + (size_t)ii * 5;
  const float4 fe_x = f[0], fe_y = f[1], fe_z = f[2], fb0 = f[3];
  const float2 fb1 = *reinterpret_cast<const float2 *>(f + 4);
  float S[8][14];
#pragma unroll
  for (int n = 0; n < 8; n++)
#pragma unroll
    for (int k = 0; k < 14; k++) S[n][k] = 0.f;
#pragma unroll 1
  for (int idx = first; idx < last; idx++) {
    float dx = p.dx[idx], dy = p.dy[idx], dz = p.dz[idx];
    float ux = p.ux[idx], uy = p.uy[idx], uz = p.uz[idx];
    const float q = p.q[idx];
    float vx, vy, vz, ke_mc, w[8], w5, w6, w7, t0, t1, t2, t3, t4;
    ux += qdt_2mc * ((fe_x.x + dy * fe_x.y) + dz * (fe_x.z + dy * fe_x.w));
    uy += qdt_2mc * ((fe_y.x + dz * fe_y.y) + dx * (fe_y.z + dz * fe_y.w));
    uz += qdt_2mc * ((fe_z.x + dx * fe_z.y) + dy * (fe_z.z + dx * fe_z.w));
    w5 = fb0.x + dx * fb0.y; w6 = fb0.z + dy * fb0.w; w7 = fb1.x + dz * fb1.y;
    ke_mc = ux * ux + uy * uy + uz * uz;
    vz = (float)sqrt((double)(1.f + ke_mc));
    ke_mc *= c / (vz + 1.f);
    vz = c / vz;
    t0 = qdt_4mc2 * vz;
    t1 = w5 * w5 + w6 * w6 + w7 * w7;
    t2 = t0 * t0 * t1;
    t3 = (float)((double)t0 * (1. + (1. / 3.) * (double)t2 * (1. + 0.4 * (double)t2)));
    t4 = t3 / (1.f + t1 * t3 * t3); t4 += t4;
    t0 = ux + t3 * (uy * w7 - uz * w6);
    t1 = uy + t3 * (uz * w5 - ux * w7);
    t2 = uz + t3 * (ux * w6 - uy * w5);
    ux += t4 * (t1 * w7 - t2 * w6);
    uy += t4 * (t2 * w5 - t0 * w7);
    uz += t4 * (t0 * w6 - t1 * w5);
    vx = ux * vz; vy = uy * vz; vz *= uz;
    w[0] = r8V * q; dx *= w[0]; w[1] = w[0] + dx; w[0] -= dx;
    w[3] = 1.f + dy; w[2] = w[0] * w[3]; w[3] *= w[1]; dy = 1.f - dy; w[0] *= dy; w[1] *= dy;
    w[7] = 1.f + dz; w[4] = w[0] * w[7]; w[5] = w[1] * w[7]; w[6] = w[2] * w[7]; w[7] *= w[3];
    dz = 1.f - dz; w[0] *= dz; w[1] *= dz; w[2] *= dz; w[3] *= dz;
#pragma unroll
    for (int n = 0; n < 8; n++) {
      float wn = w[n];
      S[n][0] += wn * vx; S[n][1] += wn * vy; S[n][2] += wn * vz; S[n][3] += wn;
      wn *= mc_q; const float ax = wn * ux, ay = wn * uy, az = wn * uz;
      S[n][4] += ax; S[n][5] += ay; S[n][6] += az; S[n][7] += wn * ke_mc;
      S[n][8] += ax * vx; S[n][9] += ay * vy; S[n][10] += az * vz;
      S[n][11] += ay * vz; S[n][12] += az * vx; S[n][13] += ax * vy;
    }
  }
  float *h = h0 + (size_t)ii * 16;
#pragma unroll
  for (int n = 0; n < 8; n++) {
    float *m = h + 16 * (size_t)((n & 1) + ((n >> 1) & 1) * sy + (n >> 2) * sz);
#pragma unroll
    for (int k = 0; k < 14; k++) atomicAdd(m + k, S[n][k]);
  }
}

int k_accumulate_hydro_p(Engine *e, Species &s) {
  if (ensure_hydro(e)) return 1;
  if (s.np == 0) return 0;
  const vpic_hip_grid_t &g = e->grid;
  const float qdt_2mc = 0.5 * s.q_m * g.dt / g.cvac;                          // hydro_p.c:49-53
  const float qdt_4mc2 = 0.25 * s.q_m * g.dt / (g.cvac * g.cvac);
  const float r8V = 0.125 * g.rdx * g.rdy * g.rdz;
  const float mc_q = g.cvac / s.q_m;
  // from a few particles per voxel on, the per-cell kernel wins by far; it needs the species sorted
  // (sorting only reorders the array, as the reference's own sort_p does)
  const bool by_cell = s.np >= 4 * (int64_t)e->gk.nv && s.nm == 0 && !e->knobs.hydro_per_particle;
  if (by_cell) {
    if (!s.partition_valid && k_sort_p(e, s)) return 1;
    hipLaunchKernelGGL(accumulate_hydro_cells_kernel, dim3((unsigned)((e->gk.nv + 255) / 256)), dim3(256), 0, e->stream,
                       reinterpret_cast<float *>(e->hydro), s.p, reinterpret_cast<const float4 *>(e->fi), s.partition, e->gk.nv,
                       qdt_2mc, qdt_4mc2, g.cvac, r8V, mc_q, e->gk.sy, e->gk.sz);
  } else
  hipLaunchKernelGGL(accumulate_hydro_p_kernel, dim3((unsigned)((s.np + 255) / 256)), dim3(256), 0, e->stream,
                     reinterpret_cast<float *>(e->hydro), s.p, reinterpret_cast<const float4 *>(e->fi), (int)s.np,
                     qdt_2mc, qdt_4mc2, g.cvac, r8V, mc_q, e->gk.sy, e->gk.sz);
  VH_CHECK(hipGetLastError());
  return 0;
}
}  // namespace vpichip

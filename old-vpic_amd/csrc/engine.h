// engine.h -- internal declarations of the resident MI355X engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "vpic_hip.h"

namespace vpichip {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define VH_CHECK(expr)                                                                     \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      vpichip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)
#define VH_FAIL(...) do { vpichip::set_error(__VA_ARGS__); return 1; } while (0)

// ---- device-side views ---------------------------------------------------------------------
// Geometry every kernel needs.  Voxel v = x + sy*(y + (ny+2)*z), sy = nx+2, sz = sy*(ny+2)
// (src/util/util_base.h:158-159).
struct GridK {
  int nx, ny, nz, sy, sz, nv;
  int fbc[6], pbc[6];
  int rank;
};

// Yee fields, struct-of-arrays over voxels.  Component order = float order inside field_t
// (src/field_advance/field_advance.h:159-171).
enum { F_EX = 0, F_EY, F_EZ, F_DIV_E_ERR, F_CBX, F_CBY, F_CBZ, F_DIV_B_ERR,
       F_TCAX, F_TCAY, F_TCAZ, F_RHOB, F_JFX, F_JFY, F_JFZ, F_RHOF, F_NCOMP };
enum { M_EMATX = 0, M_EMATY, M_EMATZ, M_NMAT, M_FMATX, M_FMATY, M_FMATZ, M_CMAT, M_NCOMP };
struct FieldsK {
  float *c[F_NCOMP];
  uint16_t *m[M_NCOMP];   // material ids; all null when the table holds a single material
};

// Particles of one species, struct-of-arrays, kept (approximately) cell-sorted.
struct ParticlesK {
  float *dx, *dy, *dz;
  int *i;
  float *ux, *uy, *uz, *q;
};

// What the cell-crossing path of advance_p needs and the push loop does not (push.hip); one
// record per species in device memory, written when the species is created.
struct DrainParams {
  int nx, ny, nz, sy, sz, rank, max_nm;
  unsigned mul_sz;                   // v / sz == umulhi(v, mul_sz) >> (shifts >> 8) for 0 <= v < 2^31 (magic_div)
  int pbc[6];
  unsigned mul_sy, shifts;           // v / sy == umulhi(v, mul_sy) >> (shifts & 255)
  vpic_particle_mover_t *pm;
  int *nm_counter;
};

// Division of a non-negative int by a fixed d >= 2 as multiply-high and shift: with
// p = 31 + ceil(log2 d) and M = floor(2^p / d) + 1 < 2^32, floor(v * M / 2^p) == v / d for all
// 0 <= v < 2^31 (the error M*d - 2^p is in (0, d], so v * error < 2^p).
inline void magic_div(unsigned d, unsigned &mul, unsigned &shift) {
  unsigned l = 0;
  while ((1ull << l) < d) l++;
  const unsigned p = 31 + l;
  mul = (unsigned)(((unsigned long long)1 << p) / d + 1);
  shift = p - 32;
}

// device counters (ints, Engine::counters): [0] movers of a species beyond MAX_SPECIES (drop-in twins), [8..13]
// injectors per face, [14] holes, [15] fills, [16+s] np of species s while particles are exchanged, [48+s] movers of
// species s (written by advance_p and by the injection), [80] species that received charge, [81] overflow flags,
// [82] movers parked because their message was full, [96+s] dead slots of species s (removals of the resident exchange)
enum { C_NM = 0, C_DISORDER = 1, C_LOCAL = 2, C_SEND = 8, C_HOLES = 14, C_FILLS = 15, C_NP = 16, C_NMS = 48, C_CHARGED = 80, C_OVER = 81, C_RETRY = 82, C_NHOLE = 96, C_TOTAL = 128 };
constexpr int MAX_SPECIES = 32;
constexpr int HEADER_BASE = 256, MAX_HEADERS = 192;   // Engine::host_counters: [0, C_TOTAL) the counters, [HEADER_BASE, + 4 x MAX_HEADERS) message headers
constexpr int RETRY_CAP = 1 << 16;                     // movers a step may park because a message was full (vpic_hip_exchange_*)
// C_OVER bits: 1 more movers than a round's kernels were launched for (the rest waits for the next round), 2 a message was
// full (the movers are parked and offered again), 4 a species ran out of particle slots, 8 of mover slots, 16 more
// parked movers than RETRY_CAP (4, 8, 16: particles were lost)
static_assert(C_NMS + MAX_SPECIES == C_CHARGED && C_NP + MAX_SPECIES == C_NMS && C_NHOLE + MAX_SPECIES == C_TOTAL, "counter layout");     // species tables handed to kernels by value (boundary_p), per-species host slots

struct Species {
  float q_m = 0;
  float q_max = 0;                   // largest |charge| of a macro-particle the host has seen go into this species (the fixed-point scale of the deterministic mode)
  int64_t np = 0, max_np = 0, nm = 0, max_nm = 0;
  // Dead slots among [0, np): the device-resident exchange removes a particle by marking its slot (i = -1) instead of
  // back-filling from the end of the array (boundary_p.c:264), so that the tile order survives and nothing moves under a
  // push that is still to come; every kernel that walks the array skips them, the next sort drops them (np -= n_holes).
  int64_t n_holes = 0;
  ParticlesK p{}, aux{};             // aux: second buffer for the out-of-place sort
  DrainParams *drain_k = nullptr;
  // adaptive sorting (vpic_hip_step, sort_interval < 0): events around the last push [0,1] and sort [2,3],
  // cost of a sort, and sum / number of the push times since the last sort (ms)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool push_timed = false, sort_timed = false, sorted_once = false;
  double t_sort = 0, t_sum = 0; int n_push = 0;
  int64_t *tag = nullptr, *tag2 = nullptr, *tag_aux = nullptr, *tag2_aux = nullptr;
  bool has_tags = false;             // tags all zero until a non-zero one is uploaded
  double t_last = 0, growth_first = 0; int n_cycle = 0;   // adaptive sorting: see sort_due
  // (one set per sort flavour, see flavour_cost: [0] by cell within a tile or by voxel, [1] by tile only)
  double t_hist[2][32] = {{0}}; int n_hist[2] = {0, 0};                // ... the push times of earlier cycles by position in the cycle
  double s_hist[2][33] = {{0}}; int sorted_after = 0;          // ... what a sort cost after n pushes (more steps, more disorder), pushes before the last sort
  double c_hist[2][34] = {{0}}; double prev_sum = 0;           // ... measured cost per step (sort included) of whole cycles of n pushes; push time of the last whole cycle
  bool wide_window = false;          // advance_p instance with the double-precision LDS window (crossing-heavy species; push.hip)
  unsigned *crossed_dev = nullptr, *crossed_host = nullptr, *crossed_host_dev = nullptr;   // particles that left their cell in the last advance_p (device word, pinned mirror)
  int64_t np_pushed_last = 0;        // particles of the previous advance_p launch (denominator of the crossing fraction)
  bool chargeless = false;           // every particle has q == 0 (tracer copies): advance_p skips all deposition
  vpic_particle_mover_t *pm = nullptr;
  int *nm_dev = nullptr;             // this species' mover counter in Engine::counters
  int *partition = nullptr;          // nv+1, valid after sort_p
  bool partition_valid = false;
  // TILE order (crossing-heavy species under the adaptive sort policy; particles.hip, push.hip): the array is grouped by
  // 4x4x4-cell tile, cell by cell within a tile; tpart[tile * 64 + cell] is where that cell's particles began at the
  // sort.  Particles appended since then sit behind n_sorted.
  int *tpart = nullptr; int64_t tpart_count = 0;
  int *ttail = nullptr;              // the same for the particles appended since (regrouped by tile before every advance_p: k_tail_sort)
  bool tail_sorted = false;
  // The histogram of the NEXT sort taken by the push that precedes it (round 3): advance_p counts every particle's final cell
  // in the tile order's keys (LDS counters per window cell, flushed with the accumulators), so that the sort starts at its
  // scan.  hist_request: asked for by the step driver (the next step sorts); hist_valid: hist[] describes the array as it is
  // (anything that adds, removes or moves particles afterwards clears it and the sort counts for itself).
  int *hist = nullptr; int64_t hist_count = 0; bool hist_request = false, hist_valid = false;
  bool tail_regrouped = false;       // this step's push takes the appended particles by tile (decided by its first launch)
  // THE SORT INSIDE THE PUSH (round 3, push.hip: advance_p_kernel<.., SORT>).  A sort by tile and cell that finds the counts of
  // the push before it (hist_valid) moves nothing: k_sort_p sets fuse_pending, and the next k_advance_p writes every particle
  // it pushes to its SORTED place in the second buffer -- places from the scan of hist[] (tpart2: where every key begins in
  // the new order; Engine::sort_next: the cursors) -- then swaps the buffers.  The order is that of the cells BEFORE that
  // push (what hist[] counted).  crossed_host[2]: cursors that did not end where the next key begins (0 when the counts
  // matched the array; checked by the species' next push, which fails loudly otherwise).
  int *tpart2 = nullptr; int64_t tpart2_count = 0; bool fuse_pending = false;
  bool phase_pending = false;        // vpic_hip_advance_p_phase: the first launch ran, the interior tiles are still to be pushed
  bool tile_valid = false, adaptive = false;   // adaptive: the engine's own policy asks for the sorts (vpic_hip_sort_due)
  int64_t n_sorted = 0;
  double cross_frac = 0;          // fraction of the particles that left their cell in the last advance_p (one launch behind)
  bool coarse_sorted = false;     // the last tile sort was by tile only
  bool coarse_order = false;      // tile sorts group this species by tile only (particles.hip)
  // ... chosen by measurement when the engine's sort policy times the cycles: cost per step of whole cycles in either
  // flavour ([0] by cell within a tile, [1] by tile only; 0 = not on record), cycles since the flavour last changed
  double flavour_cost[2] = {0, 0}; int flavour_cycles = 0;
  // ... inside the push or before it: decided by MEASUREMENT (engine.hip: sort_and_push).  A species whose cells move as one (cold
  // beams) lands in its new order in long runs and the sorting launch beats sort + push (29.7 ms against 20 + 16.8 at 256^3 x
  // 64 ppc); one whose particles have spread (the same deck from step ~120 on) writes runs of two and loses (60 against 20 +
  // 18.5).  ms of this species' sort + push in either way (0: not on record), the pair of events of the measurement under way
  float sort_push_ms[2] = {0, 0}; hipEvent_t sp_ev[2] = {nullptr, nullptr}; int sp_kind = -1; bool sp_last = true;
  // ... and of the push that counted for the sort (the last, slowest plain launch of the cycle): the yardstick that says whether
  // sorting before the push is worth a first try -- 20 + 17 ms against 27-30 inside the push where the counting launch took
  // 17.6 (ratio 1.6: no), 20 + 18.5 against 60 where it took 24 (2.5: yes)
  float hint_push_ms = 0; hipEvent_t hp_ev[2] = {nullptr, nullptr}; bool hp_pending = false;
  int64_t early_sorts = 0;        // sorts vpic_hip_step made ahead of a fixed interval because the deposits had begun to miss the windows
  bool tile_unbalanced = false;   // the fullest tile alone would keep its workgroup busy several times longer than a balanced launch takes
};

// tiles of TILE_EDGE^3 cells over the interior (the last one of an axis may be partial)
constexpr int TILE_EDGE = 4, TILE_CELLS = TILE_EDGE * TILE_EDGE * TILE_EDGE;
struct TileK {
  int sy, sz, ntx, nty, ntz, ntiles;
  unsigned mul_sy, sh_sy, mul_sz, sh_sz;     // magic_div of the voxel strides
};
TileK make_tile_k(const GridK &g);

// sort key of a voxel: its own index (the reference's order, sort_p.c:48-58), or tile-major (TILE): tile by tile,
// cell by cell within the tile
template <bool TILE>
__device__ __forceinline__ int sort_key(int voxel, const TileK &t) {
  if (!TILE) return voxel;
  const int cz = (int)(__umulhi((unsigned)voxel, t.mul_sz) >> t.sh_sz), rem = voxel - cz * t.sz;
  const int cy = (int)(__umulhi((unsigned)rem, t.mul_sy) >> t.sh_sy), cx = rem - cy * t.sy;
  // particles live in interior voxels (1..n); an index in a ghost layer (the reference's sort_p takes any voxel) is
  // counted with the nearest interior cell's tile instead of indexing outside the tables
  const int x = min(max(cx - 1, 0), 4 * t.ntx - 1), y = min(max(cy - 1, 0), 4 * t.nty - 1), z = min(max(cz - 1, 0), 4 * t.ntz - 1);
  const int tile = ((z >> 2) * t.nty + (y >> 2)) * t.ntx + (x >> 2);
  return tile * TILE_CELLS + ((z & 3) << 4 | (y & 3) << 2 | (x & 3));
}

// Environment knobs (experiments and tests; none is needed in production), read ONCE when the engine is created
// (tools/README.md lists them): no entry point of the hot path calls getenv.
struct Knobs {
  int window = 0;                  // VPIC_HIP_WINDOW: 0 unset, 't' tile order, 'w' / 'n' the reference's order with the wide / narrow row window
  int tile_coarse = -1;            // VPIC_HIP_TILE_COARSE: -1 unset, 0 / 1
  long long tail_sort_min = 4096;  // VPIC_HIP_TAIL_SORT_MIN
  bool no_tail_sort = false;       // VPIC_HIP_NO_TAIL_SORT
  int iters = 0;                   // VPIC_HIP_ITERS (row windows: passes per wavefront)
  int ablate = 0;                  // VPIC_HIP_ABLATE (honoured by builds with -DVPIC_HIP_ABLATION only)
  bool policy_debug = false;       // VPIC_HIP_POLICY_DEBUG
  int follow = -1;                 // VPIC_HIP_FOLLOW=0|1: the tile window never / always follows the tile's particles (default: once deposits miss)
  int fuse_in_step = -1;           // VPIC_HIP_SORT_IN_PUSH=0|1: a species that is due is never / always (where it can be) sorted inside its push (Species::fuse_pending); default: whichever measured cheaper (engine.hip: sort_and_push)
  bool old_sort = false;           // VPIC_HIP_OLD_SORT: the wavefront-level count / scatter kernels of rounds 1-2 (A/B timing)
  int stage = -1;                  // VPIC_HIP_STAGE=0|1: advance_p never / always parks a pass's positions until its crossers are done (default: hot species only; push.hip)
  int follow_from = 16;            // VPIC_HIP_FOLLOW_FROM: missed runs per tile in one launch from which the windows follow (tuning)
  bool early_sort = true;         // VPIC_HIP_EARLY_SORT=0: vpic_hip_step keeps to the deck's sort interval whatever the deposits miss (A/B timing)
  int unload_tiled = 1;            // VPIC_HIP_UNLOAD_TILED: clear_jf + unload_accumulator 0 one thread per voxel through L1 / L2 (rounds 2-3), 2 through LDS tiles, 1 (default) tiles on grids large enough to fill the chip with them
  int field_tiles = 0;             // VPIC_HIP_FIELD_TILES: advance_b / advance_e (one material) 0 (default) one thread per voxel through L1 / L2, 2 through LDS tiles, 1 tiles on grids large enough to fill the chip with them -- measured TWICE as slow (fields.hip)
  bool rho_per_particle = false, hydro_per_particle = false;   // VPIC_HIP_RHO_PER_PARTICLE, VPIC_HIP_HYDRO_PER_PARTICLE
};
Knobs read_knobs();

struct Engine {
  int device = 0;
  Knobs knobs;
  hipStream_t stream = nullptr;
  vpic_hip_grid_t grid{};
  GridK gk{};
  FieldsK f{};
  float *field_block = nullptr;      // one allocation backing f.c[]
  uint16_t *mat_block = nullptr;
  vpic_material_coefficient_t *mc = nullptr;
  int n_mat = 0;
  vpic_interpolator_t *fi = nullptr;
  vpic_accumulator_t *acc = nullptr;
  std::vector<Species> species;
  bool engine_order = false;          // vpic_hip_set_sort_order: sorts asked for through the ABI may use the engine's own order (TILE)
  bool push_fast = false;             // advance_p arithmetic: false = the reference's scalar pipeline bit for bit, true = FAST (push.hip)
  bool can_strand = false;           // some face absorbs or belongs to another domain: advance_p may leave movers
  // Deterministic accumulation (vpic_hip_set_accumulation; push_device.h, Window<4>): deposits are summed as 64-bit fixed-point
  // integers in acc64 (12 words per voxel, value = word / acc_scale); acc_finalize rounds the sums into the float accumulator
  // (`acc`, what unload_accumulator and the host see) once per step.  rho64: the same for accumulate_rho_p.
  bool det_acc = false, acc64_dirty = false;
  double acc_scale = 0, acc_qref = 0;
  unsigned long long *acc64 = nullptr, *rho64 = nullptr;

  // scratch
  void *stage = nullptr; size_t stage_bytes = 0;       // AoS <-> SoA staging
  int *counters = nullptr;                             // small device ints (mover count, ...)
  void *acc_block = nullptr;                           // allocation behind `acc`
  bool time_kernels = false;                           // adaptive sorting in use: time every sort_p / advance_p with events
  int *host_miss = nullptr;                            // pinned: sampled descent count (vpic_hip_measure_disorder)
  void *hydro = nullptr; float *hydro_buf[2] = {nullptr, nullptr};   // hydro_t[nv] + face messages, allocated on first use
  int *host_counters = nullptr;                        // pinned mirror
  double *dsum = nullptr; double *host_dsum = nullptr; // reduction partials
  size_t dsum_count = 0;
  int *sort_next = nullptr;                            // nv+1
  int *scan_tmp = nullptr; size_t scan_tmp_count = 0;
  float *face_buf[2] = {nullptr, nullptr}; size_t face_buf_count = 0;
  vpic_particle_injector_t *send_buf[6] = {}; int64_t send_cap = 0;
  // custom particle boundary handlers of the maxwellian_reflux kind (src/boundary/maxwellian_reflux.c)
  struct Reflux { int code; float ut_para[MAX_SPECIES], ut_perp[MAX_SPECIES]; };
  std::vector<Reflux> reflux;
  uint32_t reflux_seed = 0; uint32_t reflux_calls = 0;
  float *reflux_draws = nullptr; int64_t reflux_draws_n = 0;
  double *emit_draws = nullptr; int64_t emit_draws_n = 0;      // test mode: the emission model's draws per emitted-particle slot   // test mode: the handlers' draws per particle index
  vpic_particle_injector_t *local_buf = nullptr; int64_t local_cap = 0;   // injectors that re-enter this same domain
  int32_t send_count[6] = {};
  int *hole_list = nullptr, *fill_list = nullptr, *tail_flag = nullptr; int64_t list_cap = 0;
  // device-resident exchange (vpic_hip_exchange_*): species table, message table, whether tail_flag is all zero
  void *sp_table_dev = nullptr, *sp_table_host = nullptr, *xmsg_dev = nullptr, *xmsg_host = nullptr; bool tail_clean = false; unsigned xmsg_turn = 0;
  void *retry_buf = nullptr;          // RETRY_CAP parked movers {mover, species}
  // tiles that touch a face shared with another domain [0] (every particle that can leave the domain this step is in one of
  // them or in the appended tail) and the others [1]: vpic_hip_advance_p_phase pushes the first group, lets the exchange
  // start, and pushes the second behind it
  int *tile_list[2] = {nullptr, nullptr}; int tile_list_n[2] = {0, 0};

  hipEvent_t step_done[4] = {}; int64_t steps_enqueued = 0;   // vpic_hip_step: the host stays at most two steps ahead of the device
  // profiling
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  std::vector<int64_t> ev_particles;
  std::vector<char> ev_kind;                 // 1: a launch that sorts as it pushes (Species::fuse_pending): booked apart
  std::vector<int> ev_species;               // whose launch (vpic_hip_profile_read_species)
  double prof_sp_ms[MAX_SPECIES] = {}; int64_t prof_sp_launches[MAX_SPECIES] = {}, prof_sp_particles[MAX_SPECIES] = {};   // the plain launches, by species
  double prof_ms = 0; int64_t prof_launches = 0, prof_particles = 0;
  double prof_sort_ms = 0; int64_t prof_sort_launches = 0, prof_sort_particles = 0;
};

int ensure_stage(Engine *e, size_t bytes);
int acc_prepare_det(Engine *e);      // deterministic mode: allocate / scale the fixed-point accumulator before a kernel adds to it
int acc_finalize(Engine *e);         // ... and round its sums into the float accumulator before anything reads that
void host_will_read(const void *p, size_t bytes);   // see engine.hip: called before a HIP copy reads / writes caller memory
void host_will_write(void *p, size_t bytes);
constexpr int PUSH_TILE = 64;          // particles per wavefront pass of the push kernel (push.hip)
constexpr int64_t PARTICLE_PAD = 2048;   // allocation granularity of the particle arrays (every kernel guards its accesses by np)
int alloc_particles(ParticlesK &p, int64_t n);

// kernels (one translation unit each)
int k_fields_from_aos(Engine *e, const vpic_field_t *host);
int k_fields_to_aos(Engine *e, vpic_field_t *host);
int k_load_interpolator(Engine *e);
int k_unload_accumulator(Engine *e);
int k_clear_jf(Engine *e);
int k_clear_jf_unload_accumulator(Engine *e);
int k_synchronize_jf_local(Engine *e);
int k_local_adjust_jf(Engine *e);
int k_synchronize_jf_self(Engine *e, int axis);
int k_advance_b(Engine *e, float frac);
int k_advance_e(Engine *e, int part = 0);   // part: see AdvanceEParams (fields.hip)
int k_energy_f(Engine *e, double *en6);
int k_clear_rhof(Engine *e);
int k_accumulate_rho_p(Engine *e, Species &s);
int k_rho_count(const Engine *e, int dir);
int k_pack_rho(Engine *e, int dir, float *buf);
int k_unpack_rho(Engine *e, int dir, const float *buf);
int k_local_adjust_rho(Engine *e);
int k_synchronize_rho_self(Engine *e, int axis);
int k_synchronize_rho_local(Engine *e);
int k_compute_div_e_err(Engine *e);
int k_compute_rhob(Engine *e);
int k_rms_div_e_err_local(Engine *e, double *local2);
int k_rms_div_b_err_local(Engine *e, double *local2);
int k_clean_div_e(Engine *e);
int k_compute_div_b_err(Engine *e);
int k_clean_div_b(Engine *e);
int k_compute_curl_b(Engine *e);
int k_synchronize_tang_e_norm_b_local(Engine *e, double *err);
int k_msg_count(const Engine *e, int kind, int dir);
int k_pack_msg(Engine *e, int kind, int dir, float *buf);
int k_unpack_msg(Engine *e, int kind, int dir, const float *buf);
int k_err_begin(Engine *e);
int k_err_read(Engine *e, double *err);
int k_local_adjust_tang_e_norm_b(Engine *e);
int k_synchronize_tang_e_norm_b_self(Engine *e, int axis);
int ensure_hydro(Engine *e);
int k_dump_gather(Engine *e, int what, int layout, const int32_t *words, int nwords, int sx, int sy, int sz, void *out, size_t out_bytes);
int k_clear_hydro(Engine *e);
int k_accumulate_hydro_p(Engine *e, Species &s);
int k_local_adjust_hydro(Engine *e);
int k_hydro_count(const Engine *e, int dir);
int k_pack_hydro(Engine *e, int dir, float *buf);
int k_unpack_hydro(Engine *e, int dir, const float *buf);
int k_synchronize_hydro_self(Engine *e, int axis);
int k_synchronize_hydro_local(Engine *e);
int k_face_count(const Engine *e, int dir);
int k_pack_face(Engine *e, int dir, float *buf, int what);       // what: 0 tang_b, 1 jf
int k_unpack_face(Engine *e, int dir, const float *buf, int what);

int k_particles_from_aos(Engine *e, Species &s, const vpic_particle_t *host, int64_t n_new, int64_t at = 0);
int k_particles_to_aos(Engine *e, Species &s, vpic_particle_t *host, int64_t cap, int64_t from = 0, int64_t count = -1);
int k_load_maxwellian(Engine *e, Species &s, int ppc, unsigned seed, float q, float ux, float uy, float uz, float vth);
int k_energy_p(Engine *e, Species &s, double *energy);
int k_center_p(Engine *e, Species &s, bool uncenter);
int k_sort_p(Engine *e, Species &s, bool tile_order = false, bool may_fuse = false);   // may_fuse: the caller pushes the species next (see Species::fuse_pending)
int k_sort_scan(Engine *e, const int *counts, int *starts, int n1);                  // exclusive scan of counts[0..n1) into starts[] and Engine::sort_next[]
int k_sort_check(Engine *e, Species &s, const int *starts, int n1);                    // every cursor ended where the next key begins? (crossed_host[2]; the next push fails loudly otherwise)
int k_sort_finish(Engine *e, Species &s, bool tile_order, bool coarse);                // what follows a sort's scatter (buffers swapped, bookkeeping)
int k_tail_sort(Engine *e, Species &s);
int k_measure_disorder(Engine *e, Species &s, int slot);
int k_boundary_p_pack(Engine *e);
int k_exchange_begin(Engine *e);
int k_exchange_pack(Engine *e, void *const msg[6], const int32_t cap[6], int mover_cap, uint32_t species_mask = ~0u);
int k_compact(Engine *e, Species &s);          // drop the dead slots of a species (a sort in the order it is in)
int k_species_reserve(Engine *e, Species &s, int64_t max_np, int64_t max_nm);
int k_exchange_inject(Engine *e, const void *msg, int cap);
int k_exchange_finish(Engine *e, const void *const *recv, int n_recv, int32_t *headers_out, int32_t *flags_out);
int k_advance_p(Engine *e, Species &s, bool async = false, int phase = 0);   // async: the mover count stays on the device (vpic_hip_exchange_*); phase: see vpic_hip_advance_p_phase
int k_boundary_p_inject(Engine *e, const vpic_particle_injector_t *inj, int n, const int64_t *tags = nullptr);
int k_emit(Engine *e, int sp, const int32_t *host_components, int n, int n_emit, float ut_perp, float ut_para, float coef, float thresh, unsigned seed);
int k_inject_aged(Engine *e, const vpic_particle_injector_t *host_inj, const int64_t *host_tags, int n);
int k_accumulate_rhob(Engine *e, const vpic_particle_t *host, int64_t n, float q_scale);

// g.pbc[face] with a run-time face: a dynamically indexed kernel-argument array would be copied
// to scratch memory, a select chain stays in scalar registers.
__device__ __forceinline__ int pbc_of(const GridK &g, int face) {
  return face == 0 ? g.pbc[0] : face == 1 ? g.pbc[1] : face == 2 ? g.pbc[2]
       : face == 3 ? g.pbc[3] : face == 4 ? g.pbc[4] : g.pbc[5];
}

// XCD-aware logical block id: blocks are dealt round-robin over the 8 XCDs (b and b+8 share
// one), so give each XCD a contiguous range of logical blocks -- neighbouring tiles then share
// an L2.  Speed only: any mapping is correct.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb) {
  const unsigned per = nb >> 3;
  if (per == 0 || (nb & 7)) return b;
  return (b & 7) * per + (b >> 3);
}

}  // namespace vpichip

// the opaque handle of the C ABI is the engine itself
struct vpic_hip_engine : public vpichip::Engine {};

// engine.hip -- the resident-engine half of the C ABI (include/vpic_hip.h): device memory, host
// mirror transfers, the per-step driver and HIP-event profiling.  Kernels live in push.hip,
// fields.hip and particles.hip.
#include "engine.h"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <cstdlib>
#include <cmath>

namespace vpichip {

static thread_local char g_error[1024] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// Host arrays handed to this library may be demand-paged mirrors of the caller (the deck host protects its
// arrays while the engine owns the data, old-vpic_amd/host).  A HIP copy engine that meets a protected page does not
// raise SIGSEGV, it faults the GPU -- so EVERY entry point that is given a host array announces the range here
// before any copy touches it, and the registered callback makes it resident (and writable, for downloads).
static vpic_hip_host_access_fn g_host_access = nullptr;
void host_will_read(const void *p, size_t bytes) { if (g_host_access && p && bytes) g_host_access(p, bytes, 0); }
void host_will_write(void *p, size_t bytes) { if (g_host_access && p && bytes) g_host_access(p, bytes, 1); }

int ensure_stage(Engine *e, size_t bytes) {
  if (bytes <= e->stage_bytes) return 0;
  if (e->stage) (void)hipFree(e->stage);
  e->stage = nullptr; e->stage_bytes = 0;
  VH_CHECK(hipMalloc(&e->stage, bytes));
  e->stage_bytes = bytes;
  return 0;
}

Knobs read_knobs() {
  Knobs k;
  if (const char *w = getenv("VPIC_HIP_WINDOW")) k.window = (w[0] == 't' || w[0] == 'w' || w[0] == 'n') ? w[0] : 0;
  if (const char *c = getenv("VPIC_HIP_TILE_COARSE")) k.tile_coarse = atoi(c) != 0;
  if (const char *t = getenv("VPIC_HIP_TAIL_SORT_MIN")) k.tail_sort_min = atoll(t);
  k.no_tail_sort = getenv("VPIC_HIP_NO_TAIL_SORT") != nullptr;
  if (const char *it = getenv("VPIC_HIP_ITERS")) k.iters = atoi(it) > 0 ? atoi(it) : 0;
  if (const char *ab = getenv("VPIC_HIP_ABLATE")) k.ablate = atoi(ab);
  k.policy_debug = getenv("VPIC_HIP_POLICY_DEBUG") != nullptr;
  k.old_sort = getenv("VPIC_HIP_OLD_SORT") != nullptr;
  if (const char *v = getenv("VPIC_HIP_SORT_IN_PUSH")) k.fuse_in_step = atoi(v) != 0 ? 1 : 0;
  if (const char *v = getenv("VPIC_HIP_FOLLOW")) k.follow = atoi(v) != 0;
  if (const char *v = getenv("VPIC_HIP_STAGE")) k.stage = atoi(v) != 0;
  if (const char *v = getenv("VPIC_HIP_FOLLOW_FROM")) k.follow_from = atoi(v) > 0 ? atoi(v) : 16;
  if (const char *v = getenv("VPIC_HIP_EARLY_SORT")) k.early_sort = atoi(v) != 0;
  if (const char *v = getenv("VPIC_HIP_UNLOAD_TILED")) k.unload_tiled = atoi(v);
  if (const char *v = getenv("VPIC_HIP_FIELD_TILES")) k.field_tiles = atoi(v);
  k.rho_per_particle = getenv("VPIC_HIP_RHO_PER_PARTICLE") != nullptr;
  k.hydro_per_particle = getenv("VPIC_HIP_HYDRO_PER_PARTICLE") != nullptr;
  return k;
}

static int validate_grid(const vpic_hip_grid_t *g) {
  if (!g) VH_FAIL("Bad grid");
  if (g->nx < 1 || g->ny < 1 || g->nz < 1) VH_FAIL("Bad resolution %d x %d x %d", g->nx, g->ny, g->nz);
  const int64_t nv = (int64_t)(g->nx + 2) * (g->ny + 2) * (g->nz + 2);
  if (nv * 12 >= (1ll << 31)) VH_FAIL("domain of %lld voxels is too large for 32-bit voxel arithmetic", (long long)nv);
  if (!(g->dt > 0) || !(g->cvac > 0) || !(g->eps0 > 0)) VH_FAIL("Bad dt/cvac/eps0");
  for (int f = 0; f < 6; f++) {
    if (g->fbc[f] < VPIC_ABSORB_FIELDS) VH_FAIL("Bad field boundary code %d on face %d", g->fbc[f], f);
    // codes <= -3 are custom handlers (grid.h:68-69, add_boundary.c:31): each needs its parameters
    // (vpic_hip_set_maxwellian_reflux) before the first particle reaches the face
  }
  // a face that wraps onto this same domain must do so for both faces of the axis
  for (int a = 0; a < 3; a++) {
    if ((g->fbc[a] == g->rank) != (g->fbc[a + 3] == g->rank)) VH_FAIL("axis %d: field faces are periodic on one side only", a);
    if ((g->pbc[a] == g->rank) != (g->pbc[a + 3] == g->rank)) VH_FAIL("axis %d: particle faces are periodic on one side only", a);
  }
  return 0;
}

static int create(Engine *e, const vpic_hip_grid_t *g, int device) {
  if (validate_grid(g)) return 1;
  e->knobs = read_knobs();
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
    VH_FAIL("no HIP device is available: the MI355X engine has no CPU fallback");
  if (device >= 0) { VH_CHECK(hipSetDevice(device)); }
  VH_CHECK(hipGetDevice(&e->device));
  VH_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  e->grid = *g;
  GridK &k = e->gk;
  k.nx = g->nx; k.ny = g->ny; k.nz = g->nz;
  k.sy = g->nx + 2; k.sz = k.sy * (g->ny + 2); k.nv = k.sz * (g->nz + 2);
  for (int f = 0; f < 6; f++) { k.fbc[f] = g->fbc[f]; k.pbc[f] = g->pbc[f]; }
  k.rank = g->rank;
  for (int f = 0; f < 6; f++) if (g->pbc[f] != g->rank && g->pbc[f] != VPIC_REFLECT_PARTICLES) e->can_strand = true;
  const size_t nv = (size_t)k.nv;

  VH_CHECK(hipMalloc(&e->field_block, sizeof(float) * F_NCOMP * nv));
  VH_CHECK(hipMemsetAsync(e->field_block, 0, sizeof(float) * F_NCOMP * nv, e->stream));
  for (int c = 0; c < F_NCOMP; c++) e->f.c[c] = e->field_block + c * nv;
  for (int c = 0; c < M_NCOMP; c++) e->f.m[c] = nullptr;
  VH_CHECK(hipMalloc(&e->fi, sizeof(vpic_interpolator_t) * nv));
  VH_CHECK(hipMemsetAsync(e->fi, 0, sizeof(vpic_interpolator_t) * nv, e->stream));
  VH_CHECK(hipMalloc(&e->acc_block, sizeof(vpic_accumulator_t) * nv));
  e->acc = reinterpret_cast<vpic_accumulator_t *>(e->acc_block);
  VH_CHECK(hipHostMalloc(&e->host_miss, sizeof(int) * MAX_SPECIES));
  for (int k = 0; k < MAX_SPECIES; k++) e->host_miss[k] = 0;
  VH_CHECK(hipMemsetAsync(e->acc, 0, sizeof(vpic_accumulator_t) * nv, e->stream));

  VH_CHECK(hipMalloc(&e->counters, sizeof(int) * 256));
  VH_CHECK(hipMemsetAsync(e->counters, 0, sizeof(int) * 256, e->stream));
  VH_CHECK(hipHostMalloc(&e->host_counters, sizeof(int) * (HEADER_BASE + 4 * MAX_HEADERS)));   // [0, C_TOTAL) the counters, [HEADER_BASE, ...) message headers (k_exchange_finish)
  e->dsum_count = 6 * 1024;
  VH_CHECK(hipMalloc(&e->dsum, sizeof(double) * e->dsum_count));
  VH_CHECK(hipHostMalloc(&e->host_dsum, sizeof(double) * e->dsum_count));
  {  // sort scratch: one count per voxel, or per cell of every (possibly partial) tile
    const TileK tk = make_tile_k(e->gk);
    const size_t n1 = (size_t)std::max((int64_t)nv, (int64_t)tk.ntiles * TILE_CELLS) + 1;
    VH_CHECK(hipMalloc(&e->sort_next, sizeof(int) * n1));
    e->scan_tmp_count = (n1 + 1023) / 1024 + 1;
  }
  VH_CHECK(hipMalloc(&e->scan_tmp, sizeof(int) * e->scan_tmp_count));
  size_t face = 0;
  for (int d = 0; d < 3; d++) face = std::max(face, std::max(std::max((size_t)k_face_count(e, d), (size_t)k_rho_count(e, d)), (size_t)k_msg_count(e, 2, d)));
  e->face_buf_count = face;
  VH_CHECK(hipMalloc(&e->face_buf[0], sizeof(float) * face));
  VH_CHECK(hipMalloc(&e->face_buf[1], sizeof(float) * face));
  VH_CHECK(hipStreamSynchronize(e->stream));
  return 0;
}

// ---- deterministic accumulation: the 64-bit fixed-point accumulator (engine.h, push_device.h) -----------------------------
__global__ __launch_bounds__(256)
void acc_finalize_kernel(float *__restrict__ acc, unsigned long long *__restrict__ acc64, size_t n, double inv_scale) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const long long v = (long long)acc64[k];
  if (v) { acc[k] += (float)((double)v * inv_scale); acc64[k] = 0; }     // the exact sum, rounded once
}
// power of two such that a deposit of 4.2 x q_ref (the largest a particle of that charge makes) lands near 2^37: 2^-37 of it
// is the resolution, 2^14 times it still converts (|x * scale| < 2^51), 2^26 of them fit one 64-bit sum
static double fixed_scale(double largest) {
  int ex = 0;
  (void)frexp(largest, &ex);
  return ldexp(1.0, 37 - ex);
}
int acc_prepare_det(Engine *e) {
  if (!e->acc64) {
    VH_CHECK(hipMalloc(&e->acc64, sizeof(unsigned long long) * 12 * (size_t)e->gk.nv));
    VH_CHECK(hipMemsetAsync(e->acc64, 0, sizeof(unsigned long long) * 12 * (size_t)e->gk.nv, e->stream));
  }
  if (e->acc_scale == 0) {
    double q = e->acc_qref;
    if (!(q > 0)) for (auto &s : e->species) q = std::max(q, (double)s.q_max);
    if (!(q > 0)) VH_FAIL("deterministic accumulation: no macro-particle charge is known yet (give vpic_hip_set_accumulation a reference charge)");
    e->acc_scale = fixed_scale(4.2 * q);
  }
  e->acc64_dirty = true;
  return 0;
}
int acc_finalize(Engine *e) {
  if (!e->acc64 || !e->acc64_dirty) return 0;
  const size_t n = 12 * (size_t)e->gk.nv;
  hipLaunchKernelGGL(acc_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream,
                     reinterpret_cast<float *>(e->acc), e->acc64, n, 1.0 / e->acc_scale);
  VH_CHECK(hipGetLastError());
  e->acc64_dirty = false;
  return 0;
}

static void free_particles(ParticlesK &p) {
  (void)hipFree(p.dx); (void)hipFree(p.dy); (void)hipFree(p.dz); (void)hipFree(p.i);
  (void)hipFree(p.ux); (void)hipFree(p.uy); (void)hipFree(p.uz); (void)hipFree(p.q);
  p = ParticlesK{};
}

static void destroy(Engine *e) {
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto &s : e->species) {
    free_particles(s.p); free_particles(s.aux);
    (void)hipFree(s.tag); (void)hipFree(s.tag2); (void)hipFree(s.tag_aux); (void)hipFree(s.tag2_aux);
    (void)hipFree(s.pm); (void)hipFree(s.partition); (void)hipFree(s.tpart); (void)hipFree(s.tpart2); (void)hipFree(s.ttail); (void)hipFree(s.hist); (void)hipFree(s.drain_k); (void)hipFree(s.crossed_dev); (void)hipHostFree(s.crossed_host);
    for (int i = 0; i < 4; i++) if (s.ev[i]) (void)hipEventDestroy(s.ev[i]);
    for (auto ev : s.sp_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : s.hp_ev) if (ev) (void)hipEventDestroy(ev);
  }
  (void)hipFree(e->field_block); (void)hipFree(e->mat_block); (void)hipFree(e->mc);
  (void)hipFree(e->fi); (void)hipFree(e->acc_block); (void)hipHostFree(e->host_miss); (void)hipFree(e->stage); (void)hipFree(e->counters); (void)hipFree(e->hydro); (void)hipFree(e->hydro_buf[0]); (void)hipFree(e->hydro_buf[1]);
  (void)hipHostFree(e->host_counters); (void)hipFree(e->dsum); (void)hipHostFree(e->host_dsum);
  (void)hipFree(e->sort_next); (void)hipFree(e->scan_tmp);
  (void)hipFree(e->face_buf[0]); (void)hipFree(e->face_buf[1]);
  for (int f = 0; f < 6; f++) (void)hipFree(e->send_buf[f]);
  (void)hipFree(e->local_buf); (void)hipFree(e->reflux_draws); (void)hipFree(e->emit_draws);
  (void)hipFree(e->hole_list); (void)hipFree(e->fill_list); (void)hipFree(e->tail_flag);
  (void)hipFree(e->sp_table_dev); (void)hipHostFree(e->sp_table_host); (void)hipFree(e->xmsg_dev); (void)hipHostFree(e->xmsg_host);
  (void)hipFree(e->retry_buf); (void)hipFree(e->tile_list[0]); (void)hipFree(e->tile_list[1]);
  (void)hipFree(e->acc64); (void)hipFree(e->rho64);
  for (auto &ev : e->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (auto ev : e->step_done) if (ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
}

static int collect_profile(Engine *e) {
  for (size_t k = 0; k < e->ev_used; k++) {
    float ms = 0;
    VH_CHECK(hipEventSynchronize(e->ev_pool[k].second));
    VH_CHECK(hipEventElapsedTime(&ms, e->ev_pool[k].first, e->ev_pool[k].second));
    if (e->ev_kind[k]) { e->prof_sort_ms += ms; e->prof_sort_launches++; e->prof_sort_particles += e->ev_particles[k]; continue; }
    e->prof_ms += ms;
    if (e->ev_particles[k] >= 0) { e->prof_launches++; e->prof_particles += e->ev_particles[k]; }   // (< 0: second launch of a species pushed in two phases)
    const int sp = e->ev_species[k];
    if (sp >= 0 && sp < MAX_SPECIES) {
      e->prof_sp_ms[sp] += ms;
      if (e->ev_particles[k] >= 0) { e->prof_sp_launches[sp]++; e->prof_sp_particles[sp] += e->ev_particles[k]; }
    }
  }
  e->ev_used = 0;
  return 0;
}

}  // namespace vpichip

using namespace vpichip;

#define ENGINE(e) do { if (!(e)) { set_error("null engine"); return 1; } if (hipSetDevice((e)->device) != hipSuccess) { set_error("hipSetDevice failed"); return 1; } } while (0)
#define SPECIES(e, sp) do { if ((sp) < 0 || (size_t)(sp) >= (e)->species.size()) { set_error("bad species id %d", (sp)); return 1; } } while (0)

extern "C" {

const char *vpic_hip_last_error(void) { return g_error; }
void vpic_hip_set_host_access_hook(vpic_hip_host_access_fn fn) { g_host_access = fn; }

int vpic_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vpic_hip_create(vpic_hip_engine_t **out, const vpic_hip_grid_t *g, int device) {
  if (!out) { set_error("null output pointer"); return 1; }
  *out = nullptr;
  vpic_hip_engine *e = new vpic_hip_engine();
  if (create(e, g, device)) { destroy(e); delete e; return 1; }
  *out = e;
  return 0;
}

void vpic_hip_destroy(vpic_hip_engine_t *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  destroy(e);
  delete e;
}

int vpic_hip_sync(vpic_hip_engine_t *e) { ENGINE(e); VH_CHECK(hipStreamSynchronize(e->stream)); return 0; }
void *vpic_hip_stream(vpic_hip_engine_t *e) { return e ? (void *)e->stream : nullptr; }
int vpic_hip_nv(const vpic_hip_engine_t *e) { return e ? e->gk.nv : 0; }

int vpic_hip_set_fields(vpic_hip_engine_t *e, const vpic_field_t *f) {
  ENGINE(e);
  if (!f) VH_FAIL("Bad field");
  host_will_read(f, sizeof(*f) * (size_t)e->gk.nv);
  // a host array that carries material ids needs the id arrays on the device
  if (!e->f.m[0]) {
    bool any = false;
    const size_t nv = e->gk.nv;
    for (size_t v = 0; v < nv && !any; v++)
      any = f[v].ematx | f[v].ematy | f[v].ematz | f[v].nmat | f[v].fmatx | f[v].fmaty | f[v].fmatz | f[v].cmat;
    if (any) {
      VH_CHECK(hipMalloc(&e->mat_block, sizeof(uint16_t) * M_NCOMP * nv));
      for (int c = 0; c < M_NCOMP; c++) e->f.m[c] = e->mat_block + c * nv;
    }
  }
  return k_fields_from_aos(e, f);
}
int vpic_hip_get_fields(vpic_hip_engine_t *e, vpic_field_t *f) {
  ENGINE(e);
  if (!f) VH_FAIL("Bad field");
  host_will_write(f, sizeof(*f) * (size_t)e->gk.nv);
  return k_fields_to_aos(e, f);
}

static int copy_in(Engine *e, void *dst, const void *src, size_t bytes) {
  host_will_read(src, bytes);
  VH_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  return 0;
}
static int copy_out(Engine *e, void *dst, const void *src, size_t bytes) {
  host_will_write(dst, bytes);
  VH_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
  VH_CHECK(hipStreamSynchronize(e->stream));
  return 0;
}

int vpic_hip_set_interpolator(vpic_hip_engine_t *e, const vpic_interpolator_t *fi) {
  ENGINE(e); if (!fi) VH_FAIL("Bad interpolator");
  return copy_in(e, e->fi, fi, sizeof(*fi) * (size_t)e->gk.nv);
}
int vpic_hip_get_interpolator(vpic_hip_engine_t *e, vpic_interpolator_t *fi) {
  ENGINE(e); if (!fi) VH_FAIL("Bad interpolator");
  return copy_out(e, fi, e->fi, sizeof(*fi) * (size_t)e->gk.nv);
}
int vpic_hip_set_accumulator(vpic_hip_engine_t *e, const vpic_accumulator_t *a) {
  ENGINE(e); if (!a) VH_FAIL("Bad accumulator");
  if (e->acc64 && e->acc64_dirty) {   // the upload REPLACES the sums so far, the fixed-point ones included (they would be rounded over it otherwise)
    VH_CHECK(hipMemsetAsync(e->acc64, 0, sizeof(unsigned long long) * 12 * (size_t)e->gk.nv, e->stream));
    e->acc64_dirty = false;
  }
  return copy_in(e, e->acc, a, sizeof(*a) * (size_t)e->gk.nv);
}
int vpic_hip_get_accumulator(vpic_hip_engine_t *e, vpic_accumulator_t *a) {
  ENGINE(e); if (!a) VH_FAIL("Bad accumulator");
  if (acc_finalize(e)) return 1;
  return copy_out(e, a, e->acc, sizeof(*a) * (size_t)e->gk.nv);
}

int vpic_hip_set_material_coefficients(vpic_hip_engine_t *e, const vpic_material_coefficient_t *m, int n) {
  ENGINE(e);
  if (!m || n < 1) VH_FAIL("Empty material list");
  if (e->mc) (void)hipFree(e->mc);
  e->mc = nullptr;
  VH_CHECK(hipMalloc(&e->mc, sizeof(*m) * (size_t)n));
  e->n_mat = n;
  return copy_in(e, e->mc, m, sizeof(*m) * (size_t)n);
}

int vpic_hip_species_create(vpic_hip_engine_t *e, float q_m, int64_t max_np, int64_t max_nm) {
  if (!e) { set_error("null engine"); return -1; }
  if (hipSetDevice(e->device) != hipSuccess) { set_error("hipSetDevice failed"); return -1; }
  // indices are 32-bit ints throughout (sort keys, mover lists, the reference's own particle_mover_t::i)
  if (max_np < 1 || max_nm < 1 || max_np > (1ll << 31) - 8192) { set_error("Bad species sizes"); return -1; }
  Species s;
  s.q_m = q_m; s.max_np = max_np; s.max_nm = max_nm;
  const bool ok = alloc_particles(s.p, max_np) == 0 &&
                  hipMalloc(&s.pm, sizeof(vpic_particle_mover_t) * max_nm) == hipSuccess;
  if (!ok) { set_error("out of device memory for a species of %lld particles", (long long)max_np); return -1; }
  DrainParams d;
  memset(&d, 0, sizeof(d));
  d.nx = e->gk.nx; d.ny = e->gk.ny; d.nz = e->gk.nz; d.sy = e->gk.sy; d.sz = e->gk.sz; d.rank = e->gk.rank;
  d.max_nm = (int)max_nm;
  { unsigned shz, shy; magic_div((unsigned)d.sz, d.mul_sz, shz); magic_div((unsigned)d.sy, d.mul_sy, shy); d.shifts = (shz << 8) | shy; }
  for (int f = 0; f < 6; f++) d.pbc[f] = e->gk.pbc[f];
  s.nm_dev = e->counters + (e->species.size() < (size_t)MAX_SPECIES ? C_NMS + (int)e->species.size() : C_NM);
  d.pm = s.pm; d.nm_counter = s.nm_dev;
  if (hipMalloc(&s.crossed_dev, sizeof(unsigned) * 256 * 16) != hipSuccess || hipMemset(s.crossed_dev, 0, sizeof(unsigned) * 256 * 16) != hipSuccess ||
      hipHostMalloc(&s.crossed_host, sizeof(unsigned) * 8, hipHostMallocMapped) != hipSuccess ||   // [0] crossers of the last push, [1] particles of the fullest tile at the last tile sort, [2] see Species::fuse_pending, [3] runs that missed the tile windows in the last push, [4] the sort cycle that push belonged to
      hipHostGetDevicePointer((void **)&s.crossed_host_dev, s.crossed_host, 0) != hipSuccess) {
    set_error("out of memory for a species counter"); return -1;
  }
  s.crossed_host[0] = 0; s.crossed_host[1] = 0; s.crossed_host[2] = 0; s.crossed_host[3] = 0; s.crossed_host[4] = ~0u;
  (void)hipDeviceSynchronize();                          // the fill above ran on the null stream; the engine's stream does not wait for that one
  if (hipMalloc(&s.drain_k, sizeof(d)) != hipSuccess || hipMemcpy(s.drain_k, &d, sizeof(d), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("out of device memory for a species record"); return -1;
  }
  e->species.push_back(s);
  return (int)e->species.size() - 1;
}

int vpic_hip_species_set_particles(vpic_hip_engine_t *e, int sp, const vpic_particle_t *p, int64_t np) {
  ENGINE(e); SPECIES(e, sp);
  if (np < 0 || (np > 0 && !p)) VH_FAIL("Bad particle array");
  host_will_read(p, sizeof(*p) * (size_t)np);
  return k_particles_from_aos(e, e->species[sp], p, np);
}
int vpic_hip_emit(vpic_hip_engine_t *e, int sp, const int32_t *components, int n, int n_emit_per_face,
                  float ut_perp, float ut_para, float coef, float thresh_e_norm, uint32_t seed) {
  ENGINE(e); SPECIES(e, sp);
  if (n < 0 || (n > 0 && !components) || n_emit_per_face < 1 || !(coef > 0)) VH_FAIL("Bad emitter");
  host_will_read(components, sizeof(int32_t) * (size_t)n);
  if ((int64_t)n * n_emit_per_face > (1 << 28)) VH_FAIL("emitter too large");
  return n ? k_emit(e, sp, components, n, n_emit_per_face, ut_perp, ut_para, coef, thresh_e_norm, seed) : 0;
}
int vpic_hip_inject_aged(vpic_hip_engine_t *e, const vpic_particle_injector_t *inj, const int64_t *tags, int n) {
  ENGINE(e);
  if (n < 0 || (n > 0 && !inj)) VH_FAIL("Bad injector array");
  host_will_read(inj, sizeof(*inj) * (size_t)n);
  if (tags) host_will_read(tags, sizeof(int64_t) * 2 * (size_t)n);
  return n ? k_inject_aged(e, inj, tags, n) : 0;
}
int vpic_hip_accumulate_rhob(vpic_hip_engine_t *e, const vpic_particle_t *p, int64_t n, float q_scale) {
  ENGINE(e);
  if (n < 0 || (n > 0 && !p)) VH_FAIL("Bad particle array");
  host_will_read(p, sizeof(*p) * (size_t)n);
  for (int64_t k = 0; k < n; k++) if (p[k].i < 0 || p[k].i >= e->gk.nv - e->gk.sz - e->gk.sy - 1) VH_FAIL("particle %lld is not in a voxel of this grid", (long long)k);
  return n ? k_accumulate_rhob(e, p, n, q_scale) : 0;
}
int vpic_hip_set_maxwellian_reflux(vpic_hip_engine_t *e, int code, const float *ut_para, const float *ut_perp, int n_species, uint32_t seed) {
  ENGINE(e);
  if (code > -3) VH_FAIL("custom particle boundary codes are <= -3 (got %d)", code);
  if (!ut_para || !ut_perp || n_species < 1 || n_species > MAX_SPECIES) VH_FAIL("Bad reflux parameters");
  Engine::Reflux r;
  memset(&r, 0, sizeof(r));
  r.code = code;
  for (int k = 0; k < n_species; k++) { r.ut_para[k] = ut_para[k]; r.ut_perp[k] = ut_perp[k]; }
  for (auto &old : e->reflux) if (old.code == code) { old = r; e->reflux_seed = seed; return 0; }
  if (e->reflux.size() >= 4) VH_FAIL("more than 4 reflux handlers");
  e->reflux.push_back(r);
  e->reflux_seed = seed;
  return 0;
}
int vpic_hip_set_reflux_draws(vpic_hip_engine_t *e, const float *draws, int64_t n_particles) {
  ENGINE(e);
  if (n_particles < 0 || (n_particles > 0 && !draws) || n_particles > (1ll << 28)) VH_FAIL("Bad draw table");
  if (e->reflux_draws) { (void)hipFree(e->reflux_draws); e->reflux_draws = nullptr; }   // (the emitter's table is vpic_hip_set_emit_draws' to free)
  e->reflux_draws_n = 0;
  if (n_particles == 0) return 0;
  VH_CHECK(hipMalloc(&e->reflux_draws, sizeof(float) * 3 * (size_t)n_particles));
  e->reflux_draws_n = n_particles;
  return copy_in(e, e->reflux_draws, draws, sizeof(float) * 3 * (size_t)n_particles);
}
int vpic_hip_set_emit_draws(vpic_hip_engine_t *e, const double *draws, int64_t n_slots) {
  ENGINE(e);
  if (n_slots < 0 || (n_slots > 0 && !draws) || n_slots > (1ll << 26)) VH_FAIL("Bad draw table");
  if (e->emit_draws) { (void)hipFree(e->emit_draws); e->emit_draws = nullptr; }
  e->emit_draws_n = 0;
  if (n_slots == 0) return 0;
  VH_CHECK(hipMalloc(&e->emit_draws, sizeof(double) * 6 * (size_t)n_slots));
  e->emit_draws_n = n_slots;
  return copy_in(e, e->emit_draws, draws, sizeof(double) * 6 * (size_t)n_slots);
}
int vpic_hip_species_append_particles(vpic_hip_engine_t *e, int sp, const vpic_particle_t *p, int64_t n) {
  ENGINE(e); SPECIES(e, sp);
  if (n < 0 || (n > 0 && !p)) VH_FAIL("Bad particle array");
  if (n == 0) return 0;
  host_will_read(p, sizeof(*p) * (size_t)n);
  return k_particles_from_aos(e, e->species[sp], p, n, e->species[sp].np);
}
int vpic_hip_species_get_particles(vpic_hip_engine_t *e, int sp, vpic_particle_t *p, int64_t cap) {
  ENGINE(e); SPECIES(e, sp);
  if (!p && e->species[sp].np > 0) VH_FAIL("Bad particle array");
  host_will_write(p, sizeof(*p) * (size_t)e->species[sp].np);
  return k_particles_to_aos(e, e->species[sp], p, cap);
}
int vpic_hip_species_get_particles_range(vpic_hip_engine_t *e, int sp, vpic_particle_t *p, int64_t from, int64_t count) {
  ENGINE(e); SPECIES(e, sp);
  if (!p || count < 0) VH_FAIL("Bad particle array");
  host_will_write(p, sizeof(*p) * (size_t)count);
  return count ? k_particles_to_aos(e, e->species[sp], p, count, from, count) : 0;
}
int vpic_hip_species_load_maxwellian(vpic_hip_engine_t *e, int sp, int ppc, uint32_t seed, float q,
                                     float ux, float uy, float uz, float vth) {
  ENGINE(e); SPECIES(e, sp);
  return k_load_maxwellian(e, e->species[sp], ppc, seed, q, ux, uy, uz, vth);
}
int64_t vpic_hip_species_np(vpic_hip_engine_t *e, int sp) {
  if (!e || sp < 0 || (size_t)sp >= e->species.size()) return -1;
  return e->species[sp].np - e->species[sp].n_holes;        // live particles (dead slots: engine.h, Species::n_holes)
}
int vpic_hip_species_capacity(vpic_hip_engine_t *e, int sp, int64_t *extent, int64_t *max_np, int64_t *max_nm) {
  ENGINE(e); SPECIES(e, sp);
  if (extent) *extent = e->species[sp].np;
  if (max_np) *max_np = e->species[sp].max_np;
  if (max_nm) *max_nm = e->species[sp].max_nm;
  return 0;
}
int vpic_hip_species_reserve(vpic_hip_engine_t *e, int sp, int64_t max_np, int64_t max_nm) {
  ENGINE(e); SPECIES(e, sp);
  return k_species_reserve(e, e->species[sp], max_np, max_nm);
}
int64_t vpic_hip_species_nm(vpic_hip_engine_t *e, int sp) {
  if (!e || sp < 0 || (size_t)sp >= e->species.size()) return -1;
  return e->species[sp].nm;
}
int vpic_hip_species_set_movers(vpic_hip_engine_t *e, int sp, const vpic_particle_mover_t *pm, int64_t nm) {
  ENGINE(e); SPECIES(e, sp);
  Species &s = e->species[sp];
  if (nm < 0 || nm > s.max_nm || (nm > 0 && !pm)) VH_FAIL("Bad mover list");
  if (nm > 0 && copy_in(e, s.pm, pm, sizeof(*pm) * (size_t)nm)) return 1;
  s.nm = nm;
  return 0;
}
int vpic_hip_species_get_movers(vpic_hip_engine_t *e, int sp, vpic_particle_mover_t *pm, int64_t cap) {
  ENGINE(e); SPECIES(e, sp);
  Species &s = e->species[sp];
  if (cap < s.nm) VH_FAIL("mover buffer holds %lld, species has %lld", (long long)cap, (long long)s.nm);
  if (s.nm == 0) return 0;
  host_will_write(pm, sizeof(*pm) * (size_t)s.nm);
  if (copy_out(e, pm, s.pm, sizeof(*pm) * (size_t)s.nm)) return 1;
  // boundary_p.c:168-176 assumes pm[n].i > pm[n-1].i
  std::sort(pm, pm + s.nm, [](const vpic_particle_mover_t &a, const vpic_particle_mover_t &b) { return a.i < b.i; });
  return 0;
}
int vpic_hip_species_get_partition(vpic_hip_engine_t *e, int sp, int32_t *partition) {
  ENGINE(e); SPECIES(e, sp);
  Species &s = e->species[sp];
  if (!s.partition || !s.partition_valid) VH_FAIL("partition is only valid right after sort_p");
  return copy_out(e, partition, s.partition, sizeof(int) * ((size_t)e->gk.nv + 1));
}

#ifdef VPIC_HIP_ABLATION
// timing experiments (-DVPIC_HIP_ABLATION builds only; tools/ablate_once.py): switch parts of advance_p off for the NEXT launches
int vpic_hip_debug_set_ablate(vpic_hip_engine_t *e, int bits) { ENGINE(e); e->knobs.ablate = bits; return 0; }
#endif
// test hook (not part of include/vpic_hip.h): overwrite the pinned word the sort's tile_max kernel publishes -- the fullest tile's
// particle count, which k_advance_p reads without waiting for the device (tests/test_gpu_tiles.py: a value that lands between
// the two launches of a phased push)
int vpic_hip_debug_poke_tile_max(vpic_hip_engine_t *e, int sp, unsigned value) {
  ENGINE(e); SPECIES(e, sp);
  if (!e->species[sp].crossed_host) VH_FAIL("no counter word");
  e->species[sp].crossed_host[1] = value;
  return 0;
}
int vpic_hip_species_get_tile_partition(vpic_hip_engine_t *e, int sp, int32_t *tpart, int64_t *count) {
  ENGINE(e); SPECIES(e, sp);
  Species &s = e->species[sp];
  const int64_t n1 = (int64_t)make_tile_k(e->gk).ntiles * TILE_CELLS + 1;
  if (count) *count = n1;
  if (!tpart) return 0;
  if (!s.tpart || !s.tile_valid || s.tpart_count < n1) VH_FAIL("the tile partition is only valid while the species is in the engine's order");
  return copy_out(e, tpart, s.tpart, sizeof(int) * (size_t)n1);
}

int vpic_hip_load_interpolator(vpic_hip_engine_t *e) { ENGINE(e); return k_load_interpolator(e); }
int vpic_hip_clear_accumulators(vpic_hip_engine_t *e) {
  ENGINE(e);
  VH_CHECK(hipMemsetAsync(e->acc, 0, sizeof(vpic_accumulator_t) * (size_t)e->gk.nv, e->stream));
  if (e->acc64 && e->acc64_dirty) {
    VH_CHECK(hipMemsetAsync(e->acc64, 0, sizeof(unsigned long long) * 12 * (size_t)e->gk.nv, e->stream));
    e->acc64_dirty = false;
  }
  return 0;
}
// Accumulation mode.  0 (default): float sums -- LDS / global float atomics, order-dependent at the 1e-7 level.  1:
// DETERMINISTIC -- every deposit is rounded to 64-bit fixed point and summed as an integer, so the accumulators (and with
// them jf, rhof and everything downstream) are bit-identical from run to run whatever the array order, the scheduling or
// the decomposition's message order; the reference is reproducible by construction (reduce_accumulators.cxx:37-55: private
// accumulators reduced in a fixed order).  q_ref: |charge| of a typical macro-particle (sets the fixed-point scale; <= 0:
// the largest charge the host has put into a species so far).
int vpic_hip_set_accumulation(vpic_hip_engine_t *e, int mode, double q_ref) {
  ENGINE(e);
  if (mode != 0 && mode != 1) VH_FAIL("Bad accumulation mode %d", mode);
  if (acc_finalize(e)) return 1;
  e->det_acc = mode == 1;
  e->acc_qref = q_ref > 0 ? q_ref : 0;
  e->acc_scale = 0;                                   // chosen when the first kernel adds
  return 0;
}
int vpic_hip_reduce_accumulators(vpic_hip_engine_t *e) { ENGINE(e); return 0; }
int vpic_hip_unload_accumulator(vpic_hip_engine_t *e) { ENGINE(e); return k_unload_accumulator(e); }
int vpic_hip_set_push_mode(vpic_hip_engine_t *e, int mode) {
  ENGINE(e);
  if (mode != VPIC_HIP_PUSH_EXACT && mode != VPIC_HIP_PUSH_FAST) VH_FAIL("Bad push mode %d", mode);
  e->push_fast = mode == VPIC_HIP_PUSH_FAST;
  return 0;
}
// Deterministic accumulation without a window (a species that is not in tile order: Window<5>) is a global 64-bit atomic per
// deposit -- twelve times the time of a tiled launch.  A species that lost the tile order to a sort by voxel (the per-voxel
// moment kernels of a hydro dump or a cleaning step ask for one) is put back into it before it is pushed.
static bool wants_tile_order(const Engine *e, const Species &s);
static int order_for_deterministic_push(Engine *e, Species &s) {
  if (e->det_acc && !s.chargeless && !s.tile_valid && s.np > 0 && wants_tile_order(e, s)) return k_sort_p(e, s, true);
  return 0;
}
// a push that counts for the sort that follows (Species::hist_request) is timed with a pair of events: sort_and_push's yardstick
static int push_timing_hinted(Engine *e, Species &s) {
  const bool timed = s.hist_request && !e->time_kernels && e->knobs.fuse_in_step < 0;
  if (timed) {
    if (!s.hp_ev[0]) for (auto &ev : s.hp_ev) VH_CHECK(hipEventCreate(&ev));
    // (read here, a cycle later: when the sort that follows is decided the host is a step ahead of this launch's end)
    if (s.hp_pending && hipEventQuery(s.hp_ev[1]) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, s.hp_ev[0], s.hp_ev[1]) == hipSuccess && ms > 0) s.hint_push_ms = ms;
    }
    s.hp_pending = false;
    VH_CHECK(hipEventRecord(s.hp_ev[0], e->stream));
  }
  if (k_advance_p(e, s)) return 1;
  if (timed) { VH_CHECK(hipEventRecord(s.hp_ev[1], e->stream)); s.hp_pending = true; }
  return 0;
}
int vpic_hip_advance_p(vpic_hip_engine_t *e, int sp) { ENGINE(e); SPECIES(e, sp); if (order_for_deterministic_push(e, e->species[sp])) return 1; return push_timing_hinted(e, e->species[sp]); }
int vpic_hip_advance_p_async(vpic_hip_engine_t *e, int sp) {
  ENGINE(e); SPECIES(e, sp);
  if (sp >= MAX_SPECIES) VH_FAIL("the device-resident exchange serves %d species", MAX_SPECIES);
  return k_advance_p(e, e->species[sp], true);
}
int vpic_hip_exchange_begin(vpic_hip_engine_t *e) {
  ENGINE(e);
  // (the pushes of the exchange that begins here must not sort: a sort drops dead slots and the particle counts go to the device now)
  for (auto &s : e->species) if (order_for_deterministic_push(e, s)) return 1;
  return k_exchange_begin(e);
}
int vpic_hip_exchange_pack(vpic_hip_engine_t *e, void *const msg[6], const int32_t cap[6], int mover_cap) { ENGINE(e); if (!msg || !cap) VH_FAIL("Bad message table"); return k_exchange_pack(e, msg, cap, mover_cap); }
int vpic_hip_exchange_pack_species(vpic_hip_engine_t *e, uint32_t species_mask, void *const msg[6], const int32_t cap[6], int mover_cap) {
  ENGINE(e); if (!msg || !cap) VH_FAIL("Bad message table");
  return k_exchange_pack(e, msg, cap, mover_cap, species_mask);
}
int vpic_hip_advance_p_phase(vpic_hip_engine_t *e, int sp, int phase) {
  ENGINE(e); SPECIES(e, sp);
  if (sp >= MAX_SPECIES) VH_FAIL("the device-resident exchange serves %d species", MAX_SPECIES);
  if (phase < 0 || phase > 2) VH_FAIL("Bad phase %d", phase);
  return k_advance_p(e, e->species[sp], true, phase);
}
int vpic_hip_exchange_inject(vpic_hip_engine_t *e, const void *msg, int cap) { ENGINE(e); return k_exchange_inject(e, msg, cap); }
int vpic_hip_exchange_finish(vpic_hip_engine_t *e, const void *const *recv, int n_recv, int32_t *headers, int32_t *flags) {
  ENGINE(e); if (n_recv > 0 && (!recv || !headers)) VH_FAIL("Bad message list");
  return k_exchange_finish(e, recv, n_recv, headers, flags);
}
// Which order a sort asked for through the ABI produces.  The reference's (by voxel, sort_p.c:48-58, with partition[])
// unless the caller has left the choice to the engine -- vpic_hip_set_sort_order(e, 1), or vpic_hip_sort_due consulted for
// the species (the engine's own sort policy) --: then charged species are grouped by TILE (engine.h), the order advance_p
// is fastest on (one workgroup per tile, the tile and its halo as LDS window: push.hip).  Nothing but the array order and
// partition[] depends on the choice.  VPIC_HIP_WINDOW=tile forces tiles (tests, experiments), =wide / =narrow the
// reference's order and that row window.
static bool wants_tile_order(const Engine *e, const Species &s) {
  if (s.np > ((int64_t)1 << 30)) return false;   // one launch: 32-bit byte offsets into the arrays
  // (a chargeless species -- tracer copies -- is pushed without a window in any order; grouped by tile its interpolator
  // gathers stay local, and the sort by tile only costs a quarter of the sort by voxel on a hot species: k_sort_p)
  if (e->knobs.window == 't') return true;
  if (e->knobs.window == 'w' || e->knobs.window == 'n') return false;
  // a grid thinner than a tile on some axis (2-D decks: ny = 1) would give every workgroup a quarter tile or less of work;
  // the row windows of the reference's order serve those
  if (std::min(e->gk.nx, std::min(e->gk.ny, e->gk.nz)) < TILE_EDGE) return false;
  // see k_advance_p: one tile held far more than its share at the last tile sort; every 32nd sort looks again
  if (s.tile_unbalanced && (s.n_cycle & 31) != 31) return false;
  return e->engine_order || s.adaptive;
}
int vpic_hip_set_sort_order(vpic_hip_engine_t *e, int order) {
  ENGINE(e);
  if (order != 0 && order != 1) VH_FAIL("Bad sort order %d (0: the reference's, 1: the engine's choice)", order);
  e->engine_order = order == 1;
  if (order == 0) for (auto &s : e->species) s.adaptive = false;   // an explicit request for the reference's order stands until the engine's policy is consulted again
  return 0;
}
int vpic_hip_sort_p(vpic_hip_engine_t *e, int sp) { ENGINE(e); SPECIES(e, sp); return k_sort_p(e, e->species[sp], wants_tile_order(e, e->species[sp])); }
// A species that is due, sorted and pushed: the sort INSIDE the push (Species::fuse_pending) or before it, whichever took less
// time for this species when it was last tried (engine.h, Species::sort_push_ms) -- both are timed with a pair of events that is
// read when the species is sorted next (the host is two steps ahead of the device at most), the loser is tried again every eighth
// sort; sorting BEFORE the push is tried for the first time only where the launch that sorted was slow against the one before it.  between(): what the caller does between the two (hints for the push).
static int sort_and_push(Engine *e, Species &s, bool may_fuse, const std::function<int(Species &)> &between) {
  const bool tile = wants_tile_order(e, s);
  may_fuse = may_fuse && e->knobs.fuse_in_step != 0;
  const bool measured = may_fuse && e->knobs.fuse_in_step < 0 && tile && s.hist_valid && !e->time_kernels;     // (a sort that has to count for itself is neither of the two)
  if (measured) {
    if (s.sp_kind >= 0 && hipEventQuery(s.sp_ev[1]) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, s.sp_ev[0], s.sp_ev[1]) == hipSuccess && ms > 0) s.sort_push_ms[s.sp_kind] = ms;
      s.sp_kind = -1;
    }
    if (s.hp_pending && hipEventQuery(s.hp_ev[1]) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, s.hp_ev[0], s.hp_ev[1]) == hipSuccess && ms > 0) s.hint_push_ms = ms;
      s.hp_pending = false;
    }
    if (s.sp_kind < 0) {                                   // (otherwise: the last measurement has not come back -- as last time)
      if (s.sort_push_ms[1] == 0) s.sp_last = true;        // [1] inside the push, [0] before it
      else if (s.sort_push_ms[0] == 0)
        // before the push for the first time: when the launch that sorted took more than 1.9 x the one that counted for it (see
        // Species::hint_push_ms; a short run of a cold deck never pays for the experiment), or at the sixteenth sort at the latest
        s.sp_last = !((s.hint_push_ms > 0 && s.sort_push_ms[1] > 1.9f * s.hint_push_ms) || (s.n_cycle & 15) == 15);
      else { s.sp_last = s.sort_push_ms[1] <= s.sort_push_ms[0]; if ((s.n_cycle & 7) == 7) s.sp_last = !s.sp_last; }
      if (e->knobs.policy_debug) fprintf(stderr, "sort and push: inside %.2f ms, before %.2f ms, the push that counted %.2f ms -> %s\n", (double)s.sort_push_ms[1], (double)s.sort_push_ms[0], (double)s.hint_push_ms, s.sp_last ? "inside" : "before");
    }
    may_fuse = s.sp_last;
    if (s.sp_kind < 0) {
      if (!s.sp_ev[0]) for (auto &ev : s.sp_ev) VH_CHECK(hipEventCreate(&ev));
      VH_CHECK(hipEventRecord(s.sp_ev[0], e->stream));
    }
  }
  if (k_sort_p(e, s, tile, may_fuse)) return 1;
  const bool timing = measured && s.sp_kind < 0;
  const int kind = s.fuse_pending ? 1 : 0;
  if (between(s)) return 1;
  if (k_advance_p(e, s)) return 1;
  if (timing) { VH_CHECK(hipEventRecord(s.sp_ev[1], e->stream)); s.sp_kind = kind; }
  return 0;
}
int vpic_hip_sort_advance_p(vpic_hip_engine_t *e, int sp) {
  ENGINE(e); SPECIES(e, sp);
  return sort_and_push(e, e->species[sp], true, [](Species &) { return 0; });
}
int vpic_hip_energy_p(vpic_hip_engine_t *e, int sp, double *energy) {
  ENGINE(e); SPECIES(e, sp);
  if (!energy) VH_FAIL("Bad energy");
  return k_energy_p(e, e->species[sp], energy);
}
int vpic_hip_center_p(vpic_hip_engine_t *e, int sp) { ENGINE(e); SPECIES(e, sp); return k_center_p(e, e->species[sp], false); }
int vpic_hip_uncenter_p(vpic_hip_engine_t *e, int sp) { ENGINE(e); SPECIES(e, sp); return k_center_p(e, e->species[sp], true); }
int vpic_hip_clear_jf(vpic_hip_engine_t *e) { ENGINE(e); return k_clear_jf(e); }
int vpic_hip_clear_jf_unload_accumulator(vpic_hip_engine_t *e) { ENGINE(e); return k_clear_jf_unload_accumulator(e); }
int vpic_hip_clear_hydro(vpic_hip_engine_t *e) { ENGINE(e); return k_clear_hydro(e); }
int vpic_hip_accumulate_hydro_p(vpic_hip_engine_t *e, int sp) { ENGINE(e); SPECIES(e, sp); return k_accumulate_hydro_p(e, e->species[sp]); }
int vpic_hip_synchronize_hydro(vpic_hip_engine_t *e) { ENGINE(e); return k_synchronize_hydro_local(e); }
int vpic_hip_local_adjust_hydro(vpic_hip_engine_t *e) { ENGINE(e); return k_local_adjust_hydro(e); }
int vpic_hip_synchronize_hydro_self(vpic_hip_engine_t *e, int axis) { ENGINE(e); if (axis < 0 || axis > 2) VH_FAIL("Bad axis"); return k_synchronize_hydro_self(e, axis); }
int vpic_hip_hydro_count(const vpic_hip_engine_t *e, int dir) { if (!e || dir < 0 || dir > 5) return -1; return k_hydro_count(e, dir); }
int vpic_hip_pack_hydro(vpic_hip_engine_t *e, int dir, void *b) { ENGINE(e); if (dir < 0 || dir > 5 || !b) VH_FAIL("Bad face message"); return k_pack_hydro(e, dir, (float *)b); }
int vpic_hip_unpack_hydro(vpic_hip_engine_t *e, int dir, const void *b) { ENGINE(e); if (dir < 0 || dir > 5 || !b) VH_FAIL("Bad face message"); return k_unpack_hydro(e, dir, (const float *)b); }
int vpic_hip_set_hydro(vpic_hip_engine_t *e, const vpic_hydro_t *h) {
  ENGINE(e); if (!h) VH_FAIL("Bad hydro");
  if (ensure_hydro(e)) return 1;
  return copy_in(e, e->hydro, h, sizeof(*h) * (size_t)e->gk.nv);
}
int vpic_hip_get_hydro(vpic_hip_engine_t *e, vpic_hydro_t *h) {
  ENGINE(e); if (!h) VH_FAIL("Bad hydro");
  if (ensure_hydro(e)) return 1;
  return copy_out(e, h, e->hydro, sizeof(*h) * (size_t)e->gk.nv);
}
int vpic_hip_dump_gather(vpic_hip_engine_t *e, int what, int layout, const int32_t *words, int nwords,
                         int sx, int sy, int sz, void *out, size_t out_bytes) {
  ENGINE(e);
  if (what < 0 || what > 1 || layout < 0 || layout > 2 || !out) VH_FAIL("Bad dump request");
  if (sx < 1 || sy < 1 || sz < 1 || e->gk.nx % sx || e->gk.ny % sy || e->gk.nz % sz) VH_FAIL("stride must be an integer factor of the cell count");
  const int limit = what == VPIC_HIP_DUMP_FIELDS ? 24 : 16;
  if (layout == VPIC_HIP_DUMP_BAND) {
    if (!words || nwords < 1 || nwords > 32) VH_FAIL("Bad variable list");
    for (int k = 0; k < nwords; k++) if (words[k] < 0 || words[k] >= limit) VH_FAIL("Bad variable");
  }
  host_will_write(out, out_bytes);
  return k_dump_gather(e, what, layout, words, nwords, sx, sy, sz, out, out_bytes);
}
int vpic_hip_clear_rhof(vpic_hip_engine_t *e) { ENGINE(e); return k_clear_rhof(e); }
int vpic_hip_accumulate_rho_p(vpic_hip_engine_t *e, int sp) { ENGINE(e); SPECIES(e, sp); return k_accumulate_rho_p(e, e->species[sp]); }
int vpic_hip_synchronize_rho(vpic_hip_engine_t *e) { ENGINE(e); return k_synchronize_rho_local(e); }
int vpic_hip_local_adjust_rho(vpic_hip_engine_t *e) { ENGINE(e); return k_local_adjust_rho(e); }
int vpic_hip_synchronize_rho_self(vpic_hip_engine_t *e, int axis) { ENGINE(e); if (axis < 0 || axis > 2) VH_FAIL("Bad axis"); return k_synchronize_rho_self(e, axis); }
int vpic_hip_rho_count(const vpic_hip_engine_t *e, int dir) { if (!e || dir < 0 || dir > 5) return -1; return k_rho_count(e, dir); }
int vpic_hip_pack_rho(vpic_hip_engine_t *e, int dir, void *b) { ENGINE(e); if (dir < 0 || dir > 5 || !b) VH_FAIL("Bad face message"); return k_pack_rho(e, dir, (float *)b); }
int vpic_hip_unpack_rho(vpic_hip_engine_t *e, int dir, const void *b) { ENGINE(e); if (dir < 0 || dir > 5 || !b) VH_FAIL("Bad face message"); return k_unpack_rho(e, dir, (const float *)b); }
int vpic_hip_compute_rhob(vpic_hip_engine_t *e) { ENGINE(e); return k_compute_rhob(e); }
int vpic_hip_compute_curl_b(vpic_hip_engine_t *e) { ENGINE(e); return k_compute_curl_b(e); }
int vpic_hip_synchronize_tang_e_norm_b(vpic_hip_engine_t *e, double *err) { ENGINE(e); if (!err) VH_FAIL("Bad err"); return k_synchronize_tang_e_norm_b_local(e, err); }
int vpic_hip_face_message_count(const vpic_hip_engine_t *e, int kind, int dir) { if (!e || dir < 0 || dir > 5 || kind < 0 || kind > 2) return -1; return k_msg_count(e, kind, dir); }
int vpic_hip_pack_face_message(vpic_hip_engine_t *e, int kind, int dir, void *b) { ENGINE(e); if (dir < 0 || dir > 5 || kind < 0 || kind > 2 || !b) VH_FAIL("Bad face message"); return k_pack_msg(e, kind, dir, (float *)b); }
int vpic_hip_unpack_face_message(vpic_hip_engine_t *e, int kind, int dir, const void *b, double *err) {
  ENGINE(e); if (dir < 0 || dir > 5 || kind < 0 || kind > 2 || !b) VH_FAIL("Bad face message");
  if (kind != 2) return k_unpack_msg(e, kind, dir, (const float *)b);
  if (!err) VH_FAIL("Bad err");
  if (k_err_begin(e) || k_unpack_msg(e, kind, dir, (const float *)b)) return 1;
  return k_err_read(e, err);
}
int vpic_hip_local_adjust_tang_e_norm_b(vpic_hip_engine_t *e) { ENGINE(e); return k_local_adjust_tang_e_norm_b(e); }
int vpic_hip_synchronize_tang_e_norm_b_self(vpic_hip_engine_t *e, int axis, double *err) {
  ENGINE(e); if (axis < 0 || axis > 2 || !err) VH_FAIL("Bad argument");
  if (k_err_begin(e) || k_synchronize_tang_e_norm_b_self(e, axis)) return 1;
  return k_err_read(e, err);
}
int vpic_hip_compute_div_e_err(vpic_hip_engine_t *e) { ENGINE(e); return k_compute_div_e_err(e); }
int vpic_hip_clean_div_e(vpic_hip_engine_t *e) { ENGINE(e); return k_clean_div_e(e); }
int vpic_hip_compute_div_b_err(vpic_hip_engine_t *e) { ENGINE(e); return k_compute_div_b_err(e); }
int vpic_hip_clean_div_b(vpic_hip_engine_t *e) { ENGINE(e); return k_clean_div_b(e); }
int vpic_hip_rms_div_e_err_local(vpic_hip_engine_t *e, double *l2) { ENGINE(e); if (!l2) VH_FAIL("Bad output"); return k_rms_div_e_err_local(e, l2); }
int vpic_hip_rms_div_b_err_local(vpic_hip_engine_t *e, double *l2) { ENGINE(e); if (!l2) VH_FAIL("Bad output"); return k_rms_div_b_err_local(e, l2); }
int vpic_hip_compute_rms_div_e_err(vpic_hip_engine_t *e, double *rms) {
  ENGINE(e); if (!rms) VH_FAIL("Bad output");
  double l2[2];
  if (k_rms_div_e_err_local(e, l2)) return 1;
  *rms = e->grid.eps0 * sqrt(l2[0] / l2[1]);
  return 0;
}
int vpic_hip_compute_rms_div_b_err(vpic_hip_engine_t *e, double *rms) {
  ENGINE(e); if (!rms) VH_FAIL("Bad output");
  double l2[2];
  if (k_rms_div_b_err_local(e, l2)) return 1;
  *rms = e->grid.eps0 * sqrt(l2[0] / l2[1]);
  return 0;
}
int vpic_hip_synchronize_jf(vpic_hip_engine_t *e) { ENGINE(e); return k_synchronize_jf_local(e); }
int vpic_hip_local_adjust_jf(vpic_hip_engine_t *e) { ENGINE(e); return k_local_adjust_jf(e); }
int vpic_hip_synchronize_jf_self(vpic_hip_engine_t *e, int axis) {
  ENGINE(e);
  if (axis < 0 || axis > 2) VH_FAIL("bad axis %d", axis);
  return k_synchronize_jf_self(e, axis);
}
int vpic_hip_advance_b(vpic_hip_engine_t *e, float frac) { ENGINE(e); return k_advance_b(e, frac); }
int vpic_hip_advance_e(vpic_hip_engine_t *e) { ENGINE(e); return k_advance_e(e); }
int vpic_hip_advance_e_part(vpic_hip_engine_t *e, int part) { ENGINE(e); if (part < 0 || part > 2) VH_FAIL("Bad part"); return k_advance_e(e, part); }
int vpic_hip_stream_wait_event(vpic_hip_engine_t *e, void *event) { ENGINE(e); VH_CHECK(hipStreamWaitEvent(e->stream, (hipEvent_t)event, 0)); return 0; }
int vpic_hip_energy_f(vpic_hip_engine_t *e, double *en6) {
  ENGINE(e);
  if (!en6) VH_FAIL("Bad energy");
  return k_energy_f(e, en6);
}

int vpic_hip_boundary_p_pack(vpic_hip_engine_t *e) { ENGINE(e); return k_boundary_p_pack(e); }
// ---- staging helpers for hosts whose transport moves host memory (MPI without GPU-aware buffers) ----
void *vpic_hip_device_alloc(vpic_hip_engine_t *e, size_t bytes) {
  if (!e || hipSetDevice(e->device) != hipSuccess) return nullptr;
  void *p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { set_error("out of device memory (%zu bytes)", bytes); return nullptr; }
  return p;
}
void vpic_hip_device_free(vpic_hip_engine_t *e, void *p) { if (e && p) { (void)hipSetDevice(e->device); (void)hipFree(p); } }
int vpic_hip_copy_to_host(vpic_hip_engine_t *e, void *host, const void *dev, size_t bytes) {
  ENGINE(e); if (bytes && (!host || !dev)) VH_FAIL("Bad buffer");
  return bytes ? copy_out(e, host, dev, bytes) : 0;     // (copy_out / copy_in announce the host range)
}
int vpic_hip_copy_from_host(vpic_hip_engine_t *e, void *dev, const void *host, size_t bytes) {
  ENGINE(e); if (bytes && (!host || !dev)) VH_FAIL("Bad buffer");
  return bytes ? copy_in(e, dev, host, bytes) : 0;
}
int vpic_hip_boundary_p_counts(vpic_hip_engine_t *e, int32_t ns[6]) {
  ENGINE(e);
  for (int f = 0; f < 6; f++) ns[f] = e->send_count[f];
  return 0;
}
void *vpic_hip_boundary_p_send_buffer(vpic_hip_engine_t *e, int face) {
  return (e && face >= 0 && face < 6) ? (void *)e->send_buf[face] : nullptr;
}
int vpic_hip_boundary_p_get_injectors(vpic_hip_engine_t *e, int face, void *dev_dst) {
  ENGINE(e);
  if (face < 0 || face > 5 || !dev_dst) VH_FAIL("bad face/buffer");
  const size_t n = (size_t)e->send_count[face];
  if (n) VH_CHECK(hipMemcpyAsync(dev_dst, e->send_buf[face], n * sizeof(vpic_particle_injector_t), hipMemcpyDeviceToDevice, e->stream));
  return 0;
}
int vpic_hip_boundary_p_inject(vpic_hip_engine_t *e, const void *dev_injectors, int n) {
  ENGINE(e);
  return k_boundary_p_inject(e, (const vpic_particle_injector_t *)dev_injectors, n);
}

int vpic_hip_face_count(const vpic_hip_engine_t *e, int dir) { return (e && dir >= 0 && dir < 6) ? k_face_count(e, dir) : 0; }
int vpic_hip_pack_tang_b(vpic_hip_engine_t *e, int dir, void *buf) { ENGINE(e); return k_pack_face(e, dir, (float *)buf, 0); }
int vpic_hip_unpack_tang_b(vpic_hip_engine_t *e, int dir, const void *buf) { ENGINE(e); return k_unpack_face(e, dir, (const float *)buf, 0); }
int vpic_hip_pack_jf(vpic_hip_engine_t *e, int dir, void *buf) { ENGINE(e); return k_pack_face(e, dir, (float *)buf, 1); }
int vpic_hip_unpack_jf(vpic_hip_engine_t *e, int dir, const void *buf) { ENGINE(e); return k_unpack_face(e, dir, (const float *)buf, 1); }

// src/vpic/advance.cxx:38-214 for a domain that needs no other domain
// The adaptive decision for one species (see vpic_hip_step): reads the event pairs k_advance_p and
// k_sort_p record while e->time_kernels is set.
static int sort_due(Engine *e, Species &s, int max_interval, int *due) {
  e->time_kernels = true;
  s.adaptive = true;
  float ms = 0;
  *due = 0;
  if (s.sort_timed && hipEventSynchronize(s.ev[3]) == hipSuccess && hipEventElapsedTime(&ms, s.ev[2], s.ev[3]) == hipSuccess) {
    s.t_sort = ms; s.sort_timed = false;
    const int fl = s.coarse_sorted ? 1 : 0;                // (a sort right after a change of flavour is booked to the new one: one stray sample)
    if (s.sorted_after >= 1 && s.sorted_after <= 32) { s.s_hist[fl][s.sorted_after] = ms; s.c_hist[fl][s.sorted_after] = (ms + s.prev_sum) / s.sorted_after; }
    // cost per step of the cycle this sort closed, booked to the flavour it ran in (cycles right after a change of
    // flavour still carry the other one's disorder and are not counted)
    if (s.sorted_after >= 1 && s.tile_valid && s.flavour_cycles >= 2) {
      const int f = s.coarse_sorted ? 1 : 0;
      const double c = (ms + s.prev_sum) / s.sorted_after;
      s.flavour_cost[f] = s.flavour_cost[f] > 0 ? 0.5 * (s.flavour_cost[f] + c) : c;
    }
  }
  if (s.push_timed && hipEventSynchronize(s.ev[1]) == hipSuccess && hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) {
    // predicted cost of the NEXT push: the last one plus the growth to expect.  Push times grow faster than linearly
    // once particles outrun the LDS window (a ballistic plasma leaves a tile's halo after a few steps and every deposit
    // outside costs twelve global atomics), so the growth is the one an EARLIER cycle saw at this position when one got
    // that far; otherwise the growth seen last (within this cycle, or -- after one push -- the first growth of the
    // latest cycle that had two).  Every 64th cycle forgets the recorded growths, so that one that has died down gets
    // measured again.
    const int at = s.n_push;                            // position of the push just timed within its cycle
    const int fl = s.coarse_sorted ? 1 : 0;
    double *t_hist = s.t_hist[fl], *s_hist = s.s_hist[fl], *c_hist = s.c_hist[fl]; int &n_hist = s.n_hist[fl];
    double growth = 0;
    if (at >= 1) { growth = ms - s.t_last; if (at == 1) s.growth_first = growth; }
    else if ((s.n_cycle & 63) != 63) growth = s.growth_first;
    if ((s.n_cycle & 63) == 63) n_hist = 0;
    if (at + 1 < n_hist && at + 1 < 32) growth = std::max(growth, t_hist[at + 1] - t_hist[at]);
    if (at < 32) { t_hist[at] = ms; if (n_hist < at + 1) n_hist = at + 1; }
    s.t_last = ms;
    s.t_sum += ms; s.n_push++; s.push_timed = false;
    // Sort now, after n pushes, or after one more?  Whichever has the lower cost per step, the sort included.  The sort
    // is dearer the longer it is put off (the disorder it undoes grows: 2.4 ms after one step of a vth = 0.6 c species,
    // 4.5 after three), so its cost is the one seen at that cycle length when there is one on record.
    const int n = s.n_push;
    const double t_next = (double)ms + (growth > 0 ? growth : 0);
    const double sort_now = (n <= 32 && s_hist[n] > 0) ? s_hist[n] : s.t_sort;
    double sort_later = (n + 1 <= 32 && s_hist[n + 1] > 0) ? s_hist[n + 1] : sort_now;
    if (sort_later < sort_now) sort_later = sort_now;
    if ((s.n_cycle & 63) == 63) sort_later = sort_now;
    *due = (sort_later + s.t_sum + t_next) * n >= (sort_now + s.t_sum) * (n + 1);
    // What whole cycles of n and of n + 1 pushes actually cost per step, when both are on record, overrules the
    // prediction; and every eighth cycle is ended one push earlier than the last one when no cycle of that length is on record yet (the
    // prediction cannot know what a sort costs after fewer steps than it has ever been put off).
    if ((s.n_cycle & 63) == 63) for (int k = 0; k < 34; k++) c_hist[k] = 0;
    if (n <= 32 && c_hist[n] > 0 && c_hist[n + 1] > 0) *due = c_hist[n] <= c_hist[n + 1];
    else if (!*due && n <= 32 && c_hist[n] == 0 && (s.n_cycle & 7) == 7 && n == s.sorted_after - 1) *due = 1;   // one push earlier than last time
    if (e->knobs.policy_debug) fprintf(stderr, "sort policy: n=%d T=%.3f T_next=%.3f S_now=%.3f S_later=%.3f sum=%.3f c[n]=%.3f c[n+1]=%.3f flavour %d (%.3f / %.3f per step) -> %s\n", n, (double)ms, t_next, sort_now, sort_later, s.t_sum, n <= 32 ? c_hist[n] : 0.0, n <= 32 ? c_hist[n + 1] : 0.0, (int)s.coarse_sorted, s.flavour_cost[0], s.flavour_cost[1], *due ? "sort" : "go on");

  }
  if (!s.sorted_once || (max_interval > 0 && s.n_push >= max_interval)) *due = 1;
  return 0;
}
int vpic_hip_sort_due(vpic_hip_engine_t *e, int sp, int max_interval, int *due) {
  ENGINE(e); SPECIES(e, sp); if (!due) VH_FAIL("Bad output");
  return sort_due(e, e->species[sp], max_interval, due);
}

// The next vpic_hip_advance_p of this species also counts its particles' final cells for the sort that follows it (hosts that
// drive the steps themselves and know that the next step sorts; vpic_hip_step does it by itself).
int vpic_hip_species_sort_hint(vpic_hip_engine_t *e, int sp) {
  ENGINE(e); SPECIES(e, sp);
  e->species[sp].hist_request = true;
  return 0;
}
int vpic_hip_species_stats(vpic_hip_engine_t *e, int sp, int64_t out[8]) {
  ENGINE(e); SPECIES(e, sp); if (!out) VH_FAIL("Bad output");
  const Species &s = e->species[sp];
  out[0] = s.crossed_host ? s.crossed_host[0] : 0; out[1] = s.crossed_host ? s.crossed_host[1] : 0; out[2] = s.crossed_host ? s.crossed_host[3] : 0;
  out[3] = s.n_cycle; out[4] = s.early_sorts; out[5] = s.tile_unbalanced; out[6] = s.coarse_sorted; out[7] = s.n_holes;
  return 0;
}
int vpic_hip_species_sort_order(vpic_hip_engine_t *e, int sp, int *order) {
  ENGINE(e); SPECIES(e, sp); if (!order) VH_FAIL("Bad output");
  const Species &s = e->species[sp];
  *order = s.tile_valid ? 2 : s.partition_valid ? 1 : 0;
  return 0;
}

int vpic_hip_step(vpic_hip_engine_t *e, int64_t step, int sort_interval) {
  ENGINE(e);
  for (int f = 0; f < 6; f++) {
    const int fb = e->gk.fbc[f], pb = e->gk.pbc[f];
    if ((fb >= 0 && fb != e->gk.rank) || (pb >= 0 && pb != e->gk.rank))
      VH_FAIL("vpic_hip_step drives single-domain steps; face %d is shared with another domain", f);
  }
  // The host may enqueue steps faster than the device runs them; what it reads of the device's state without waiting (the
  // counts behind the sort decisions below: pinned words the device publishes behind every launch) is then many steps old --
  // a bench loop of 80 steps is enqueued before the first has finished, and the early sort never fired.  So the host stays at
  // most TWO steps ahead: it waits for the end of step n - 2 before it enqueues step n (the queue never runs dry).
  if (!e->step_done[0]) for (auto &ev : e->step_done) VH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (e->steps_enqueued >= 2) VH_CHECK(hipEventSynchronize(e->step_done[(e->steps_enqueued - 2) & 3]));
  if (vpic_hip_clear_accumulators(e)) return 1;                                   // advance.cxx:38
  // advance.cxx:43-51.  sort_interval > 0: every sort_interval steps, as a deck says; < 0: ADAPTIVE, at
  // the latest every -sort_interval steps.  Sorting changes no physics, only the array order (and with
  // it the rounding order of accumulator sums); what it buys is a cheaper advance_p.  The engine times
  // every sort and every advance_p of a species with HIP events and sorts the species again as soon as
  // its NEXT push (the last one plus the growth last seen) would cost at least the average cost per
  // step of the current cycle, the sort included:
  //     T_next >= (S + T_1 + ... + T_n) / n           (the optimality condition of a periodic policy
  // for push times that grow between sorts).  Reconnection-hot species (0.3 cells per step) end up
  // sorted every step or two, warm ones every 3-5, cold beams every 7-10.
  std::vector<int> due_list;
  for (size_t k = 0; k < e->species.size(); k++) {
    Species &s = e->species[k];
    int due = sort_interval > 0 && step % sort_interval == 0;
    if (sort_interval < 0 && sort_due(e, s, -sort_interval, &due)) return 1;
    // A fixed interval that outlasts the windows: once the species' particles have left what their tiles' windows can follow
    // (three cells: ~27 steps of the two-stream beams), every deposit is twelve global atomics and a launch costs four times
    // what it should (round 3: sort_interval = 40 ran at 14 G pushes/s).  The runs that missed the windows in the last launch
    // (a pinned word the device publishes behind every launch: stale by a launch or two, which is early enough) cost ~0.17 ns
    // and more each (they pile up on the same few accumulators: round 3 measured +50 ms per launch), a sort ~18 ps per particle:
    // when one launch missed more than 32 runs per tile -- half of where the windows stop following (publish_counter_kernel) --
    // and at least two steps are left, sort now.
    if (!due && sort_interval > 0 && e->knobs.early_sort && s.tile_valid && !s.chargeless && s.crossed_host) {
      const int64_t left = sort_interval - step % sort_interval;
      // (word 4: the sort cycle the count was taken in -- the host runs ahead of the device, and a count from before the last sort must not trigger another)
      // ... and the steps left must pay for it: a missed run costs ~0.34 ns (39.3 against 34.1 ms per step at 7.4e6 of them per
      // launch, profiles/r04_sort_interval_40_step_by_step.txt), an unscheduled sort ~19 ps per particle (it cannot happen inside
      // the push), and the misses grow: sort when misses x steps left exceed a 27th of the particles.  (The heated two-stream
      // deck at interval 10 reaches 32 runs per tile three steps before its scheduled sort: not worth one of its own.)
      const int64_t missed = s.crossed_host[3];
      if (s.crossed_host[4] == (unsigned)s.n_cycle && left >= 2 && missed > 32ll * make_tile_k(e->gk).ntiles && missed * left * 27 > s.np) { due = 1; s.early_sorts++; }
    }
    if (due) due_list.push_back((int)k);
  }
  std::vector<char> pushed(e->species.size(), 0), sort_first(e->species.size(), 0);
  auto push = [&](size_t k) -> int {                                              // advance.cxx:70-73
    Species &s = e->species[k];
    // a species that is due is sorted right before its own push (not all sorts first): a sort that finds the counts of the
    // push before it hands the work to this push (Species::fuse_pending: the particles are written straight to their sorted
    // places; what vpic_hip_sort_advance_p does).  Measured at 256^3 x 64 ppc (profiles/r03_sort_inside_push_ab.txt and the
    // r03_bench_sort_then_push.json of three profile runs): a 30 ms launch instead of 16.4 + 20, the step gains 1.6-3.8 %, the
    // plain launches measure 0-1.3 % slower.  VPIC_HIP_SORT_IN_PUSH=0 turns it off.
    // (... or before it, where that is cheaper: sort_and_push)
    pushed[k] = 1;
    auto hints = [&](Species &sp) -> int {
      // the next step sorts this species: let this push count for that sort (push.hip, Species::hist)
      if (sort_interval > 0 && (step + 1) % sort_interval == 0 && wants_tile_order(e, sp)) sp.hist_request = true;
      return order_for_deterministic_push(e, sp);
    };
    if (sort_first[k]) { sort_first[k] = 0; return sort_and_push(e, s, sort_interval > 0, hints); }
    if (hints(s)) return 1;
    return push_timing_hinted(e, s);
  };
  for (int k : due_list) sort_first[(size_t)k] = 1;
  for (size_t k = 0; k < e->species.size(); k++) if (!pushed[k] && push(k)) return 1;
  // advance.cxx:74 reduce_accumulators: single accumulator, nothing to do
  for (int round = 0; round < 3; round++) {                                       // advance.cxx:94-96: num_comm_round rounds;
    if (k_boundary_p_pack(e)) return 1;                                           // here absorbing / refluxing faces only
    bool pending = false;
    for (auto &s : e->species) pending = pending || s.nm > 0;
    if (!pending) break;
  }
  if (k_clear_jf_unload_accumulator(e)) return 1;                                 // advance.cxx:109-110, one pass
  if (k_synchronize_jf_local(e)) return 1;                                        // advance.cxx:112
  if (k_advance_b(e, 0.5f)) return 1;                                             // advance.cxx:129
  if (k_advance_e(e)) return 1;                                                   // advance.cxx:133
  if (k_advance_b(e, 0.5f)) return 1;                                             // advance.cxx:147
  if (k_load_interpolator(e)) return 1;                                           // advance.cxx:214
  VH_CHECK(hipEventRecord(e->step_done[e->steps_enqueued & 3], e->stream));
  e->steps_enqueued++;
  return 0;
}

int vpic_hip_measure_disorder(vpic_hip_engine_t *e, int sp, double *fraction) {
  ENGINE(e); SPECIES(e, sp); if (!fraction || sp >= MAX_SPECIES) VH_FAIL("Bad argument");
  if (k_measure_disorder(e, e->species[sp], sp)) return 1;
  VH_CHECK(hipStreamSynchronize(e->stream));
  *fraction = e->species[sp].np ? 8.0 * e->host_miss[sp] / (double)e->species[sp].np : 0.0;
  return 0;
}
int vpic_hip_profile_enable(vpic_hip_engine_t *e, int on) {
  ENGINE(e);
  if (collect_profile(e)) return 1;
  e->profile = on != 0;
  e->prof_ms = 0; e->prof_launches = 0; e->prof_particles = 0;
  e->prof_sort_ms = 0; e->prof_sort_launches = 0; e->prof_sort_particles = 0;
  for (int k = 0; k < MAX_SPECIES; k++) { e->prof_sp_ms[k] = 0; e->prof_sp_launches[k] = 0; e->prof_sp_particles[k] = 0; }
  return 0;
}
int vpic_hip_profile_read_species(vpic_hip_engine_t *e, int sp, double *ms, int64_t *launches, int64_t *particles) {
  ENGINE(e); SPECIES(e, sp); if (sp >= MAX_SPECIES) VH_FAIL("bad species id %d", sp);
  VH_CHECK(hipStreamSynchronize(e->stream));
  if (collect_profile(e)) return 1;
  if (ms) *ms = e->prof_sp_ms[sp];
  if (launches) *launches = e->prof_sp_launches[sp];
  if (particles) *particles = e->prof_sp_particles[sp];
  return 0;
}
int vpic_hip_profile_read(vpic_hip_engine_t *e, double *ms, int64_t *launches, int64_t *particles) {
  ENGINE(e);
  VH_CHECK(hipStreamSynchronize(e->stream));
  if (collect_profile(e)) return 1;
  if (ms) *ms = e->prof_ms;
  if (launches) *launches = e->prof_launches;
  if (particles) *particles = e->prof_particles;
  return 0;
}
int vpic_hip_profile_read_sorting(vpic_hip_engine_t *e, double *ms, int64_t *launches, int64_t *particles) {
  ENGINE(e);
  VH_CHECK(hipStreamSynchronize(e->stream));
  if (collect_profile(e)) return 1;
  if (ms) *ms = e->prof_sort_ms;
  if (launches) *launches = e->prof_sort_launches;
  if (particles) *particles = e->prof_sort_particles;
  return 0;
}

}  // extern "C"

// dropin.hip -- the reference's L3 C signatures over host arrays (include/vpic_hip_dropin.h),
// implemented as upload -> resident-engine kernel -> download.  One engine is cached per grid.
#include "engine.h"
#include "vpic_hip_dropin.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

using namespace vpichip;

namespace {

// src/util/util_base.h:213-219
#define DIE(...) do { fprintf(stderr, "Error at %s(%i):\n\t", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define CK(call) do { if (call) DIE("%s", vpic_hip_last_error()); } while (0)

int g_acc_copies = 1;
int g_n_mat = 1;

struct Key {
  int nx, ny, nz; float dt, cvac, eps0, damp, dx, dy, dz; int fbc[6], pbc[6]; int rank;
  bool operator<(const Key &o) const { return memcmp(this, &o, sizeof(Key)) < 0; }
};
struct Cached {
  vpic_hip_engine_t *e; int sp; int64_t sp_cap; void *inj_dev;          // inj_dev: one injector, for move_p
  int64_t sp_mcap = 0;                                                  // mover slots the scratch species was ALLOCATED with
  // grids that share faces with other ranks (vpic_hip_ref_set_transport): four message buffers (device + host staging) and
  // one engine species per species id of the caller's list (boundary_p moves them all at once)
  void *xdev[4] = {nullptr, nullptr, nullptr, nullptr}; size_t xbytes[4] = {0, 0, 0, 0}; std::vector<char> xhost[4];
  std::vector<int> bsp; std::vector<int64_t> bsp_cap, bsp_mcap;
};
std::map<Key, Cached> g_engines;

// BOUNDARY(i,j,k) = INDEX_FORTRAN_3(i,j,k,-1,1,-1,1,-1,1) (src/grid/grid.h:54)
int boundary_index(int face) {
  static const int d[6][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  return (d[face][0] + 1) + 3 * ((d[face][1] + 1) + 3 * (d[face][2] + 1));
}

// Per-face codes from the reference's tables.  Particle side: the neighbor[] entries of every
// boundary voxel of a face must agree (what size_grid + join_grid + set_pbc produce, ops.c:74-231).
vpic_hip_grid_t describe(const vpic_grid_t *g) {
  if (!g) DIE("Bad grid");
  if (!g->neighbor) DIE("grid has no neighbor table");
  vpic_hip_grid_t d;
  d.dt = g->dt; d.cvac = g->cvac; d.eps0 = g->eps0; d.damp = g->damp;
  d.dx = g->dx; d.dy = g->dy; d.dz = g->dz; d.rdx = g->rdx; d.rdy = g->rdy; d.rdz = g->rdz;
  d.nx = g->nx; d.ny = g->ny; d.nz = g->nz;
  d.rank = g->bc[13];
  const int n[3] = {g->nx, g->ny, g->nz};
  const int64_t sy = g->nx + 2, sz = sy * (g->ny + 2);
  for (int face = 0; face < 6; face++) {
    const int bc = g->bc[boundary_index(face)];
    d.fbc[face] = bc;                                     // >= 0: rank sharing the face; < 0: local code
    const int axis = face % 3, plane = face < 3 ? 1 : n[axis];
    int64_t first = 0;
    bool have = false;
    int lo[3] = {1, 1, 1}, hi[3] = {n[0], n[1], n[2]};
    lo[axis] = hi[axis] = plane;
    int code = 0;
    for (int z = lo[2]; z <= hi[2]; z++) for (int y = lo[1]; y <= hi[1]; y++) for (int x = lo[0]; x <= hi[0]; x++) {
      const int64_t v = x + sy * y + sz * z;
      const int64_t nb = g->neighbor[6 * v + face];
      int c;
      if (nb < 0) c = (int)nb;                            // reflect / absorb / custom handler code
      else if (nb >= g->rangel && nb <= g->rangeh) c = d.rank;
      else c = (bc >= 0 && bc != d.rank) ? bc : -1000;    // another domain's voxel: that domain is bc
      if (!have) { first = nb; code = c; have = true; }
      else if (c != code) DIE("face %d mixes particle boundary conditions: not a box deck", face);
    }
    (void)first;
    if (code == -1000) DIE("face %d leads to another domain but bc[] names none", face);
    if (code < VPIC_ABSORB_PARTICLES) DIE("custom particle boundary handlers are not supported on the HIP path");
    d.pbc[face] = code;
  }
  return d;
}

Cached &engine_for(const vpic_grid_t *g) {
  const vpic_hip_grid_t d = describe(g);
  Key k;
  memset(&k, 0, sizeof(k));
  k.nx = d.nx; k.ny = d.ny; k.nz = d.nz; k.dt = d.dt; k.cvac = d.cvac; k.eps0 = d.eps0; k.damp = d.damp;
  k.dx = d.dx; k.dy = d.dy; k.dz = d.dz; k.rank = d.rank;
  for (int f = 0; f < 6; f++) { k.fbc[f] = d.fbc[f]; k.pbc[f] = d.pbc[f]; }
  auto it = g_engines.find(k);
  if (it != g_engines.end()) return it->second;
  Cached c;
  c.e = nullptr; c.sp = -1; c.sp_cap = 0; c.inj_dev = nullptr;
  CK(vpic_hip_create(&c.e, &d, -1));
  return g_engines.emplace(k, c).first->second;
}

int nv_of(const vpic_grid_t *g) { return (g->nx + 2) * (g->ny + 2) * (g->nz + 2); }

// a species slot large enough for np particles / max_nm movers (recreated when it must grow)
int species_for(Cached &c, float q_m, int64_t np, int64_t max_nm) {
  const int64_t need = np > 0 ? np : 1;
  Engine *e = c.e;
  // (the species' max_nm below is the CALLER's limit for this call -- advance_p.cxx:463-465 drops movers beyond it -- and
  // changes from call to call; what decides whether the slot must be recreated is what it was allocated with)
  if (c.sp < 0 || c.sp_cap < need || c.sp_mcap < max_nm) {
    const int64_t cap = need + (need >> 2) + 1024, mcap = max_nm > 1024 ? max_nm : 1024;
    c.sp = vpic_hip_species_create(c.e, q_m, cap, mcap);
    if (c.sp < 0) DIE("%s", vpic_hip_last_error());
    c.sp_cap = cap; c.sp_mcap = mcap;
  }
  e->species[c.sp].q_m = q_m;
  e->species[c.sp].max_nm = max_nm > 0 ? (c.sp_mcap < max_nm ? c.sp_mcap : max_nm) : 1;
  return c.sp;
}


// ---- grids of SEVERAL ranks ----------------------------------------------------------------------------------------------
// The reference exchanges ghost planes, boundary sums and particles through its port layer (src/grid/grid_comm.c:7-78 over
// src/util/mp).  The twins do not link against it: the host registers a transport -- two callbacks, see
// include/vpic_hip_dropin.h -- and the twins run the engine's face messages (pack on the device, stage through host memory,
// exchange, unpack) in the order remote.c / boundary_p.c do.  Without a transport a grid that shares a face is refused.
static vpic_hip_ref_transport_t g_tr = {nullptr, nullptr, nullptr};

static bool shared_face(const Cached &c, int f, bool particles = false) {
  const int b = particles ? c.e->gk.pbc[f] : c.e->gk.fbc[f];
  return b >= 0 && b != c.e->gk.rank;
}
static bool shared_axis(const Cached &c, int a, bool particles = false) { return shared_face(c, a, particles) || shared_face(c, a + 3, particles); }
static bool multi(const Cached &c, bool particles = false) { return shared_axis(c, 0, particles) || shared_axis(c, 1, particles) || shared_axis(c, 2, particles); }
static void need_transport(const Cached &c, const char *who, bool particles = false) {
  if (multi(c, particles) && !(g_tr.exchange && g_tr.allsum_d))
    DIE("%s: the grid shares a face with another rank and no transport with both callbacks (exchange, allsum_d) is registered (vpic_hip_ref_set_transport)", who);
}
static void allsum(const vpic_grid_t *g, double *v, int n) { if (g_tr.allsum_d) g_tr.allsum_d(g_tr.ctx, g, v, n); }
static void *xbuf(Cached &c, int k, size_t bytes) {
  if (bytes > c.xbytes[k]) {
    vpic_hip_device_free(c.e, c.xdev[k]);
    c.xbytes[k] = bytes + bytes / 4 + 4096;
    c.xdev[k] = vpic_hip_device_alloc(c.e, c.xbytes[k]);
    if (!c.xdev[k]) DIE("%s", vpic_hip_last_error());
    c.xhost[k].resize(c.xbytes[k]);
  }
  return c.xdev[k];
}
// one message each way along `axis`: send s_lo towards -axis, s_hi towards +axis; what travelled towards -axis (from the rank
// behind the high face) arrives in r_lo, what travelled towards +axis in r_hi
static void axis_exchange(Cached &c, const vpic_grid_t *g, int axis, bool particles, const void *s_lo, size_t ns_lo, const void *s_hi, size_t ns_hi,
                          void *r_lo, size_t nr_lo, void *r_hi, size_t nr_hi) {
  const bool lo = shared_face(c, axis, particles), hi = shared_face(c, axis + 3, particles);
  const void *send[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void *recv[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t n_send[6] = {0, 0, 0, 0, 0, 0}, n_recv[6] = {0, 0, 0, 0, 0, 0};
  if (lo) { send[axis] = s_lo; n_send[axis] = ns_lo; recv[axis + 3] = r_hi; n_recv[axis + 3] = nr_hi; }
  if (hi) { send[axis + 3] = s_hi; n_send[axis + 3] = ns_hi; recv[axis] = r_lo; n_recv[axis] = nr_lo; }
  g_tr.exchange(g_tr.ctx, g, send, n_send, recv, n_recv);
}
template <class Pack, class Unpack>
static void plane_exchange(Cached &c, const vpic_grid_t *g, int axis, size_t bytes, Pack pack, Unpack unpack) {
  const int lo = axis, hi = axis + 3;
  void *s0 = xbuf(c, 0, bytes), *s3 = xbuf(c, 1, bytes), *r0 = xbuf(c, 2, bytes), *r3 = xbuf(c, 3, bytes);
  if (shared_face(c, lo)) { pack(lo, s0); CK(vpic_hip_copy_to_host(c.e, &c.xhost[0][0], s0, bytes)); }
  if (shared_face(c, hi)) { pack(hi, s3); CK(vpic_hip_copy_to_host(c.e, &c.xhost[1][0], s3, bytes)); }
  axis_exchange(c, g, axis, false, &c.xhost[0][0], bytes, &c.xhost[1][0], bytes, &c.xhost[2][0], bytes, &c.xhost[3][0], bytes);
  if (shared_face(c, hi)) { CK(vpic_hip_copy_from_host(c.e, r0, &c.xhost[2][0], bytes)); unpack(lo, r0); }   // travelled towards -axis: came from the high side
  if (shared_face(c, lo)) { CK(vpic_hip_copy_from_host(c.e, r3, &c.xhost[3][0], bytes)); unpack(hi, r3); }
}
static void x_tang_b(Cached &c, const vpic_grid_t *g) {                     // remote.c:61-134
  vpic_hip_engine_t *e = c.e;
  for (int a = 0; a < 3; a++)
    if (shared_axis(c, a)) plane_exchange(c, g, a, sizeof(float) * (size_t)vpic_hip_face_count(e, a),
                                          [e](int d, void *b) { CK(vpic_hip_pack_tang_b(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_tang_b(e, d, b)); });
}
static double x_message(Cached &c, const vpic_grid_t *g, int kind, int axis) {   // normal E / div_b_err ghosts, tang E + norm B averages
  double err = 0, *perr = &err;
  vpic_hip_engine_t *e = c.e;
  plane_exchange(c, g, axis, sizeof(float) * (size_t)vpic_hip_face_message_count(e, kind, axis),
                 [e, kind](int d, void *b) { CK(vpic_hip_pack_face_message(e, kind, d, b)); },
                 [e, kind, perr](int d, void *b) { double x = 0; CK(vpic_hip_unpack_face_message(e, kind, d, b, &x)); *perr += x; });
  return err;
}
static double x_message(Cached &c, const vpic_grid_t *g, int kind) {
  double err = 0;
  for (int a = 0; a < 3; a++) if (shared_axis(c, a)) err += x_message(c, g, kind, a);
  return err;
}


}  // namespace

extern "C" {

void vpic_hip_ref_set_transport(const vpic_hip_ref_transport_t *t) {
  if (t) g_tr = *t; else { g_tr.exchange = nullptr; g_tr.allsum_d = nullptr; g_tr.ctx = nullptr; }
}
void vpic_hip_ref_set_accumulator_copies(int n) { g_acc_copies = n < 1 ? 1 : n; }
void vpic_hip_ref_set_material_count(int n) { g_n_mat = n < 1 ? 1 : n; }

// ---- allocation slots ---------------------------------------------------------------------------------
// Blocks the reference may free itself (sp->partition in delete_species, or a table mixed from both
// libraries) must look like its MALLOC_ALIGNED blocks (src/util/util.c:46-91): the address malloc
// returned is kept in the word just below the aligned address, and FREE_ALIGNED frees THAT.
static void *ref_aligned(size_t bytes, size_t align) {
  const size_t a = (align < 16 ? 16 : align) - 1;
  char *raw = (char *)malloc(bytes + a + sizeof(char *));
  if (!raw) DIE("Failed to allocate.");
  char *aligned = (char *)(((uintptr_t)(raw + a + sizeof(char *))) & ~(uintptr_t)a);
  ((char **)aligned)[-1] = raw;
  return aligned;
}
static void ref_aligned_free(void *p) { if (p) free(((char **)p)[-1]); }
static void *zeroed(size_t bytes) {
  void *p = ref_aligned(bytes ? bytes : 128, 128);
  memset(p, 0, bytes);
  return p;
}
static size_t voxels(const vpic_grid_t *g) {
  if (!g || g->nx < 1 || g->ny < 1 || g->nz < 1) DIE("Bad grid.");
  return (size_t)(g->nx + 2) * (g->ny + 2) * (g->nz + 2);
}
vpic_field_t *vpic_hip_ref_new_field(vpic_grid_t *g) { return (vpic_field_t *)zeroed(voxels(g) * sizeof(vpic_field_t)); }
void vpic_hip_ref_delete_field(vpic_field_t *f) { ref_aligned_free(f); }
vpic_hydro_t *vpic_hip_ref_new_hydro(vpic_grid_t *g) { return (vpic_hydro_t *)zeroed(voxels(g) * sizeof(vpic_hydro_t)); }
void vpic_hip_ref_delete_hydro(vpic_hydro_t *h) { ref_aligned_free(h); }
vpic_interpolator_t *vpic_hip_ref_new_interpolator(vpic_grid_t *g) { return (vpic_interpolator_t *)zeroed(voxels(g) * sizeof(vpic_interpolator_t)); }
void vpic_hip_ref_delete_interpolator(vpic_interpolator_t *fi) { ref_aligned_free(fi); }
vpic_accumulator_t *vpic_hip_ref_new_accumulators(vpic_grid_t *g) {            // sf_interface.c:56-75
  const size_t stride = (voxels(g) + 1) & ~(size_t)1;
  return (vpic_accumulator_t *)zeroed((size_t)g_acc_copies * stride * sizeof(vpic_accumulator_t));
}
void vpic_hip_ref_delete_accumulators(vpic_accumulator_t *a) { ref_aligned_free(a); }
// sfa.c:80-177: one coefficient record per material, indexed by id; double exp / sinh on float operands
vpic_material_coefficient_t *vpic_hip_ref_new_material_coefficients(vpic_grid_t *g, vpic_material_t *m_list) {
  if (!g) DIE("Invalid grid.");
  if (!m_list) DIE("Empty material list.");
  int n = 0;
  for (const vpic_material_t *m = m_list; m; m = m->next) n = m->id + 1 > n ? m->id + 1 : n;
  vpic_material_coefficient_t *table = (vpic_material_coefficient_t *)zeroed((size_t)n * sizeof(vpic_material_coefficient_t));
  for (const vpic_material_t *m = m_list; m; m = m->next) {
    vpic_material_coefficient_t *mc = table + m->id;
    const float eps[3] = {m->epsx, m->epsy, m->epsz}, sigma[3] = {m->sigmax, m->sigmay, m->sigmaz};
    float a[3], decay[3], drive[3];
    for (int k = 0; k < 3; k++) {
      a[k] = (sigma[k] * g->dt) / (eps[k] * g->eps0);
      decay[k] = exp(-a[k]);
      if (a[k] == 0) drive[k] = 1. / eps[k];
      else if (decay[k] == 0) drive[k] = 0;
      else drive[k] = 2. * exp(-0.5 * a[k]) * sinh(0.5 * a[k]) / (a[k] * eps[k]);
    }
    mc->decayx = decay[0]; mc->decayy = decay[1]; mc->decayz = decay[2];
    mc->drivex = drive[0]; mc->drivey = drive[1]; mc->drivez = drive[2];
    mc->rmux = 1. / m->mux; mc->rmuy = 1. / m->muy; mc->rmuz = 1. / m->muz;
    mc->nonconductive = (a[0] == 0 && a[1] == 0 && a[2] == 0) ? 1. : 0.;
    mc->epsx = m->epsx; mc->epsy = m->epsy; mc->epsz = m->epsz;
  }
  g_n_mat = n;
  return table;
}
void vpic_hip_ref_delete_material_coefficients(vpic_material_coefficient_t *mc) { ref_aligned_free(mc); }

void vpic_hip_ref_load_interpolator(vpic_interpolator_t *fi, const vpic_field_t *f, const vpic_grid_t *g) {
  if (!fi) DIE("Bad interpolator");
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_set_interpolator(c.e, fi));        // ghost voxels keep the caller's values
  CK(vpic_hip_load_interpolator(c.e));
  CK(vpic_hip_get_interpolator(c.e, fi));
}

void vpic_hip_ref_clear_accumulators(vpic_accumulator_t *a, const vpic_grid_t *g) {
  if (!a) DIE("Invalid accumulator");
  if (!g) DIE("Invalid grid");
  const size_t stride = ((size_t)nv_of(g) + 1) & ~(size_t)1;          // POW2_CEIL(nv,2)
  memset(a, 0, sizeof(*a) * stride * (size_t)g_acc_copies);           // host array: plain memset
}

// sf_interface/reduce_accumulators.cxx:37-55 over the caller's replicated host array.  The HIP push
// only ever writes copy 0, so this matters only for copies user code filled itself.
void vpic_hip_ref_reduce_accumulators(vpic_accumulator_t *a, const vpic_grid_t *g) {
  if (!a) DIE("Bad accumulator");
  if (!g) DIE("Bad grid");
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const size_t stride = ((size_t)nv_of(g) + 1) & ~(size_t)1;
  for (int z = 1; z <= nz; z++) for (int y = 1; y <= ny; y++) for (int x = 1; x <= nx; x++) {
    float *da = (float *)&a[x + (nx + 2) * (y + (ny + 2) * z)];
    for (int n = 1; n < g_acc_copies; n++) {
      const float *sa = da + 12 * stride * n;
      for (int k = 0; k < 12; k++) da[k] += sa[k];
    }
  }
}

void vpic_hip_ref_unload_accumulator(vpic_field_t *f, const vpic_accumulator_t *a, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  if (!a) DIE("Bad accumulator");
  Cached &c = engine_for(g);
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_set_accumulator(c.e, a));
  CK(vpic_hip_unload_accumulator(c.e));
  CK(vpic_hip_get_fields(c.e, f));
}

int vpic_hip_ref_advance_p(vpic_particle_t *p0, int np, const float q_m, vpic_particle_mover_t *pm, int max_nm,
                           vpic_accumulator_t *a0, const vpic_interpolator_t *f0, const vpic_grid_t *g) {
  if (!p0) DIE("Bad particle array");
  if (np < 0) DIE("Bad number of particles");
  if (!pm) DIE("Bad particle mover");
  if (max_nm < 0) DIE("Bad number of movers");
  if (!a0) DIE("Bad accumulator");
  if (!f0) DIE("Bad interpolator");
  Cached &c = engine_for(g);
  const int sp = species_for(c, q_m, np, max_nm);
  CK(vpic_hip_set_interpolator(c.e, f0));
  CK(vpic_hip_set_accumulator(c.e, a0));          // advance_p ADDS to the caller's sums (copy 0)
  CK(vpic_hip_species_set_particles(c.e, sp, p0, np));
  CK(vpic_hip_advance_p(c.e, sp));
  CK(vpic_hip_species_get_particles(c.e, sp, p0, np));
  CK(vpic_hip_get_accumulator(c.e, a0));
  const int nm = (int)vpic_hip_species_nm(c.e, sp);
  CK(vpic_hip_species_get_movers(c.e, sp, pm, max_nm));
  return nm;
}

double vpic_hip_ref_energy_p(const vpic_particle_t *p0, int np, float q_m, const vpic_interpolator_t *f0,
                             const vpic_grid_t *g) {
  if (np < 0) DIE("Bad number of particles");
  if (!f0) DIE("Bad interpolator");
  Cached &c = engine_for(g);
  const int sp = species_for(c, q_m, np, 1);
  CK(vpic_hip_set_interpolator(c.e, f0));
  CK(vpic_hip_species_set_particles(c.e, sp, p0, np));
  double en = 0;
  CK(vpic_hip_energy_p(c.e, sp, &en));
  return en;
}

static void center_common(vpic_particle_t *p0, int np, float q_m, const vpic_interpolator_t *f0, const vpic_grid_t *g, bool un) {
  if (!p0) DIE("Bad particle array");
  if (np < 0) DIE("Bad number of particles");
  if (!f0) DIE("Bad interpolator");
  Cached &c = engine_for(g);
  const int sp = species_for(c, q_m, np, 1);
  CK(vpic_hip_set_interpolator(c.e, f0));
  CK(vpic_hip_species_set_particles(c.e, sp, p0, np));
  CK(un ? vpic_hip_uncenter_p(c.e, sp) : vpic_hip_center_p(c.e, sp));
  CK(vpic_hip_species_get_particles(c.e, sp, p0, np));
}
void vpic_hip_ref_center_p(vpic_particle_t *p0, int np, const float q_m, const vpic_interpolator_t *f0, const vpic_grid_t *g) { center_common(p0, np, q_m, f0, g, false); }
void vpic_hip_ref_uncenter_p(vpic_particle_t *p0, int np, const float q_m, const vpic_interpolator_t *f0, const vpic_grid_t *g) { center_common(p0, np, q_m, f0, g, true); }

void vpic_hip_ref_sort_p(vpic_species_t *sp, const vpic_grid_t *g) {
  if (!sp) DIE("Bad species");
  Cached &c = engine_for(g);
  const int nv = nv_of(g);
  if (!sp->partition) {                                               // sort_p.c:32
    sp->partition = (int32_t *)ref_aligned(sizeof(int32_t) * (size_t)(nv + 1), 128);   // delete_species frees it with FREE_ALIGNED
  }
  if (sp->np == 0) return;                                            // sort_p.c:35
  const int s = species_for(c, sp->q_m, sp->np, 1);
  CK(vpic_hip_species_set_particles(c.e, s, sp->p, sp->np));
  CK(vpic_hip_sort_p(c.e, s));
  CK(vpic_hip_species_get_particles(c.e, s, sp->p, sp->np));
  CK(vpic_hip_species_get_partition(c.e, s, sp->partition));
}


void vpic_hip_ref_advance_b(vpic_field_t *f, const vpic_grid_t *g, float frac) {
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_advance_b(c.e, frac));
  CK(vpic_hip_get_fields(c.e, f));
}

void vpic_hip_ref_advance_e(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  if (!m) DIE("Bad material coefficients");
  Cached &c = engine_for(g);
  need_transport(c, "advance_e");
  CK(vpic_hip_set_material_coefficients(c.e, m, g_n_mat));
  CK(vpic_hip_set_fields(c.e, f));
  if (multi(c)) x_tang_b(c, g);                           // advance_e.c:114,153: begin / end_remote_ghost_tang_b
  CK(vpic_hip_advance_e(c.e));
  CK(vpic_hip_get_fields(c.e, f));
}

void vpic_hip_ref_clear_jf(vpic_field_t *f, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  if (!g) DIE("Bad grid");
  const int nv = nv_of(g);
  for (int v = 0; v < nv; v++) f[v].jfx = f[v].jfy = f[v].jfz = 0;    // host array: sfa.c:188-211 as is
}

void vpic_hip_ref_synchronize_jf(vpic_field_t *f, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  need_transport(c, "synchronize_jf");
  CK(vpic_hip_set_fields(c.e, f));
  if (!multi(c)) CK(vpic_hip_synchronize_jf(c.e));
  else {
    vpic_hip_engine_t *e = c.e;
    CK(vpic_hip_local_adjust_jf(e));
    for (int a = 0; a < 3; a++) {                          // x, then y, then z: edges and corners propagate (remote.c:284-289)
      if (shared_axis(c, a)) plane_exchange(c, g, a, sizeof(float) * (size_t)vpic_hip_face_count(e, a),
                                            [e](int d, void *b) { CK(vpic_hip_pack_jf(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_jf(e, d, b)); });
      else CK(vpic_hip_synchronize_jf_self(e, a));
    }
  }
  CK(vpic_hip_get_fields(c.e, f));
}

void vpic_hip_ref_energy_f(double *energy6, const vpic_field_t *f, const vpic_material_coefficient_t *m,
                           const vpic_grid_t *g) {
  if (!energy6) DIE("Bad energy");
  if (!f) DIE("Bad field");
  if (!m) DIE("Bad material coefficients");
  Cached &c = engine_for(g);
  CK(vpic_hip_set_material_coefficients(c.e, m, g_n_mat));
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_energy_f(c.e, energy6));
  need_transport(c, "energy_f");                           // (a per-rank value must not pass for the global one)
  allsum(g, energy6, 6);                                   // energy_f.c:178
}

// ---- divergence cleaning family and charge densities: the remaining slots of
// field_advance_methods_t (field_advance.h:242-302) and accumulate_rho_p (spa.h:108-113) ----------
// (before: what a grid of several ranks exchanges first -- the ghost planes the kernel reads)
#define FIELD_TWIN(name, before, call, needs_m)                                                  \
  do {                                                                                           \
    if (!f) DIE("Bad field");                                                                    \
    Cached &c = engine_for(g);                                                                   \
    need_transport(c, name);                                                                     \
    if (needs_m) CK(vpic_hip_set_material_coefficients(c.e, m, g_n_mat));                        \
    CK(vpic_hip_set_fields(c.e, f));                                                             \
    if (multi(c)) { before; }                                                                    \
    CK(call);                                                                                    \
    CK(vpic_hip_get_fields(c.e, f));                                                             \
  } while (0)

void vpic_hip_ref_clear_rhof(vpic_field_t *f, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  if (!g) DIE("Bad grid");
  const int nv = nv_of(g);
  for (int v = 0; v < nv; v++) f[v].rhof = 0;                         // host array: sfa.c:213-234 as is
}
// boundary_p.c:9-71: one particle's charge spread over the 8 nodes of its cell into rhob (called inline by
// vpic.hxx:483-484 inject_particle_raw with update_rhob, and by boundary handlers)
void vpic_hip_ref_accumulate_rhob(vpic_field_t *f, const vpic_particle_t *p, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  if (!p) DIE("Bad particle");
  Cached &c = engine_for(g);
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_accumulate_rhob(c.e, p, 1, 1.f));
  CK(vpic_hip_get_fields(c.e, f));
}

void vpic_hip_ref_accumulate_rho_p(vpic_field_t *f, const vpic_particle_t *p0, int np, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  if (!p0) DIE("Bad particle array");
  if (np < 0) DIE("Bad number of particles");
  Cached &c = engine_for(g);
  const int sp = species_for(c, 1.f, np, 1);
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_species_set_particles(c.e, sp, p0, np));
  CK(vpic_hip_accumulate_rho_p(c.e, sp));
  CK(vpic_hip_get_fields(c.e, f));
}
void vpic_hip_ref_synchronize_rho(vpic_field_t *f, const vpic_grid_t *g) {
  const vpic_material_coefficient_t *m = nullptr;
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  need_transport(c, "synchronize_rho");
  (void)m;
  CK(vpic_hip_set_fields(c.e, f));
  if (!multi(c)) CK(vpic_hip_synchronize_rho(c.e));
  else {                                                   // remote.c:533-622
    vpic_hip_engine_t *e = c.e;
    CK(vpic_hip_local_adjust_rho(e));
    for (int a = 0; a < 3; a++) {
      if (shared_axis(c, a)) plane_exchange(c, g, a, sizeof(float) * (size_t)vpic_hip_rho_count(e, a),
                                            [e](int d, void *b) { CK(vpic_hip_pack_rho(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_rho(e, d, b)); });
      else CK(vpic_hip_synchronize_rho_self(e, a));
    }
  }
  CK(vpic_hip_get_fields(c.e, f));
}
void vpic_hip_ref_compute_rhob(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g) {
  if (!m) DIE("Bad material coefficients");
  FIELD_TWIN("compute_rhob", x_message(c, g, VPIC_HIP_MSG_NORM_E), vpic_hip_compute_rhob(c.e), true);
}
void vpic_hip_ref_compute_curl_b(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g) {
  if (!m) DIE("Bad material coefficients");
  FIELD_TWIN("compute_curl_b", x_tang_b(c, g), vpic_hip_compute_curl_b(c.e), true);
}
double vpic_hip_ref_synchronize_tang_e_norm_b(vpic_field_t *f, const vpic_grid_t *g) {
  const vpic_material_coefficient_t *m = nullptr;
  double err = 0;
  if (!f) DIE("Bad field");
  (void)m;
  Cached &c = engine_for(g);
  need_transport(c, "synchronize_tang_e_norm_b");
  CK(vpic_hip_set_fields(c.e, f));
  if (!multi(c)) CK(vpic_hip_synchronize_tang_e_norm_b(c.e, &err));
  else {                                                              // remote.c:298-414
    double x;
    CK(vpic_hip_local_adjust_tang_e_norm_b(c.e));
    for (int a = 0; a < 3; a++) {
      if (shared_axis(c, a)) err += x_message(c, g, VPIC_HIP_MSG_TANG_E_NORM_B, a);
      else { CK(vpic_hip_synchronize_tang_e_norm_b_self(c.e, a, &x)); err += x; }
    }
  }
  CK(vpic_hip_get_fields(c.e, f));
  allsum(g, &err, 1);                                                 // remote.c:410-412
  return err;
}
void vpic_hip_ref_compute_div_e_err(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g) {
  if (!m) DIE("Bad material coefficients");
  FIELD_TWIN("compute_div_e_err", x_message(c, g, VPIC_HIP_MSG_NORM_E), vpic_hip_compute_div_e_err(c.e), true);
}
double vpic_hip_ref_compute_rms_div_e_err(vpic_field_t *f, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  double l2[2];
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_rms_div_e_err_local(c.e, l2));
  need_transport(c, "compute_rms_div_e_err");
  allsum(g, l2, 2);                                                   // compute_rms_div_e_err.c:156-159
  return c.e->grid.eps0 * sqrt(l2[0] / l2[1]);
}
void vpic_hip_ref_clean_div_e(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g) {
  if (!m) DIE("Bad material coefficients");
  FIELD_TWIN("clean_div_e", (void)0, vpic_hip_clean_div_e(c.e), true);
}
void vpic_hip_ref_compute_div_b_err(vpic_field_t *f, const vpic_grid_t *g) {
  const vpic_material_coefficient_t *m = nullptr;
  FIELD_TWIN("compute_div_b_err", (void)0, vpic_hip_compute_div_b_err(c.e), false);
}
double vpic_hip_ref_compute_rms_div_b_err(vpic_field_t *f, const vpic_grid_t *g) {
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  double l2[2];
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_rms_div_b_err_local(c.e, l2));
  need_transport(c, "compute_rms_div_b_err");
  allsum(g, l2, 2);
  return c.e->grid.eps0 * sqrt(l2[0] / l2[1]);
}
void vpic_hip_ref_clean_div_b(vpic_field_t *f, const vpic_grid_t *g) {
  const vpic_material_coefficient_t *m = nullptr;
  FIELD_TWIN("clean_div_b", x_message(c, g, VPIC_HIP_MSG_DIV_B), vpic_hip_clean_div_b(c.e), false);
}

// ---- hydro (sf_interface.h:90-163, spa.h:115-123) --------------------------------------------------
void vpic_hip_ref_clear_hydro(vpic_hydro_t *h, const vpic_grid_t *g) {
  if (!h) DIE("Bad hydro");
  if (!g) DIE("Bad grid");
  memset(h, 0, sizeof(*h) * (size_t)nv_of(g));                        // host array: sf_interface.c:29-36 as is
}
void vpic_hip_ref_accumulate_hydro_p(vpic_hydro_t *h0, const vpic_particle_t *p0, int np, float q_m,
                                     const vpic_interpolator_t *f0, const vpic_grid_t *g) {
  if (!h0) DIE("Bad hydro");
  if (!p0) DIE("Bad particle array");
  if (np < 0) DIE("Bad number of particles");
  if (!f0) DIE("Bad field");
  Cached &c = engine_for(g);
  const int sp = species_for(c, q_m, np, 1);
  CK(vpic_hip_set_interpolator(c.e, f0));
  CK(vpic_hip_set_hydro(c.e, h0));
  CK(vpic_hip_species_set_particles(c.e, sp, p0, np));
  CK(vpic_hip_accumulate_hydro_p(c.e, sp));
  CK(vpic_hip_get_hydro(c.e, h0));
}
void vpic_hip_ref_synchronize_hydro(vpic_hydro_t *h, const vpic_grid_t *g) {
  if (!h) DIE("Bad hydro");
  Cached &c = engine_for(g);
  need_transport(c, "synchronize_hydro");
  CK(vpic_hip_set_hydro(c.e, h));
  if (!multi(c)) CK(vpic_hip_synchronize_hydro(c.e));
  else {                                                              // sf_interface/hydro.c:28-163
    vpic_hip_engine_t *e = c.e;
    CK(vpic_hip_local_adjust_hydro(e));
    for (int a = 0; a < 3; a++) {
      if (shared_axis(c, a)) plane_exchange(c, g, a, sizeof(float) * (size_t)vpic_hip_hydro_count(e, a),
                                            [e](int d, void *b) { CK(vpic_hip_pack_hydro(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_hydro(e, d, b)); });
      else CK(vpic_hip_synchronize_hydro_self(e, a));
    }
  }
  CK(vpic_hip_get_hydro(c.e, h));
}
void vpic_hip_ref_local_adjust_hydro(vpic_hydro_t *h, const vpic_grid_t *g) {
  if (!h) DIE("Bad hydro");
  Cached &c = engine_for(g);
  CK(vpic_hip_set_hydro(c.e, h));
  CK(vpic_hip_local_adjust_hydro(c.e));
  CK(vpic_hip_get_hydro(c.e, h));
}

// ---- move_p (spa.h:50-54 -> move_p.c:20-136): finish the move of particle pm->i of p0 --------------
// The one particle travels as an injector (its whole state plus the remaining displacement) through
// the kernel boundary_p uses for arrivals; it deposits into the caller's accumulator (copy 0).
int vpic_hip_ref_move_p(vpic_particle_t *p0, vpic_particle_mover_t *pm, vpic_accumulator_t *a0, const vpic_grid_t *g) {
  if (!p0 || !pm || !a0) DIE("Bad argument");
  Cached &c = engine_for(g);
  const int sp = species_for(c, 1.f, 1, 1);
  vpic_particle_t *p = p0 + pm->i;
  vpic_particle_injector_t inj;
  inj.dx = p->dx; inj.dy = p->dy; inj.dz = p->dz; inj.i = p->i;
  inj.ux = p->ux; inj.uy = p->uy; inj.uz = p->uz; inj.q = p->q;
  inj.dispx = pm->dispx; inj.dispy = pm->dispy; inj.dispz = pm->dispz; inj.sp_id = sp;
  CK(vpic_hip_set_accumulator(c.e, a0));
  CK(vpic_hip_species_set_particles(c.e, sp, p0, 0));                 // empty species: the particle lands in slot 0
  for (size_t k = 0; k < c.e->species.size(); k++) c.e->species[k].nm = 0;
  if (!c.inj_dev && hipMalloc(&c.inj_dev, sizeof(inj)) != hipSuccess) DIE("out of device memory");
  if (hipMemcpy(c.inj_dev, &inj, sizeof(inj), hipMemcpyHostToDevice) != hipSuccess) DIE("copy to the device failed");
  CK(vpic_hip_boundary_p_inject(c.e, c.inj_dev, 1));
  vpic_particle_t out;
  CK(vpic_hip_species_get_particles(c.e, sp, &out, 1));
  p->dx = out.dx; p->dy = out.dy; p->dz = out.dz; p->i = out.i; p->ux = out.ux; p->uy = out.uy; p->uz = out.uz;
  CK(vpic_hip_get_accumulator(c.e, a0));
  const int stuck = (int)vpic_hip_species_nm(c.e, sp);
  if (stuck) {
    vpic_particle_mover_t m;
    CK(vpic_hip_species_get_movers(c.e, sp, &m, 1));
    pm->dispx = m.dispx; pm->dispy = m.dispy; pm->dispz = m.dispz;   // pm->i keeps naming the caller's particle
    c.e->species[sp].nm = 0;
  } else {
    pm->dispx = pm->dispy = pm->dispz = 0;                            // move_p.c:103-105 leaves disp - disp = +0 behind
  }
  CK(vpic_hip_species_set_particles(c.e, sp, p0, 0));
  return stuck;
}

// ---- boundary_p (spa.h:35-43 -> boundary_p.c:77-505) on a grid of ONE rank -------------------------
// Every face then is local or wraps onto the rank itself, so the only movers left are those on
// absorbing faces: they are charged to rhob (boundary_p.c:9-71) and removed by back-filling from
// the end of the array (boundary_p.c:264).  Custom particle boundary handlers are not supported.
void vpic_hip_ref_boundary_p(vpic_species_t *sp_list, vpic_field_t *f, vpic_accumulator_t *a0, const vpic_grid_t *g, void *rng) {
  (void)rng;
  if (!f) DIE("Bad field");
  Cached &c = engine_for(g);
  need_transport(c, "boundary_p", true);
  for (int face = 0; face < 6; face++)
    if (c.e->gk.pbc[face] < VPIC_ABSORB_PARTICLES) DIE("boundary_p: custom particle boundary handlers are not supported");
  for (size_t k = 0; k < c.e->species.size(); k++) c.e->species[k].nm = 0;
  if (!multi(c, true)) {
    // Every face is local or wraps onto the rank itself: the only movers left are those on absorbing faces, charged to
    // rhob (boundary_p.c:9-71) and removed by back-filling from the end of the array (boundary_p.c:264).
    for (vpic_species_t *sp = sp_list; sp; sp = sp->next) {
      if (sp->nm == 0) continue;
      const int s = species_for(c, sp->q_m, sp->np, sp->nm);
      CK(vpic_hip_set_fields(c.e, f));
      CK(vpic_hip_species_set_particles(c.e, s, sp->p, sp->np));
      CK(vpic_hip_species_set_movers(c.e, s, sp->pm, sp->nm));
      CK(vpic_hip_boundary_p_pack(c.e));
      int32_t ns[6];
      CK(vpic_hip_boundary_p_counts(c.e, ns));
      for (int face = 0; face < 6; face++) if (ns[face]) DIE("boundary_p: a particle left for another rank on a one-rank grid");
      sp->np = (int32_t)vpic_hip_species_np(c.e, s);
      sp->nm = 0;
      CK(vpic_hip_species_get_particles(c.e, s, sp->p, sp->np));
      CK(vpic_hip_get_fields(c.e, f));
    }
    return;
  }
  // ---- a grid of several ranks: boundary_p.c:77-505 with the engine's kernels and the registered transport ----------
  // All species travel together (an injector names its species, species_advance.h:48-55): each species id of the list
  // gets an engine species, the movers are classified (absorbed into rhob / injectors per shared face), counts and
  // payloads cross axis by axis, arrivals are appended and finish their move into the accumulator; num_comm_round
  // rounds (vpic.cxx:17), ended early when no rank has a mover left.
  if (!a0) DIE("Bad accumulator");
  int n_id = 0;
  for (vpic_species_t *sp = sp_list; sp; sp = sp->next) { if (sp->id < 0 || sp->id >= 32) DIE("boundary_p: species id %d", sp->id); if (sp->id + 1 > n_id) n_id = sp->id + 1; }
  if ((int)c.bsp.size() < n_id) { c.bsp.resize(n_id, -1); c.bsp_cap.resize(n_id, 0); c.bsp_mcap.resize(n_id, 0); }
  CK(vpic_hip_set_fields(c.e, f));
  CK(vpic_hip_set_accumulator(c.e, a0));
  for (vpic_species_t *sp = sp_list; sp; sp = sp->next) {
    const int64_t want = std::max<int64_t>(sp->max_np, sp->np) + (sp->max_np >> 2) + 1024, mwant = std::max<int64_t>(sp->max_nm, 1024);
    if (c.bsp[sp->id] < 0 || c.bsp_cap[sp->id] < want || c.bsp_mcap[sp->id] < mwant) {
      c.bsp[sp->id] = vpic_hip_species_create(c.e, sp->q_m, want, mwant);
      if (c.bsp[sp->id] < 0) DIE("%s", vpic_hip_last_error());
      c.bsp_cap[sp->id] = want; c.bsp_mcap[sp->id] = mwant;
    }
    const int s = c.bsp[sp->id];
    c.e->species[s].q_m = sp->q_m;
    CK(vpic_hip_species_set_particles(c.e, s, sp->p, sp->np));
    CK(vpic_hip_species_set_movers(c.e, s, sp->pm, sp->nm));
  }
  std::vector<int> id_of(c.e->species.size(), -1);                      // engine species -> the caller's species id, and back
  for (int id = 0; id < n_id; id++) if (c.bsp[id] >= 0) id_of[c.bsp[id]] = id;
  const size_t rec = sizeof(vpic_particle_injector_t);
  for (int round = 0; round < 3; round++) {
    CK(vpic_hip_boundary_p_pack(c.e));
    int32_t ns[6], nr[6] = {0, 0, 0, 0, 0, 0};
    CK(vpic_hip_boundary_p_counts(c.e, ns));
    for (int a = 0; a < 3; a++) {                                       // the reference posts all six faces at once; the axes are independent
      if (!shared_axis(c, a, true)) continue;
      const int lo = a, hi = a + 3;
      axis_exchange(c, g, a, true, &ns[lo], 4, &ns[hi], 4, &nr[lo], 4, &nr[hi], 4);   // counts first (boundary_p.c:333-337)
      if (!shared_face(c, lo, true)) ns[lo] = nr[hi] = 0;
      if (!shared_face(c, hi, true)) ns[hi] = nr[lo] = 0;
      void *r0 = xbuf(c, 2, (size_t)nr[lo] * rec), *r3 = xbuf(c, 3, (size_t)nr[hi] * rec);
      xbuf(c, 0, (size_t)ns[lo] * rec); xbuf(c, 1, (size_t)ns[hi] * rec);
      if (ns[lo]) CK(vpic_hip_copy_to_host(c.e, &c.xhost[0][0], vpic_hip_boundary_p_send_buffer(c.e, lo), ns[lo] * rec));
      if (ns[hi]) CK(vpic_hip_copy_to_host(c.e, &c.xhost[1][0], vpic_hip_boundary_p_send_buffer(c.e, hi), ns[hi] * rec));
      for (int k = 0; k < 2; k++) {                                     // on the wire an injector names the species by the caller's id
        vpic_particle_injector_t *inj = reinterpret_cast<vpic_particle_injector_t *>(&c.xhost[k][0]);
        for (int n = 0; n < (k ? ns[hi] : ns[lo]); n++) inj[n].sp_id = id_of[inj[n].sp_id];
      }
      axis_exchange(c, g, a, true, &c.xhost[0][0], ns[lo] * rec, &c.xhost[1][0], ns[hi] * rec, &c.xhost[2][0], nr[lo] * rec, &c.xhost[3][0], nr[hi] * rec);
      for (int k = 2; k < 4; k++) {
        vpic_particle_injector_t *inj = reinterpret_cast<vpic_particle_injector_t *>(&c.xhost[k][0]);
        for (int n = 0; n < (k == 2 ? nr[lo] : nr[hi]); n++) {
          const int id = inj[n].sp_id;
          if (id < 0 || id >= n_id || c.bsp[id] < 0) DIE("boundary_p: an arriving particle names species %d", id);
          inj[n].sp_id = c.bsp[id];
        }
      }
      if (nr[lo]) { CK(vpic_hip_copy_from_host(c.e, r0, &c.xhost[2][0], nr[lo] * rec)); CK(vpic_hip_boundary_p_inject(c.e, r0, nr[lo])); }
      if (nr[hi]) { CK(vpic_hip_copy_from_host(c.e, r3, &c.xhost[3][0], nr[hi] * rec)); CK(vpic_hip_boundary_p_inject(c.e, r3, nr[hi])); }
    }
    double pending = 0;
    for (int id = 0; id < n_id; id++) if (c.bsp[id] >= 0) pending += (double)vpic_hip_species_nm(c.e, c.bsp[id]);
    allsum(g, &pending, 1);
    if (pending == 0) break;
  }
  for (vpic_species_t *sp = sp_list; sp; sp = sp->next) {
    const int s = c.bsp[sp->id];
    const int64_t np = vpic_hip_species_np(c.e, s);
    if (np > sp->max_np) {                                              // boundary_p.c:416-432: the array grows by 1.3125 of what is needed
      const int64_t cap = (int64_t)(np * 1.3125) + 1;
      vpic_particle_t *bigger = (vpic_particle_t *)ref_aligned(sizeof(vpic_particle_t) * (size_t)cap, 128);
      ref_aligned_free(sp->p);
      sp->p = bigger; sp->max_np = (int32_t)cap;
    }
    sp->np = (int32_t)np;
    sp->nm = 0;
    c.e->species[s].nm = 0;                                             // (a mover still pending after the last round is dropped, boundary_p.c:498-503)
    CK(vpic_hip_species_get_particles(c.e, s, sp->p, sp->np));
    CK(vpic_hip_species_set_particles(c.e, s, sp->p, 0));               // the engine keeps nothing between calls
  }
  CK(vpic_hip_get_fields(c.e, f));
  CK(vpic_hip_get_accumulator(c.e, a0));
}

// field_advance_methods_t, slot by slot (src/field_advance/field_advance.h:185-302); host pass only
#if !defined(__HIP_DEVICE_COMPILE__)
void *const vpic_hip_ref_field_advance_methods[20] = {
  (void *)vpic_hip_ref_new_field, (void *)vpic_hip_ref_delete_field,
  (void *)vpic_hip_ref_new_material_coefficients, (void *)vpic_hip_ref_delete_material_coefficients,
  (void *)vpic_hip_ref_advance_b, (void *)vpic_hip_ref_advance_e, (void *)vpic_hip_ref_energy_f,
  (void *)vpic_hip_ref_clear_jf, (void *)vpic_hip_ref_synchronize_jf, (void *)vpic_hip_ref_clear_rhof,
  (void *)vpic_hip_ref_synchronize_rho, (void *)vpic_hip_ref_compute_rhob, (void *)vpic_hip_ref_compute_curl_b,
  (void *)vpic_hip_ref_synchronize_tang_e_norm_b, (void *)vpic_hip_ref_compute_div_e_err,
  (void *)vpic_hip_ref_compute_rms_div_e_err, (void *)vpic_hip_ref_clean_div_e, (void *)vpic_hip_ref_compute_div_b_err,
  (void *)vpic_hip_ref_compute_rms_div_b_err, (void *)vpic_hip_ref_clean_div_b};
#endif

}  // extern "C"

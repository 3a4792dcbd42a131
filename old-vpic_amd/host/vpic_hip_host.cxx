// vpic_hip_host.cxx -- see vpic_hip_host.hxx.  The reference functions each piece stands in for
// are cited; state that the reference keeps in host arrays lives in the HIP engine here.
#include "vpic_hip_host.hxx"
#include <cstdarg>
#include <climits>
#include <vector>
#include <map>
#include <algorithm>
#include <ctime>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

vpic_simulation *vpic_host_current = NULL;

void vpic_host_log(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
}

#define CK(call) do { if (call) ERROR(("%s", vpic_hip_last_error())); } while (0)

// ---- message passing between domains (src/util/mp: mp_init, mp_rank/mp_nproc, mp_allsum_d and the
// port send/receive pairs of grid_comm.c) ------------------------------------------------------------
// Compiled with -DVPIC_HIP_HOST_MPI the host is one MPI rank per domain / GPU: x-slab decompositions
// (define_*_grid with gpx = nproc, gpy = gpz = 1), messages staged through host memory.  Without it
// there is one domain and these are no-ops.
#ifdef VPIC_HIP_HOST_MPI
#include <mpi.h>
static int g_mp_rank = 0, g_mp_nproc = 1;
static double g_mp_t0;
void vpic_host_mp_init(int *argc, char ***argv) { MPI_Init(argc, argv); g_mp_t0 = MPI_Wtime(); MPI_Comm_rank(MPI_COMM_WORLD, &g_mp_rank); MPI_Comm_size(MPI_COMM_WORLD, &g_mp_nproc); }
void vpic_host_mp_finalize(void) { MPI_Finalize(); }
static void mp_allsum_d(double *v, int n) { std::vector<double> t(v, v + n); MPI_Allreduce(&t[0], v, n, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD); }
// a message travelling in direction d (0..2: towards -x, -y, -z; 3..5: towards +) is tagged d: it goes to the neighbour on
// that side and is received from the neighbour on the other side (grid_comm.c:7-78).  A neighbour that is this rank itself
// (self-sends across a periodic axis, grid_comm.c:17-19,49) is served by a copy.
static void mp_exchange6(const void *const s[6], const size_t ns[6], void *const r[6], const size_t nr[6], const int to[6], const int from[6]) {
  MPI_Request rq[12]; int k = 0;
  for (int d = 0; d < 6; d++) if (nr[d] && from[d] != g_mp_rank) MPI_Irecv(r[d], (int)nr[d], MPI_BYTE, from[d], d, MPI_COMM_WORLD, &rq[k++]);
  for (int d = 0; d < 6; d++) if (ns[d] && to[d] != g_mp_rank) MPI_Isend(const_cast<void *>(s[d]), (int)ns[d], MPI_BYTE, to[d], d, MPI_COMM_WORLD, &rq[k++]);
  for (int d = 0; d < 6; d++) if (ns[d] && to[d] == g_mp_rank) memcpy(r[d], s[d], ns[d] < nr[d] ? ns[d] : nr[d]);
  MPI_Waitall(k, rq, MPI_STATUSES_IGNORE);
}
static void mp_bcast(void *buf, int bytes) { MPI_Bcast(buf, bytes, MPI_BYTE, 0, MPI_COMM_WORLD); }
static double mp_allmax_d(double v) { double m = v; MPI_Allreduce(&v, &m, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD); return m; }
double mp_elapsed(void *) { double t = MPI_Wtime() - g_mp_t0, m = t; MPI_Allreduce(&t, &m, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD); return m; }   // mp_dmp.c: max over ranks
void mp_barrier(void *) { MPI_Barrier(MPI_COMM_WORLD); }
void mp_finalize(void *) { MPI_Finalize(); }
void mp_send_i(int *buf, int n, int dst, void *) { MPI_Send(buf, n, MPI_INT, dst, 0, MPI_COMM_WORLD); }
void mp_recv_i(int *buf, int n, int src, void *) { MPI_Recv(buf, n, MPI_INT, src, 0, MPI_COMM_WORLD, MPI_STATUS_IGNORE); }
#else
#include <chrono>
static const std::chrono::steady_clock::time_point g_mp_t0 = std::chrono::steady_clock::now();
double mp_elapsed(void *) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_mp_t0).count(); }
void mp_barrier(void *) {}
void mp_finalize(void *) {}
void mp_send_i(int *, int, int, void *) {}
void mp_recv_i(int *, int, int, void *) {}
static const int g_mp_rank = 0, g_mp_nproc = 1;
void vpic_host_mp_init(int *, char ***) {}
void vpic_host_mp_finalize(void) {}
static void mp_allsum_d(double *, int) {}
static void mp_exchange6(const void *const s[6], const size_t ns[6], void *const r[6], const size_t nr[6], const int to[6], const int *) {
  for (int d = 0; d < 6; d++) if (ns[d] && to[d] == 0) memcpy(r[d], s[d], ns[d] < nr[d] ? ns[d] : nr[d]);   // one rank: self-sends only
}
static void mp_bcast(void *, int) {}
static double mp_allmax_d(double v) { return v; }
#endif
// which transport moves the messages between domains (see "exchanges with the neighbouring domains" below)
enum { XPORT_NONE = 0, XPORT_MPI = 1, XPORT_RCCL = 2 };
static bool self_sends(void) { const char *v = getenv("VPIC_HIP_HOST_SELF_SEND"); return g_mp_nproc == 1 && v && atoi(v); }
static int chosen_transport(void) {
  if (g_mp_nproc == 1 && !self_sends()) return XPORT_NONE;
  const char *v = getenv("VPIC_HIP_HOST_TRANSPORT");
  if (v && !strcmp(v, "mpi")) return XPORT_MPI;
  if (v && *v && strcmp(v, "rccl")) ERROR(("VPIC_HIP_HOST_TRANSPORT=%s: rccl or mpi", v));
  return XPORT_RCCL;
}
int vpic_host_mp_rank(void) { return g_mp_rank; }
int vpic_host_mp_nproc(void) { return g_mp_nproc; }

// ---- host mirrors on demand (VPIC_HIP_MIRROR=demand) -------------------------------------------------------
// field, interpolator and every species' particle array live in page-aligned blocks.  While the engine works
// they are inaccessible (PROT_NONE); the first access from deck code raises SIGSEGV, the handler brings the
// array over from the device and makes it readable; a first WRITE to a readable array raises it again and
// the array is marked dirty; when the deck's hook returns, dirty arrays go back to the device and everything
// is made inaccessible again.  So an unchanged deck can read and modify fields and particles in any of its
// hooks (antennas in user_field_injection, collisions, tracer tagging ...) and pays only for what it touches.
// One thing a fault cannot catch: the kernel reading a protected array on the deck's behalf (write(), fwrite
// of a large block) fails with EFAULT instead -- FileIO and the deck wrapper's fwrite touch the array first.
namespace {
enum { M_PLAIN = 0, M_STALE, M_CLEAN, M_DIRTY };
struct Mirror { char *base; size_t bytes, mapped; int kind, sp, state; };   // kind 0 fields, 1 interpolator, 2 particles of species sp
std::vector<Mirror> g_mirrors;
bool g_demand = false, g_handler_installed = false;
struct sigaction g_old_segv;
void mirror_protect(Mirror &m, int prot) { if (mprotect(m.base, m.mapped, prot) != 0) { perror("mprotect"); abort(); } }
void mirror_download(Mirror &m);
void on_segv(int sig, siginfo_t *si, void *ctx) {
  char *a = (char *)si->si_addr;
  for (size_t k = 0; k < g_mirrors.size(); k++) {
    Mirror &m = g_mirrors[k];
    if (a < m.base || a >= m.base + m.mapped) continue;
    if (m.state == M_STALE) { mirror_protect(m, PROT_READ | PROT_WRITE); mirror_download(m); mirror_protect(m, PROT_READ); m.state = M_CLEAN; return; }
    if (m.state == M_CLEAN) { mirror_protect(m, PROT_READ | PROT_WRITE); m.state = M_DIRTY; return; }
    break;
  }
  // not one of ours (or a genuine fault): hand over to whoever was there before and let the access fault again
  sigaction(SIGSEGV, &g_old_segv, NULL);
  (void)sig; (void)ctx;
}
void *mirror_alloc(size_t bytes, int kind, int sp) {
  const size_t page = (size_t)sysconf(_SC_PAGESIZE), mapped = (bytes + page - 1) / page * page + page;
  void *p = mmap(NULL, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) ERROR(("Failed to allocate %lu bytes.", (unsigned long)bytes));
  Mirror m = {(char *)p, bytes, mapped, kind, sp, M_PLAIN};
  g_mirrors.push_back(m);
  return p;
}
}  // namespace
// make [p, p+bytes) safe to hand to the kernel (write, fwrite): fault the mirrors it overlaps in now
void vpic_host_touch(const void *p, size_t bytes) {
  if (!g_demand || !p || !bytes) return;
  const char *a = (const char *)p;
  for (size_t k = 0; k < g_mirrors.size(); k++) {
    const Mirror &m = g_mirrors[k];
    if (a < m.base + m.mapped && a + bytes > m.base) { volatile char c = *(volatile const char *)(a < m.base ? m.base : a); (void)c; }
  }
}

void vpic_host_touch_for_write(void *p, size_t bytes) {
  if (!g_demand || !p || !bytes) return;
  char *a = (char *)p;
  for (size_t k = 0; k < g_mirrors.size(); k++) {
    const Mirror &m = g_mirrors[k];
    if (a < m.base + m.mapped && a + bytes > m.base) { volatile char *c = (volatile char *)(a < m.base ? m.base : a); *c = *c; }
  }
}

// ---- field_advance->method table ---------------------------------------------------------------
static void host_energy_f(double *en, const field_t *f, const material_coefficient_t *m, const grid_t *g) {
  if (vpic_host_current && vpic_host_current->resident_energy_f(en, f)) { mp_allsum_d(en, 6); return; }   // energy_f.c:172
  vpic_hip_ref_energy_f(en, f, m, g);
}
// A deck that calls the table's entries itself hands over host arrays; when those are the simulation's own
// mirrors they have to be resident (and dirty, these entries write them) before a HIP copy may read them:
// the device's DMA engines do not raise SIGSEGV, they fault.
static void host_advance_b(field_t *f, const grid_t *g, float frac) { vpic_hip_ref_advance_b(f, g, frac); }
static void host_advance_e(field_t *f, const material_coefficient_t *m, const grid_t *g) { vpic_hip_ref_advance_e(f, m, g); }
static void host_clear_jf(field_t *f, const grid_t *g) { vpic_hip_ref_clear_jf(f, g); }
static void host_synchronize_jf(field_t *f, const grid_t *g) { vpic_hip_ref_synchronize_jf(f, g); }
field_advance_methods_t standard_field_advance[1] = {{ host_advance_b, host_advance_e, host_energy_f, host_clear_jf, host_synchronize_jf }};

double energy_p(const particle_t *p0, int np, float q_m, const interpolator_t *f0, const grid_t *g) {
  if (vpic_host_current && vpic_host_current->owns(p0)) {
    double en = vpic_host_current->resident_energy_p(p0);
    mp_allsum_d(&en, 1);                                  // energy_p.cxx:155
    return en;
  }
  return vpic_hip_ref_energy_p(p0, np, q_m, f0, g);
}

// ---- MT19937 (Matsumoto & Nishimura) with the reference's seeding and 53-bit conversion ---------
// src/util/mtrand/mtrand.c:69-76 (seed), :43-46 (draw + temper), mtrand_conv.h:61 (drand53_o)
static void mt_seed(mt_rng_t *r, unsigned seed) {
  r->next = 624;
  r->state[0] = seed ^ 0x900df00cu;
  for (int j = 1; j < 624; j++) r->state[j] = 1812433253u * (r->state[j - 1] ^ (r->state[j - 1] >> 30)) + (unsigned)j;
}
static uint32_t mt_u32(mt_rng_t *r) {
  if (r->next == 624) {
    uint32_t *s = r->state;
    for (int k = 0; k < 624; k++) {
      const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
      s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    r->next = 0;
  }
  uint32_t y = r->state[r->next++];
  y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
  return y;
}
static double mt_drand(mt_rng_t *r) {
  const uint32_t a = mt_u32(r), b = mt_u32(r);
  return ((a >> 5) * 67108864. + (b >> 6) + 1.5) * (1. / 9007199254740994.);
}

// ---- construction ------------------------------------------------------------------------------
vpic_simulation::vpic_simulation() {       // src/vpic/vpic.cxx:13-49
  verbose = 1; step = 0; num_step = 0; num_comm_round = 3; status_interval = 0;
  clean_div_e_interval = clean_div_b_interval = sync_shared_interval = 0;
  quota = 11; restart_interval = hydro_interval = field_interval = particle_interval = 0;
  rng = NULL; grid = NULL; species_list = NULL; emitter_list = NULL; field_advance = NULL; field = NULL;
  interpolator = NULL; accumulator = NULL;
  memset(user_global, 0, sizeof(user_global));
  hip_mirror_interval = 1;
  hip_adaptive_sort = 1;   // a species' sort_interval is the upper bound; the engine sorts earlier when that pays (hot plasmas)
  // an unchanged deck cannot set the HIP knobs: the environment can
  if (const char *v = getenv("VPIC_HIP_MIRROR_INTERVAL")) hip_mirror_interval = atoi(v);
  if (const char *v = getenv("VPIC_HIP_ADAPTIVE_SORT")) hip_adaptive_sort = atoi(v);
  for (int f = 0; f < 6; f++) face_rank[f] = -1;
  comm = NULL; hip_transport = XPORT_NONE; hip_resident_exchange = false; hip_deterministic = false; x_mover_cap = 0; x_flags = 0; x_messages = x_syncs = x_recoveries = 0;
  engine = NULL; mirrors_current = false; movers_pending = false;
  for (int a = 0; a < 3; a++) { topo_index[a] = 0; topo_size[a] = 1; }
  px = py = pz = 1;
  vpic_host_current = this;
}

vpic_simulation::~vpic_simulation() {
  if (engine) {
    if (comm) vpic_hip_comm_destroy(comm);
    for (std::map<int, XBuf>::iterator it = xbufs.begin(); it != xbufs.end(); ++it) vpic_hip_device_free(engine, it->second.dev);
    vpic_hip_destroy(engine);
  }
  vpic_host_current = NULL;
}

// ---- grid: partition_*_box for one rank (src/grid/partition.c:35-85, ops.c:25-231) -------------
void vpic_simulation::box(double xl, double yl, double zl, double xh, double yh, double zh,
                          int nx, int ny, int nz, int pbc, int fbc) {
  grid_t *g = grid;
  g->dx = (xh - xl) / (double)nx; g->dy = (yh - yl) / (double)ny; g->dz = (zh - zl) / (double)nz;
  g->rdx = (double)nx / (xh - xl); g->rdy = (double)ny / (yh - yl); g->rdz = (double)nz / (zh - zl);
  g->x0 = xl; g->y0 = yl; g->z0 = zl; g->x1 = xh; g->y1 = yh; g->z1 = zh;
  g->nx = nx; g->ny = ny; g->nz = nz;
  const int64_t sy = nx + 2, sz = sy * (ny + 2), nv = sz * (nz + 2);
  for (int k = 0; k < 27; k++) g->bc[k] = pec_fields;      // size_grid (ops.c:41-45); only faces are joined / set
  g->bc[13] = 0;
  g->bc[BOUNDARY(-1, 0, 0)] = g->bc[BOUNDARY(1, 0, 0)] = g->bc[BOUNDARY(0, -1, 0)] = g->bc[BOUNDARY(0, 1, 0)] =
      g->bc[BOUNDARY(0, 0, -1)] = g->bc[BOUNDARY(0, 0, 1)] = fbc;
  g->range = (int64_t *)malloc(2 * sizeof(int64_t));
  g->range[0] = 0; g->range[1] = nv;
  g->rangel = 0; g->rangeh = nv - 1;
  g->neighbor = (int64_t *)malloc(6 * nv * sizeof(int64_t));
  const int n[3] = {nx, ny, nz};
  const int64_t stride[3] = {1, sy, sz};
  for (int64_t z = 0; z <= nz + 1; z++) for (int64_t y = 0; y <= ny + 1; y++) for (int64_t x = 0; x <= nx + 1; x++) {
    const int64_t v = x + sy * y + sz * z, c[3] = {x, y, z};
    const bool ghost = x == 0 || x == nx + 1 || y == 0 || y == ny + 1 || z == 0 || z == nz + 1;
    for (int f = 0; f < 6; f++) {
      const int a = f % 3, hi = f >= 3;
      int64_t nb;
      if (ghost) nb = reflect_particles;
      else if (hi ? c[a] < n[a] : c[a] > 1) nb = v + (hi ? stride[a] : -stride[a]);
      else nb = pbc >= 0 ? v + (hi ? -(n[a] - 1) : (n[a] - 1)) * stride[a] : pbc;   // join_grid wrap / set_pbc
      g->neighbor[6 * v + f] = nb;
    }
  }
}

// partition_periodic_box / partition_metal_box (src/grid/partition.c:35-131): the global box cut into
// gpx x gpy x gpz equal bricks, rank = ix + gpx*(iy + gpy*iz) (RANK_TO_INDEX); cell sizes from the GLOBAL
// box, this rank's extent by the reference's interpolation formula.  (The name is historic: x-slabs are
// the gpy = gpz = 1 case.)
void vpic_simulation::slab(double gx0, double gy0, double gz0, double gx1, double gy1, double gz1,
                           int gnx, int gny, int gnz, int gpx, int gpy, int gpz, int pbc, int fbc, bool periodic) {
  if (gpx < 1 || gpy < 1 || gpz < 1 || gpx * gpy * gpz != g_mp_nproc)
    ERROR(("Bad domain decomposition (%ix%ix%i) for %i processes", gpx, gpy, gpz, g_mp_nproc));
  if (gnx % gpx || gny % gpy || gnz % gpz) ERROR(("Incompatible res"));
  px = (size_t)gpx; py = (size_t)gpy; pz = (size_t)gpz;
  const int gp[3] = {gpx, gpy, gpz}, gn[3] = {gnx, gny, gnz};
  const double lo[3] = {gx0, gy0, gz0}, hi[3] = {gx1, gy1, gz1};
  topo_index[0] = g_mp_rank % gpx; topo_index[1] = (g_mp_rank / gpx) % gpy; topo_index[2] = g_mp_rank / (gpx * gpy);
  for (int a = 0; a < 3; a++) topo_size[a] = gp[a];
  float e0[3], e1[3];
  for (int a = 0; a < 3; a++) {
    double f;
    f = (double)topo_index[a] / (double)gp[a];       e0[a] = lo[a] * (1 - f) + hi[a] * f;
    f = (double)(topo_index[a] + 1) / (double)gp[a]; e1[a] = lo[a] * (1 - f) + hi[a] * f;
  }
  box(e0[0], e0[1], e0[2], e1[0], e1[1], e1[2], gnx / gpx, gny / gpy, gnz / gpz, pbc, fbc);
  grid_t *g = grid;
  g->dx = (gx1 - gx0) / (double)gnx; g->rdx = (double)gnx / (gx1 - gx0);
  g->dy = (gy1 - gy0) / (double)gny; g->rdy = (double)gny / (gy1 - gy0);
  g->dz = (gz1 - gz0) / (double)gnz; g->rdz = (double)gnz / (gz1 - gz0);
  g->x0 = e0[0]; g->y0 = e0[1]; g->z0 = e0[2]; g->x1 = e1[0]; g->y1 = e1[1]; g->z1 = e1[2];
  for (int f = 0; f < 6; f++) face_rank[f] = -1;
  static const int lob[3] = {BOUNDARY(-1, 0, 0), BOUNDARY(0, -1, 0), BOUNDARY(0, 0, -1)}, hib[3] = {BOUNDARY(1, 0, 0), BOUNDARY(0, 1, 0), BOUNDARY(0, 0, 1)};
  const int stride[3] = {1, gpx, gpx * gpy};
  if (periodic)                                          // faces that wrap onto this same rank: join_grid(g, face, rank)
    for (int a = 0; a < 3; a++) g->bc[lob[a]] = g->bc[hib[a]] = g_mp_rank;
  for (int a = 0; a < 3; a++) {
    if (gp[a] == 1) {                                    // wraps onto this rank: handled on the device -- or, on request, by messages to self
      if (periodic && self_sends() && gn[a] > 1) face_rank[a] = face_rank[a + 3] = g_mp_rank;
      continue;
    }                                                    // join_grid (ops.c:135-182) on the faces shared with a neighbour
    const int i = topo_index[a];
    const int left = g_mp_rank + ((i + gp[a] - 1) % gp[a] - i) * stride[a], right = g_mp_rank + ((i + 1) % gp[a] - i) * stride[a];
    if (periodic || i > 0)         { g->bc[lob[a]] = left;  face_rank[a] = left; }
    if (periodic || i < gp[a] - 1) { g->bc[hib[a]] = right; face_rank[a + 3] = right; }
  }
  g->bc[13] = g_mp_rank;
}
void vpic_simulation::define_periodic_grid(double xl, double yl, double zl, double xh, double yh, double zh,
                                           double gnx, double gny, double gnz, double gpx, double gpy, double gpz) {
  slab(xl, yl, zl, xh, yh, zh, (int)gnx, (int)gny, (int)gnz, (int)gpx, (int)gpy, (int)gpz, 0, 0, true);
}
void vpic_simulation::define_reflecting_grid(double xl, double yl, double zl, double xh, double yh, double zh,
                                             double gnx, double gny, double gnz, double gpx, double gpy, double gpz) {
  slab(xl, yl, zl, xh, yh, zh, (int)gnx, (int)gny, (int)gnz, (int)gpx, (int)gpy, (int)gpz, reflect_particles, pec_fields, false);
}

// partition_absorbing_box (src/grid/partition.c:86-137): the periodic box, then every face on the outside of the
// global box -- on axes with more than one cell -- absorbs fields and gives particles the bc asked for
void vpic_simulation::define_absorbing_grid(double xl, double yl, double zl, double xh, double yh, double zh,
                                            double gnx, double gny, double gnz, double gpx, double gpy, double gpz, int pbc) {
  slab(xl, yl, zl, xh, yh, zh, (int)gnx, (int)gny, (int)gnz, (int)gpx, (int)gpy, (int)gpz, 0, 0, true);
  const int gn[3] = {(int)gnx, (int)gny, (int)gnz};
  const bool outer_lo[3] = {topo_index[0] == 0, topo_index[1] == 0, topo_index[2] == 0};
  const bool outer_hi[3] = {topo_index[0] == topo_size[0] - 1, topo_index[1] == topo_size[1] - 1, topo_index[2] == topo_size[2] - 1};
  static const int lo[3] = {BOUNDARY(-1, 0, 0), BOUNDARY(0, -1, 0), BOUNDARY(0, 0, -1)}, hi[3] = {BOUNDARY(1, 0, 0), BOUNDARY(0, 1, 0), BOUNDARY(0, 0, 1)};
  for (int a = 0; a < 3; a++) {
    if (gn[a] <= 1) continue;
    if (outer_lo[a]) { set_domain_field_bc(lo[a], absorb_fields); set_domain_particle_bc(lo[a], pbc); face_rank[a] = -1; }
    if (outer_hi[a]) { set_domain_field_bc(hi[a], absorb_fields); set_domain_particle_bc(hi[a], pbc); face_rank[a + 3] = -1; }
  }
}

void vpic_simulation::set_domain_field_bc(int boundary, int fbc) {      // set_fbc, ops.c:184-197
  if (boundary < 0 || boundary >= 27 || boundary == 13) ERROR(("Bad boundary"));
  grid->bc[boundary] = fbc;
  // a boundary condition proper replaces the link to whoever was behind the face (a neighbour, or -- sending to itself -- this rank)
  static const int b2f[27] = {-1,-1,-1,-1,2,-1,-1,-1,-1, -1,1,-1,0,-1,3,-1,4,-1, -1,-1,-1,-1,5,-1,-1,-1,-1};
  if (fbc < 0 && b2f[boundary] >= 0) face_rank[b2f[boundary]] = -1;
}
void vpic_simulation::set_domain_particle_bc(int boundary, int pbc) {   // set_pbc, ops.c:199-231
  static const int b2f[27] = {-1,-1,-1,-1,2,-1,-1,-1,-1, -1,1,-1,0,-1,3,-1,4,-1, -1,-1,-1,-1,5,-1,-1,-1,-1};
  const int f = (boundary >= 0 && boundary < 27) ? b2f[boundary] : -1;
  if (f < 0) ERROR(("Bad boundary"));
  if (pbc < 0) face_rank[f] = -1;                        // (see set_domain_field_bc)
  const grid_t *g = grid;
  const int n[3] = {g->nx, g->ny, g->nz}, a = f % 3, plane = f < 3 ? 1 : n[a];
  const int64_t sy = g->nx + 2, sz = sy * (g->ny + 2);
  int lo[3] = {1, 1, 1}, hi[3] = {n[0], n[1], n[2]};
  lo[a] = hi[a] = plane;
  for (int z = lo[2]; z <= hi[2]; z++) for (int y = lo[1]; y <= hi[1]; y++) for (int x = lo[0]; x <= hi[0]; x++)
    grid->neighbor[6 * (x + sy * y + sz * z) + f] = pbc;
}

// ---- materials: new_material (material.c:25-70) and, at finalize_field_advance, their coefficient
// records (new_material_coefficients, src/field_advance/standard/sfa.c:80-177) ------------------------
material_id vpic_simulation::define_material(const char *name, double epsx, double epsy, double epsz, double mux, double muy,
                                             double muz, double sigmax, double sigmay, double sigmaz,
                                             double zetax, double zetay, double zetaz) {
  if (!name || !name[0]) ERROR(("Cannot create a nameless material."));
  if (lookup_material(name) != invalid_material_id) ERROR(("There is already a material named \"%s\".", name));
  if (zetax != 0 || zetay != 0 || zetaz != 0) WARNING(("Standard field advance does not support magnetic conductivity yet."));
  if (field_advance) ERROR(("materials must be defined before finalize_field_advance"));
  material_rec rec = {name, {(float)epsx, (float)epsy, (float)epsz}, {(float)mux, (float)muy, (float)muz},
                      {(float)sigmax, (float)sigmay, (float)sigmaz}};
  material_records.push_back(rec);
  return (material_id)(material_records.size() - 1);
}
material_id vpic_simulation::define_material(const char *name, double eps, double mu, double sigma, double zeta) {
  return define_material(name, eps, eps, eps, mu, mu, mu, sigma, sigma, sigma, zeta, zeta, zeta);
}
material_id vpic_simulation::lookup_material(const char *name) {
  for (size_t k = 0; name && k < material_records.size(); k++) if (material_records[k].name == name) return (material_id)k;
  return invalid_material_id;
}
// sfa.c:145-177: decay = exp(-sigma dt / (eps eps0)); drive = the exactly integrated source weight (1/eps
// without conductivity, 0 for a perfect conductor to numerical precision); double exp / sinh on float
// operands, stored as float, as the reference does
static vpic_material_coefficient_t material_coefficients(const float *eps, const float *mu, const float *sigma, float dt, float eps0) {
  vpic_material_coefficient_t mc;
  memset(&mc, 0, sizeof(mc));
  float a[3], decay[3], drive[3];
  for (int k = 0; k < 3; k++) {
    a[k] = (sigma[k] * dt) / (eps[k] * eps0);
    decay[k] = exp(-a[k]);
    if (a[k] == 0) drive[k] = 1. / eps[k];
    else if (decay[k] == 0) drive[k] = 0;
    else drive[k] = 2. * exp(-0.5 * a[k]) * sinh(0.5 * a[k]) / (a[k] * eps[k]);
  }
  mc.decayx = decay[0]; mc.decayy = decay[1]; mc.decayz = decay[2];
  mc.drivex = drive[0]; mc.drivey = drive[1]; mc.drivez = drive[2];
  mc.rmux = 1. / mu[0]; mc.rmuy = 1. / mu[1]; mc.rmuz = 1. / mu[2];
  mc.nonconductive = (a[0] == 0 && a[1] == 0 && a[2] == 0) ? 1. : 0.;
  mc.epsx = eps[0]; mc.epsy = eps[1]; mc.epsz = eps[2];
  return mc;
}

void vpic_simulation::finalize_field_advance(field_advance_methods_t *fam) {   // vpic.hxx:373-400
  if (material_records.empty()) ERROR(("Empty material list."));
  materials.clear();
  for (size_t k = 0; k < material_records.size(); k++) {
    const material_rec &m = material_records[k];
    materials.push_back(material_coefficients(m.eps, m.mu, m.sigma, grid->dt, grid->eps0));
  }
  const size_t nv = (size_t)(grid->nx + 2) * (grid->ny + 2) * (grid->nz + 2);
  field_advance = new field_advance_t;
  field = (field_t *)mirror_alloc(nv * sizeof(field_t), 0, -1);
  interpolator = (interpolator_t *)mirror_alloc(nv * sizeof(interpolator_t), 1, -1);
  accumulator = (accumulator_t *)calloc(nv + 1, sizeof(accumulator_t));
  field_advance->f = field; field_advance->m = &materials[0]; field_advance->g = grid;
  field_advance->method[0] = fam[0];
  vpic_hip_ref_set_material_count((int)materials.size());
}

// ---- species: new_species (species_advance.c:21-63) via define_species (vpic.hxx:407-420) --------
species_t *vpic_simulation::define_species(const char *name, double q_m, double max_local_np, double max_local_nm,
                                           double sort_interval, double sort_out_of_place) {
  if (max_local_nm <= -1) {
    max_local_nm = 2 * max_local_np / 25;
    if (max_local_nm < 16 * 17) max_local_nm = 16 * 17;
  }
  if (max_local_np < 1) ERROR(("Bad max_local_np"));
  const size_t len = strlen(name);
  species_t *sp = (species_t *)calloc(1, sizeof(species_t) + len);
  strcpy(sp->name, name);
  sp->id = (int)species_order.size();
  sp->max_np = (int)max_local_np; sp->max_nm = (int)max_local_nm;
  sp->p = (particle_t *)mirror_alloc((size_t)sp->max_np * sizeof(particle_t), 2, sp->id);
  sp->pm = (particle_mover_t *)calloc((size_t)sp->max_nm, sizeof(particle_mover_t));
  sp->q_m = (float)q_m; sp->sort_interval = (int)sort_interval; sp->sort_out_of_place = (int)sort_out_of_place;
  sp->next = species_list;                              // new_species pushes on the front of the list
  species_list = sp;
  species_order.push_back(sp);
  return sp;
}
species_t *vpic_simulation::find_species(const char *name) {
  species_t *sp;
  LIST_FOR_EACH(sp, species_list) if (strcmp(sp->name, name) == 0) return sp;
  return NULL;
}

// ---- inject_particle: src/vpic/misc.cxx:16-105 (no aging, no rhob update on this path yet) -------
void vpic_simulation::inject_particle(species_t *sp, double x, double y, double z, double ux, double uy, double uz,
                                      double q, int64_t tag, double age, int update_rhob) {
  if (!grid) ERROR(("Grid not setup yet"));
  if (!accumulator) ERROR(("Accumulator not setup yet"));
  if (!sp) ERROR(("Invalid species"));
  if (age != 0 && !engine) ERROR(("inject_particle with an age before the run has started is not supported by this host"));
  // update_rhob before the run starts has no effect in the reference either: initialize() recomputes rhob from
  // div E and the loaded charge (initialize.cxx:56-60).  Once the run is under way it is applied on the device.
  const double x0 = (double)grid->x0, y0 = (double)grid->y0, z0 = (double)grid->z0;
  const double x1 = (double)grid->x1, y1 = (double)grid->y1, z1 = (double)grid->z1;
  const int nx = grid->nx, ny = grid->ny, nz = grid->nz;
  if ((x < x0) | (x > x1) | ((x == x1) & (grid->bc[BOUNDARY(1, 0, 0)] >= 0))) return;
  if ((y < y0) | (y > y1) | ((y == y1) & (grid->bc[BOUNDARY(0, 1, 0)] >= 0))) return;
  if ((z < z0) | (z > z1) | ((z == z1) & (grid->bc[BOUNDARY(0, 0, 1)] >= 0))) return;
  if (sp->np >= sp->max_np) ERROR(("No room to inject particle"));
  int ix, iy, iz;
  x = ((double)nx) * ((x - x0) / (x1 - x0)); ix = (int)x; x -= (double)ix; x = (x + x) - 1;
  if (ix == nx) x = 1; if (ix == nx) ix = nx - 1; ix++;
  y = ((double)ny) * ((y - y0) / (y1 - y0)); iy = (int)y; y -= (double)iy; y = (y + y) - 1;
  if (iy == ny) y = 1; if (iy == ny) iy = ny - 1; iy++;
  z = ((double)nz) * ((z - z0) / (z1 - z0)); iz = (int)z; z -= (double)iz; z = (z + z) - 1;
  if (iz == nz) z = 1; if (iz == nz) iz = nz - 1; iz++;
  particle_t one;
  memset(&one, 0, sizeof(one));
  particle_t *p = engine ? &one : sp->p + (sp->np++);
  p->dx = (float)x; p->dy = (float)y; p->dz = (float)z;
  p->i = INDEX_FORTRAN_3(ix, iy, iz, 0, nx + 1, 0, ny + 1, 0, nz + 1);
  p->ux = (float)ux; p->uy = (float)uy; p->uz = (float)uz; p->q = q; p->tag = tag;
  if (engine && update_rhob) injected_rhob.push_back(one);
  if (engine && age != 0) {                               // misc.cxx:93-103: the part of the step it has already lived through
    int id = -1;
    for (size_t k = 0; k < species_order.size(); k++) if (species_order[k] == sp) id = (int)k;
    if (id < 0) ERROR(("injection into a species this simulation does not hold"));
    age *= grid->cvac * grid->dt / sqrt(ux * ux + uy * uy + uz * uz + 1);
    particle_injector_t inj;
    inj.dx = one.dx; inj.dy = one.dy; inj.dz = one.dz; inj.i = one.i; inj.ux = one.ux; inj.uy = one.uy; inj.uz = one.uz; inj.q = one.q;
    inj.dispx = ux * age * grid->rdx; inj.dispy = uy * age * grid->rdy; inj.dispz = uz * age * grid->rdz; inj.sp_id = id;
    injected_aged.push_back(inj);
    injected_aged_tags.push_back(one.tag); injected_aged_tags.push_back(one.tag2);
    movers_pending = true;                                // it may stop on a face
  } else if (engine) queue_injected(sp, one);
}
// vpic.hxx:463-470: no checks, as in the reference
void vpic_simulation::inject_particle_raw(species_t *sp, float dx, float dy, float dz, int32_t i, float ux, float uy, float uz, float q) {
  particle_t one;
  memset(&one, 0, sizeof(one));
  particle_t *p = engine ? &one : sp->p + (sp->np++);
  p->dx = dx; p->dy = dy; p->dz = dz; p->i = i; p->ux = ux; p->uy = uy; p->uz = uz; p->q = q;
  if (engine) queue_injected(sp, one);
}
// Once the engine holds the particles, the host arrays are mirrors: an injected particle waits in a
// list and is appended to the device species when the deck's injection call returns (flush_injected).
void vpic_simulation::queue_injected(species_t *sp, const particle_t &p) {
  int id = -1;
  for (size_t k = 0; k < species_order.size(); k++) if (species_order[k] == sp) id = (int)k;
  if (id < 0) ERROR(("injection into a species this simulation does not hold"));
  if (injected.size() < species_order.size()) injected.resize(species_order.size());
  injected[id].push_back(p);
}
void vpic_simulation::flush_injected(void) {
  if (!injected_aged.empty()) {
    CK(vpic_hip_inject_aged(engine, &injected_aged[0], &injected_aged_tags[0], (int)injected_aged.size()));
    injected_aged.clear(); injected_aged_tags.clear();
    mirrors_current = false;
  }
  if (!injected_rhob.empty()) {                           // misc.cxx:87-91
    CK(vpic_hip_accumulate_rhob(engine, &injected_rhob[0], (int64_t)injected_rhob.size(), -1.f));
    injected_rhob.clear();
  }
  for (size_t k = 0; k < injected.size(); k++) {
    if (injected[k].empty()) continue;
    CK(vpic_hip_species_append_particles(engine, (int)k, &injected[k][0], (int64_t)injected[k].size()));
    injected[k].clear();
    mirrors_current = false;
  }
}

void vpic_simulation::seed_rand(double seed) { mt_seed(rng, (unsigned)(int)seed); }
double vpic_simulation::uniform_rand(double low, double high) { const double dx = mt_drand(rng); return low * (1 - dx) + high * dx; }
// The reference's normal generator is a 256-layer ziggurat on the same Mersenne twister (mtrand.c:395-440) whose
// tables are a data file of the reference: they cannot be regenerated (their entries are several ulp off the exactly
// rounded values) and are not copied.  So by default this host draws normals with Box-Muller: decks that call
// maxwellian_rand load statistically equivalent particles.  For parity runs, VPIC_HIP_NORMALS=<prefix> replays the
// normals a REFERENCE run of the same deck drew (<prefix>.<rank>: int64 n, n doubles, n bytes = generator words each
// draw consumed; recorded by oracle/normals_shim.c through the reference's public API): the value is returned and the
// twister is advanced by the same number of words, so the uniforms drawn in between stay the reference's too.
static struct NormalReplay {
  int state = 0;                              // 0 not looked yet, 1 replaying, 2 off
  std::vector<double> value; std::vector<unsigned char> words; size_t next = 0;
} g_normals;
double vpic_simulation::maxwellian_rand(double dev) {
  NormalReplay &R = g_normals;
  if (R.state == 0) {
    R.state = 2;
    if (const char *prefix = getenv("VPIC_HIP_NORMALS")) {
      char path[4096];
      snprintf(path, sizeof(path), "%s.%d", prefix, g_mp_rank);
      FILE *f = fopen(path, "rb");
      long long n = 0;
      if (!f || fread(&n, 8, 1, f) != 1 || n < 0) ERROR(("VPIC_HIP_NORMALS: cannot read %s", path));
      R.value.resize((size_t)n); R.words.resize((size_t)n);
      if (fread(R.value.data(), 8, (size_t)n, f) != (size_t)n || fread(R.words.data(), 1, (size_t)n, f) != (size_t)n)
        ERROR(("VPIC_HIP_NORMALS: %s is truncated", path));
      fclose(f);
      R.state = 1;
    }
  }
  if (R.state == 1) {
    if (R.next >= R.value.size()) ERROR(("VPIC_HIP_NORMALS: the deck draws more than the %zu recorded normals", R.value.size()));
    for (int k = 0; k < (int)R.words[R.next]; k++) (void)mt_u32(rng);
    return dev * R.value[R.next++];
  }
  const double u1 = mt_drand(rng), u2 = mt_drand(rng);       // Box-Muller
  return dev * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2);
}

// ---- engine plumbing -----------------------------------------------------------------------------
void vpic_simulation::describe(vpic_hip_grid_t &d) {
  const grid_t *g = grid;
  d.dt = g->dt; d.cvac = g->cvac; d.eps0 = g->eps0; d.damp = g->damp;
  d.dx = g->dx; d.dy = g->dy; d.dz = g->dz; d.rdx = g->rdx; d.rdy = g->rdy; d.rdz = g->rdz;
  d.nx = g->nx; d.ny = g->ny; d.nz = g->nz; d.rank = g_mp_rank;
  static const int fb[6] = {BOUNDARY(-1,0,0), BOUNDARY(0,-1,0), BOUNDARY(0,0,-1), BOUNDARY(1,0,0), BOUNDARY(0,1,0), BOUNDARY(0,0,1)};
  const int n[3] = {g->nx, g->ny, g->nz};
  const int64_t sy = g->nx + 2, sz = sy * (g->ny + 2);
  for (int f = 0; f < 6; f++) {
    d.fbc[f] = g->bc[fb[f]];
    const int a = f % 3;
    const int64_t c[3] = {a == 0 ? (f < 3 ? 1 : n[0]) : 1, a == 1 ? (f < 3 ? 1 : n[1]) : 1, a == 2 ? (f < 3 ? 1 : n[2]) : 1};
    const int64_t nb = g->neighbor[6 * (c[0] + sy * c[1] + sz * c[2]) + f];
    d.pbc[f] = nb < 0 ? (int)nb : g_mp_rank;
    if (d.fbc[f] >= 0) d.fbc[f] = g_mp_rank;            // periodic onto this same domain ...
    // ... or shared with a neighbour.  A neighbour that is this rank itself (self-sends) is given a code of its own: to the
    // engine a face is shared when its code names another domain
    if (face_rank[f] >= 0) d.fbc[f] = d.pbc[f] = face_rank[f] == g_mp_rank ? g_mp_nproc + g_mp_rank : face_rank[f];
  }
}

int vpic_simulation::resident_id(const particle_t *p0) const {
  if (!engine) return -1;
  for (size_t k = 0; k < species_order.size(); k++) if (species_order[k]->p == p0) return (int)k;
  return -1;
}
bool vpic_simulation::owns(const particle_t *p0) const {
  for (size_t k = 0; k < species_order.size(); k++) if (species_order[k]->p == p0) return engine != NULL;
  return false;
}
double vpic_simulation::resident_energy_p(const particle_t *p0) {
  for (size_t k = 0; k < species_order.size(); k++) if (species_order[k]->p == p0) {
    double en = 0;
    CK(vpic_hip_energy_p(engine, (int)k, &en));
    return en;
  }
  return 0;
}
bool vpic_simulation::resident_energy_f(double *en, const field_t *f) {
  if (!engine || f != field) return false;
  CK(vpic_hip_energy_f(engine, en));
  return true;
}

namespace {
void mirror_download(Mirror &m) {
  vpic_simulation *sim = vpic_host_current;
  if (!sim) return;
  sim->mirror_download(m.kind, m.sp);
}
}  // namespace
void vpic_simulation::mirror_download(int kind, int sp) {
  if (kind == 0) CK(vpic_hip_get_fields(engine, field));
  else if (kind == 1) CK(vpic_hip_get_interpolator(engine, interpolator));
  else {
    species_t *s = species_order[sp];
    pending_sort(sp);                                       // (the hook sees the species as sorted as the reference's would be)
    const int64_t np = vpic_hip_species_np(engine, sp);
    if (np > s->max_np) ERROR(("species %s outgrew its host mirror", s->name));
    CK(vpic_hip_species_get_particles(engine, sp, s->p, s->max_np));
    s->np = (int)np;
  }
}
// demand mode, after the engine changed state: counts are cheap to keep current, arrays become inaccessible
void vpic_simulation::mirrors_stale(void) {
  if (!g_demand || !engine) return;
  for (size_t k = 0; k < species_order.size(); k++) species_order[k]->np = (int)vpic_hip_species_np(engine, (int)k);
  for (size_t k = 0; k < g_mirrors.size(); k++) {
    Mirror &m = g_mirrors[k];
    if (m.state != M_STALE) { mirror_protect(m, PROT_NONE); m.state = M_STALE; }
  }
}
// demand mode, when a deck hook returns: what it wrote goes to the device
void vpic_simulation::mirrors_after_user_code(void) {
  if (!g_demand || !engine) return;
  bool fields_changed = false;
  for (size_t k = 0; k < g_mirrors.size(); k++) {
    Mirror &m = g_mirrors[k];
    if (m.state != M_DIRTY) continue;
    if (m.kind == 0) { CK(vpic_hip_set_fields(engine, field)); fields_changed = true; }
    else if (m.kind == 1) CK(vpic_hip_set_interpolator(engine, interpolator));
    else CK(vpic_hip_species_set_particles(engine, m.sp, species_order[m.sp]->p, species_order[m.sp]->np));
  }
  (void)fields_changed;
  mirrors_stale();
}
// what the library calls before a HIP copy reads or writes caller memory (vpic_hip_set_host_access_hook): the ONE
// place where a protected mirror is made resident on behalf of the device -- a copy engine that meets a PROT_NONE
// page faults the GPU instead of raising SIGSEGV
static void host_access(const void *p, size_t bytes, int for_write) {
  if (for_write) vpic_host_touch_for_write(const_cast<void *>(p), bytes); else vpic_host_touch(p, bytes);
}
void vpic_simulation::start_demand_mirrors(void) {
  // (an "eager" scheme -- whole arrays refreshed before user_diagnostics every N steps, hook edits never pushed
  // back -- existed and was removed: hooks that edit fields silently lost their edits)
  if (const char *mode = getenv("VPIC_HIP_MIRROR"))
    if (strcmp(mode, "demand") != 0 && vpic_host_mp_rank() == 0)
      fprintf(stderr, "hip host: VPIC_HIP_MIRROR=%s is not supported any more; host mirrors are kept coherent on demand\n", mode);
  g_demand = true;
  vpic_hip_set_host_access_hook(host_access);
  if (!g_handler_installed) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &g_old_segv);
    g_handler_installed = true;
  }
  mirrors_stale();
}

void vpic_simulation::hip_sync_mirrors(void) {
  if (g_demand) {                                         // bring over what is stale, leave it readable
    for (size_t k = 0; k < g_mirrors.size(); k++) {
      Mirror &m = g_mirrors[k];
      if (m.state == M_STALE) { mirror_protect(m, PROT_READ | PROT_WRITE); ::mirror_download(m); mirror_protect(m, PROT_READ); m.state = M_CLEAN; }
    }
    mirrors_current = true;
    return;
  }
  CK(vpic_hip_get_fields(engine, field));
  CK(vpic_hip_get_interpolator(engine, interpolator));
  for (size_t k = 0; k < species_order.size(); k++) {
    species_t *sp = species_order[k];
    const int64_t np = vpic_hip_species_np(engine, (int)k);
    if (np > sp->max_np) ERROR(("species %s outgrew its host mirror", sp->name));
    CK(vpic_hip_species_get_particles(engine, (int)k, sp->p, sp->max_np));
    sp->np = (int)np; sp->nm = 0;
  }
  mirrors_current = true;
}
void vpic_simulation::hip_upload_mirrors(void) {
  if (g_demand) { mirrors_after_user_code(); CK(vpic_hip_load_interpolator(engine)); return; }   // what the deck wrote, nothing else
  CK(vpic_hip_set_fields(engine, field));
  for (size_t k = 0; k < species_order.size(); k++)
    CK(vpic_hip_species_set_particles(engine, (int)k, species_order[k]->p, species_order[k]->np));
  CK(vpic_hip_load_interpolator(engine));
  mirrors_stale();
}

// ---- exchanges with the neighbouring domains -----------------------------------------------------------------------
// Messages are DEVICE buffers the engine packs and unpacks.  Two transports move them (hip_transport, chosen once):
//   rccl -- the default with more than one rank: vpic_hip_comm_* (include/vpic_hip.h), RCCL point-to-point over xGMI on a
//           communication stream, ordered against the engine's stream with events; the host never waits for a message.
//           MPI only launches the ranks and carries the 128-byte communicator id.  One rank per GPU.
//   mpi  -- has to be asked for (VPIC_HIP_HOST_TRANSPORT=mpi): every message staged through host memory, blocking -- for
//           boxes where several ranks must share one device (RCCL refuses that) and for MPI builds without RCCL.
// ONE rank with VPIC_HIP_HOST_SELF_SEND=1: the periodic axes are cut into faces shared with this same rank and every
// message is sent to self (what the reference does on a periodic rank of its own, grid_comm.c:17-19,49: MPI_Issend to
// itself) -- the whole multi-domain path, RCCL included, on a one-GPU box.
const char *vpic_simulation::transport_name(void) const {
  return hip_transport == XPORT_RCCL ? "rccl" : hip_transport == XPORT_MPI ? "mpi (host-staged)" : "none";
}

enum { XB_SEND = 0, XB_RECV = 1 };
// a device buffer per (kind, direction, tag), grown on demand; new memory is cleared
void *vpic_simulation::xbuf(int kind, int d, int tag, size_t bytes) {
  XBuf &b = xbufs[(tag * 6 + d) * 2 + kind];
  if (bytes > b.bytes) {
    if (b.dev) { CK(vpic_hip_sync(engine)); vpic_hip_device_free(engine, b.dev); }
    b.bytes = bytes + bytes / 4 + 4096;
    b.dev = vpic_hip_device_alloc(engine, b.bytes);
    if (!b.dev) ERROR(("%s", vpic_hip_last_error()));
    b.host.assign(b.bytes, 0);
    CK(vpic_hip_copy_from_host(engine, b.dev, &b.host[0], b.bytes));
  }
  return b.dev;
}
std::vector<char> &vpic_simulation::xhost_of(void *dev) {
  for (std::map<int, XBuf>::iterator it = xbufs.begin(); it != xbufs.end(); ++it) if (it->second.dev == dev) return it->second.host;
  ERROR(("not an exchange buffer"));
  return xbufs[0].host;
}
// Post one exchange: x.s[d] (x.ns[d] bytes) travels in direction d to the rank behind face d; x.r[d] receives the message
// travelling in direction d, from the rank behind the opposite face.  Returns a token for x_finish (rccl) or -1 (done).
int vpic_simulation::x_start(const XferSet &x) {
  int to[6], from[6];
  for (int d = 0; d < 6; d++) { to[d] = face_rank[d]; from[d] = face_rank[(d + 3) % 6]; }
  x_messages++;
  if (hip_transport == XPORT_RCCL) {
    const void *sb[6]; size_t sn[6]; int sp[6]; void *rb[6]; size_t rn[6]; int rp[6]; int ns = 0, nr = 0;
    for (int d = 0; d < 6; d++) if (x.ns[d]) { if (to[d] < 0) ERROR(("send across a face without a neighbour")); sb[ns] = x.s[d]; sn[ns] = x.ns[d]; sp[ns++] = to[d]; }
    for (int d = 0; d < 6; d++) if (x.nr[d]) { if (from[d] < 0) ERROR(("receive across a face without a neighbour")); rb[nr] = x.r[d]; rn[nr] = x.nr[d]; rp[nr++] = from[d]; }
    int token = -1;
    CK(vpic_hip_comm_start(comm, ns, sb, sn, sp, nr, rb, rn, rp, &token));
    return token;
  }
  // host-staged: device -> host, MPI (or a copy, to self), host -> device; the host blocks
  const void *hs[6]; void *hr[6];
  for (int d = 0; d < 6; d++) {
    hs[d] = hr[d] = NULL;
    if (x.ns[d]) { std::vector<char> &h = xhost_of(const_cast<void *>(x.s[d])); CK(vpic_hip_copy_to_host(engine, &h[0], x.s[d], x.ns[d])); hs[d] = &h[0]; }
    if (x.nr[d]) hr[d] = &xhost_of(x.r[d])[0];
  }
  mp_exchange6(hs, x.ns, hr, x.nr, to, from);
  for (int d = 0; d < 6; d++) if (x.nr[d]) CK(vpic_hip_copy_from_host(engine, x.r[d], hr[d], x.nr[d]));
  return -1;
}
void vpic_simulation::x_finish(int token) { if (token >= 0) CK(vpic_hip_comm_finish(comm, token)); }

// one message of `bytes` each way along an axis: pack(dir, device buffer), exchange, unpack(dir, device buffer)
template <class Pack, class Unpack>
void vpic_simulation::plane_exchange(int axis, size_t bytes, Pack pack, Unpack unpack, int tag) {
  const int lo = axis, hi = axis + 3;
  XferSet x; memset(&x, 0, sizeof(x));
  for (int k = 0; k < 2; k++) {
    const int d = k ? hi : lo;
    if (face_rank[d] >= 0) { void *b = xbuf(XB_SEND, d, tag, bytes); pack(d, b); x.s[d] = b; x.ns[d] = bytes; }
    if (face_rank[(d + 3) % 6] >= 0) { x.r[d] = xbuf(XB_RECV, d, tag, bytes); x.nr[d] = bytes; }      // travelling in direction d: from the opposite side
  }
  x_finish(x_start(x));
  if (x.nr[lo]) unpack(lo, x.r[lo]);
  if (x.nr[hi]) unpack(hi, x.r[hi]);
}
bool vpic_simulation::shared(int axis) const { return face_rank[axis] >= 0 || face_rank[axis + 3] >= 0; }
bool vpic_simulation::multi(void) const { return shared(0) || shared(1) || shared(2); }

void vpic_simulation::pending_sort(int id) {
  if ((size_t)id < sort_pending.size() && sort_pending[(size_t)id]) { sort_pending[(size_t)id] = 0; CK(vpic_hip_sort_p(engine, id)); }
}
void vpic_simulation::resident_advance_p(int id) {
  if ((size_t)id < sort_pending.size() && sort_pending[(size_t)id]) { sort_pending[(size_t)id] = 0; CK(vpic_hip_sort_advance_p(engine, id)); }
  else CK(vpic_hip_advance_p(engine, id));
  movers_pending = true;
}
void vpic_simulation::resident_boundary_p(void) { if (movers_pending) x_boundary_p(); }
// The reference's own protocol -- counts first, payload second, per round (boundary_p.c:77-505, advance.cxx:94-96): what
// a deck with custom particle boundary handlers (maxwellian_reflux) runs, and any deck on request (VPIC_HIP_HOST_EXCHANGE=legacy)
void vpic_simulation::x_boundary_p(void) {
  movers_pending = false;
  if (multi() && hip_resident_exchange) { x_exchange_rounds(num_comm_round); return; }
  for (int round = 0; round < num_comm_round; round++) {
    CK(vpic_hip_boundary_p_pack(engine));
    if (!multi()) continue;
    int32_t ns[6], nr[6] = {0, 0, 0, 0, 0, 0};
    CK(vpic_hip_boundary_p_counts(engine, ns));
    const size_t rec = sizeof(particle_injector_t);
    int to[6], from[6];
    for (int d = 0; d < 6; d++) { to[d] = face_rank[d]; from[d] = face_rank[(d + 3) % 6]; }
    {                                                       // the counts of all six faces at once (boundary_p.c:333-337)
      const void *s[6]; void *r[6]; size_t sn[6], rn[6];
      for (int d = 0; d < 6; d++) { s[d] = &ns[d]; r[d] = &nr[d]; sn[d] = to[d] >= 0 ? 4 : 0; rn[d] = from[d] >= 0 ? 4 : 0; if (to[d] < 0) ns[d] = 0; }
      mp_exchange6(s, sn, r, rn, to, from);
    }
    XferSet x; memset(&x, 0, sizeof(x));
    for (int d = 0; d < 6; d++) {
      if (ns[d]) { x.s[d] = xbuf(XB_SEND, d, 60, (size_t)ns[d] * rec); x.ns[d] = (size_t)ns[d] * rec; CK(vpic_hip_boundary_p_get_injectors(engine, d, const_cast<void *>(x.s[d]))); }
      if (nr[d]) { x.r[d] = xbuf(XB_RECV, d, 60, (size_t)nr[d] * rec); x.nr[d] = (size_t)nr[d] * rec; }
    }
    x_finish(x_start(x));
    for (int d = 0; d < 6; d++) if (nr[d]) CK(vpic_hip_boundary_p_inject(engine, x.r[d], nr[d]));
    // a round in which no domain has a mover left does nothing: stop as soon as that is known
    double pending = 0;
    for (size_t k = 0; k < species_order.size(); k++) pending += (double)vpic_hip_species_nm(engine, (int)k);
    mp_allsum_d(&pending, 1);
    if (pending == 0) break;
  }
}

// ---- the device-resident exchange, overlapped with the push (what old-vpic_amd/domain.py's SlabDomain.push_and_exchange
// does over torch.distributed, here on the transports above) -----------------------------------------------------------
// The reference's begin / interior / end pattern (advance_e.c:114,153,191-197) applied to boundary_p.c:341-384.  Per
// species: push the tiles on the shared faces (and what arrived since the sort), pack the species' movers into one
// fixed-capacity message per shared face, start the transfer, push the interior tiles behind it -- species k is on the wire
// while its own interior and species k + 1 are pushed.  Then the arrivals join their species; small later rounds (all
// species in one message per face: stragglers the interior launches left on a face, particles an earlier round delivered
// onto yet another boundary; one round more than there are cut axes, num_comm_round at most) follow.  Counts stay on the
// device: ONE read-back (vpic_hip_exchange_finish) ends the step's exchange.  Removals leave dead slots; a message that
// was full parks its movers and both of its ends run an extra round (x_recover).
static int round_cap(double n) { const int g = 4096; const long c = ((long)n + g - 1) / g * g; return (int)(c < g ? g : c > (1 << 26) ? (1 << 26) : c); }
static int next_cap(int wanted) { return round_cap(1.5 * wanted + 4096); }     // both ends evaluate this on the same header
static size_t msg_bytes(int cap) { return 16 + 48 * (size_t)cap; }

int &vpic_simulation::x_cap(int kind, int d, int k) {
  const int key = (k * 6 + d) * 2 + kind;
  std::map<int, int>::iterator it = x_caps.find(key);
  if (it != x_caps.end()) return it->second;
  int c = 4096;
  if (k >= 0 && (size_t)k < x_np_max.size()) {
    // first use: an eighth of what a boundary plane of cells holds, from the LARGEST population of this species on any
    // rank (both ends of a message must begin with the same capacity: x_np_max, taken collectively)
    const grid_t *g = grid;
    const int n[3] = {g->nx, g->ny, g->nz};
    c = round_cap(x_np_max[(size_t)k] / n[d % 3] / 8);
  }
  return x_caps[key] = c;
}

vpic_simulation::XRound vpic_simulation::x_round(int tag, const int cs[6], const int cr[6], int mover_cap, uint32_t species_mask) {
  XRound r; memset(&r, 0, sizeof(r));
  r.tag = tag;
  void *ptrs[6]; int32_t caps[6];
  XferSet x; memset(&x, 0, sizeof(x));
  for (int d = 0; d < 6; d++) {
    ptrs[d] = NULL; caps[d] = 0; r.cs[d] = cs[d]; r.cr[d] = cr[d];
    if (cs[d] > 0) { r.ms[d] = xbuf(XB_SEND, d, tag, msg_bytes(cs[d])); ptrs[d] = r.ms[d]; caps[d] = cs[d]; x.s[d] = r.ms[d]; x.ns[d] = msg_bytes(cs[d]); }
    if (cr[d] > 0) { r.mr[d] = xbuf(XB_RECV, d, tag, msg_bytes(cr[d])); x.r[d] = r.mr[d]; x.nr[d] = msg_bytes(cr[d]); }
  }
  CK(vpic_hip_exchange_pack_species(engine, species_mask, ptrs, caps, mover_cap));
  r.token = x_start(x);
  return r;
}
void vpic_simulation::x_land(const XRound &r) {
  x_finish(r.token);
  for (int d = 0; d < 6; d++) if (r.mr[d]) CK(vpic_hip_exchange_inject(engine, r.mr[d], r.cr[d]));
}
// vpic_hip_exchange_finish over the messages of `log`: headers {count, wanted, 0, 0}, received first, then sent, per round
void vpic_simulation::x_read_back(const std::vector<XRound> &log, std::vector<XHeader> &H) {
  std::vector<const void *> ptr;
  H.clear();
  for (size_t j = 0; j < log.size(); j++)
    for (int kind = XB_RECV; kind >= XB_SEND; kind--)
      for (int d = 0; d < 6; d++) {
        void *m = kind == XB_RECV ? log[j].mr[d] : log[j].ms[d];
        if (!m) continue;
        XHeader h; h.kind = kind; h.tag = log[j].tag; h.d = d; h.count = h.wanted = 0;
        H.push_back(h); ptr.push_back(m);
      }
  std::vector<int32_t> raw(4 * ptr.size() + 4);
  int32_t flags = 0;
  CK(vpic_hip_exchange_finish(engine, ptr.empty() ? NULL : &ptr[0], (int)ptr.size(), &raw[0], &flags));
  for (size_t j = 0; j < H.size(); j++) { H[j].count = raw[4 * j]; H[j].wanted = raw[4 * j + 1]; }
  x_flags = flags;
  x_syncs++;
}
// A message that was full left its movers parked on their lists, the particles untouched (the reference grows its buffers
// instead, boundary_p.c:131-150, 416-448; here both ends must know a message's size beforehand).  Both ends of such a
// message read the same header -- wanted > count -- so exactly the two ranks concerned run an extra round over that face.
void vpic_simulation::x_recover(std::vector<XHeader> &H) {
  const int ns = (int)species_order.size();
  for (int attempt = 0; attempt < 4; attempt++) {
    long left = 0;
    for (int k = 0; k < ns; k++) left += (long)vpic_hip_species_nm(engine, k);
    int over_s[6] = {0, 0, 0, 0, 0, 0}, over_r[6] = {0, 0, 0, 0, 0, 0}; bool any = false;
    for (int d = 0; d < 6; d++) {
      int need[2] = {0, 0}; bool over[2] = {false, false};
      for (size_t j = 0; j < H.size(); j++) if (H[j].d == d) { need[H[j].kind] = std::max(need[H[j].kind], H[j].wanted); if (H[j].wanted > H[j].count) over[H[j].kind] = true; }
      if (over[XB_SEND]) { over_s[d] = round_cap(need[XB_SEND] + 1); any = true; }
      if (over[XB_RECV]) { over_r[d] = round_cap(need[XB_RECV] + 1); any = true; }
    }
    if (!any) {
      if (left) ERROR(("boundary_p: %ld movers left after the step's rounds (a particle crossed more domains than that in one step, or more movers than the exchange kernels were launched for: flags %d)", left, x_flags));
      return;
    }
    x_recoveries++;
    CK(vpic_hip_exchange_begin(engine));
    XRound r = x_round(40 + attempt, over_s, over_r, 1 << 30, ~0u);
    x_land(r);
    std::vector<XRound> log(1, r);
    x_read_back(log, H);
  }
  ERROR(("boundary_p: messages kept overflowing"));
}
// keep the species' arrays from running out between sorts: arrivals are appended, departures leave dead slots until the
// next sort.  From 85 % full: sort now when that frees at least 5 % of the array, otherwise enlarge it by the reference's
// growth factor (boundary_p.c:416-448); the mover list likewise.
void vpic_simulation::x_make_room(void) {
  for (size_t k = 0; k < species_order.size(); k++) {
    int64_t extent = 0, max_np = 0, max_nm = 0;
    CK(vpic_hip_species_capacity(engine, (int)k, &extent, &max_np, &max_nm));
    if (extent > 0.85 * max_np) {
      if (extent - vpic_hip_species_np(engine, (int)k) > 0.05 * max_np) CK(vpic_hip_sort_p(engine, (int)k));
      else CK(vpic_hip_species_reserve(engine, (int)k, (int64_t)(max_np * 1.3125) + 4096, max_nm));
    }
    if (x_mover_cap && x_mover_cap > 0.7 * max_nm) CK(vpic_hip_species_reserve(engine, (int)k, 0, (int64_t)(std::max<int64_t>(max_nm, x_mover_cap) * 1.3125) + 4096));
  }
}
// rounds that carry every species (what is left after the per-species messages; the movers of particles a deck emitted or
// injected; the species a deck advances itself)
void vpic_simulation::x_exchange_rounds(int rounds) {
  movers_pending = false;
  CK(vpic_hip_exchange_begin(engine));
  std::vector<XRound> log;
  for (int rnd = 0; rnd < rounds; rnd++) {
    int cs[6], cr[6];
    for (int d = 0; d < 6; d++) { cs[d] = face_rank[d] >= 0 ? x_cap(XB_SEND, d, -1) : 0; cr[d] = face_rank[(d + 3) % 6] >= 0 ? x_cap(XB_RECV, d, -1) : 0; }
    XRound r = x_round(32 + rnd, cs, cr, 1 << 30, ~0u);
    x_land(r);
    log.push_back(r);
  }
  std::vector<XHeader> H;
  x_read_back(log, H);
  for (size_t j = 0; j < H.size(); j++) { int &c = x_cap(H[j].kind, H[j].d, -1); if (H[j].wanted > c / 2) c = std::max(c, round_cap(4.0 * H[j].wanted)); }
  x_recover(H);
  x_make_room();
}
void vpic_simulation::x_push_and_exchange(const std::vector<char> &listed) {
  vpic_hip_engine_t *e = engine;
  const int ns = (int)species_order.size();
  movers_pending = false;
  if (x_np_max.size() != (size_t)ns) {                      // (collective: every rank, every species, once)
    x_np_max.assign((size_t)ns, 0.0);
    for (int k = 0; k < ns; k++) x_np_max[(size_t)k] = mp_allmax_d((double)vpic_hip_species_np(e, k));
  }
  // (sorts drop the dead slots earlier exchanges left: the particle counts must be final before the exchange puts them on the device)
  for (int k = 0; k < ns; k++) if (listed[k]) pending_sort(k);
  CK(vpic_hip_exchange_begin(e));
  const int mover_cap = x_mover_cap ? x_mover_cap : (1 << 30);
  int n_axes = 0;
  for (int a = 0; a < 3; a++) if (shared(a)) n_axes++;
  const int rounds = std::min(n_axes + 1, num_comm_round);
  std::vector<XRound> log;
  for (int k = 0; k < ns; k++) {
    if (!listed[k]) continue;
    CK(vpic_hip_advance_p_phase(e, k, 1));                  // the tiles on the shared faces and the appended particles
    int cs[6], cr[6];
    for (int d = 0; d < 6; d++) { cs[d] = face_rank[d] >= 0 ? x_cap(XB_SEND, d, k) : 0; cr[d] = face_rank[(d + 3) % 6] >= 0 ? x_cap(XB_RECV, d, k) : 0; }
    log.push_back(x_round(k, cs, cr, mover_cap, 1u << k)); // this species' movers are on the wire ...
    CK(vpic_hip_advance_p_phase(e, k, 2));                  // ... while its interior is pushed
  }
  const size_t flights = log.size();
  for (size_t j = 0; j < flights; j++) x_land(log[j]);
  for (int rnd = 1; rnd < rounds; rnd++) {
    int cs[6], cr[6];
    for (int d = 0; d < 6; d++) { cs[d] = face_rank[d] >= 0 ? x_cap(XB_SEND, d, -1) : 0; cr[d] = face_rank[(d + 3) % 6] >= 0 ? x_cap(XB_RECV, d, -1) : 0; }
    XRound r = x_round(32 + rnd, cs, cr, mover_cap, ~0u);
    x_land(r);
    log.push_back(r);
  }
  std::vector<XHeader> H;
  x_read_back(log, H);
  // capacities of the next step, from what each sender wanted to send in this one (both ends read the same header)
  std::vector<long> per_species((size_t)ns, 0);
  for (size_t j = 0; j < H.size(); j++) {
    const XHeader &h = H[j];
    if (h.tag < 32) { x_cap(h.kind, h.d, h.tag) = next_cap(h.wanted); if (h.kind == XB_SEND) per_species[(size_t)h.tag] += h.wanted; }
    else { int &c = x_cap(h.kind, h.d, -1); if (h.wanted > c / 2) c = std::max(c, round_cap(4.0 * h.wanted)); }
  }
  long most = 0;
  for (int k = 0; k < ns; k++) most = std::max(most, per_species[(size_t)k]);
  x_mover_cap = (int)std::max<long>(65536, 2 * most + 4096);   // the movers of a species leave through ALL its shared faces
  x_recover(H);
  x_make_room();
}
// advance_e with the remote tangential-B ghosts fetched first (advance_e.c:114,153,191-197: begin_remote_ghost_tang_b,
// the interior, end_remote_ghost_tang_b): all shared faces in one exchange (ghost planes need no edge propagation,
// remote.c:61-134); with x the only cut axis the planes x = 2..nx, which read none of the ghosts, are advanced while the
// messages travel
void vpic_simulation::x_advance_e(void) {
  vpic_hip_engine_t *e = engine;
  if (!multi()) { CK(vpic_hip_advance_e(e)); return; }
  XferSet x; memset(&x, 0, sizeof(x));
  for (int d = 0; d < 6; d++) {
    const size_t bytes = sizeof(float) * (size_t)vpic_hip_face_count(e, d % 3);
    if (face_rank[d] >= 0) { void *b = xbuf(XB_SEND, d, 51, bytes); CK(vpic_hip_pack_tang_b(e, d, b)); x.s[d] = b; x.ns[d] = bytes; }
    if (face_rank[(d + 3) % 6] >= 0) { x.r[d] = xbuf(XB_RECV, d, 51, bytes); x.nr[d] = bytes; }
  }
  const int token = x_start(x);
  const bool split = shared(0) && !shared(1) && !shared(2);
  if (split) CK(vpic_hip_advance_e_part(e, 1));
  x_finish(token);
  for (int d = 0; d < 6; d++) if (x.nr[d]) CK(vpic_hip_unpack_tang_b(e, d, x.r[d]));
  if (split) CK(vpic_hip_advance_e_part(e, 2)); else CK(vpic_hip_advance_e(e));
}
void vpic_simulation::x_tang_b(void) {                     // remote.c:61-134
  vpic_hip_engine_t *e = engine;
  for (int a = 0; a < 3; a++)
    if (shared(a)) plane_exchange(a, sizeof(float) * (size_t)vpic_hip_face_count(e, a),
                                  [e](int d, void *b) { CK(vpic_hip_pack_tang_b(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_tang_b(e, d, b)); });
}
void vpic_simulation::x_synchronize_jf(void) { // remote.c:416-506
  if (!multi()) { CK(vpic_hip_synchronize_jf(engine)); return; }
  vpic_hip_engine_t *e = engine;
  CK(vpic_hip_local_adjust_jf(e));
  for (int a = 0; a < 3; a++) {                          // x, then y, then z: edges and corners propagate (remote.c:284-289)
    if (shared(a)) plane_exchange(a, sizeof(float) * (size_t)vpic_hip_face_count(e, a),
                                  [e](int d, void *b) { CK(vpic_hip_pack_jf(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_jf(e, d, b)); });
    else CK(vpic_hip_synchronize_jf_self(e, a));
  }
}
void vpic_simulation::x_synchronize_rho(void) { // remote.c:533-622
  if (!multi()) { CK(vpic_hip_synchronize_rho(engine)); return; }
  vpic_hip_engine_t *e = engine;
  CK(vpic_hip_local_adjust_rho(e));
  for (int a = 0; a < 3; a++) {                          // x, then y, then z: edges and corners propagate (remote.c:284-289)
    if (shared(a)) plane_exchange(a, sizeof(float) * (size_t)vpic_hip_rho_count(e, a),
                                  [e](int d, void *b) { CK(vpic_hip_pack_rho(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_rho(e, d, b)); });
    else CK(vpic_hip_synchronize_rho_self(e, a));
  }
}
void vpic_simulation::x_synchronize_hydro(void) { // sf_interface/hydro.c:28-163
  if (!multi()) { CK(vpic_hip_synchronize_hydro(engine)); return; }
  vpic_hip_engine_t *e = engine;
  CK(vpic_hip_local_adjust_hydro(e));
  for (int a = 0; a < 3; a++) {                          // x, then y, then z: edges and corners propagate (remote.c:284-289)
    if (shared(a)) plane_exchange(a, sizeof(float) * (size_t)vpic_hip_hydro_count(e, a),
                                  [e](int d, void *b) { CK(vpic_hip_pack_hydro(e, d, b)); }, [e](int d, void *b) { CK(vpic_hip_unpack_hydro(e, d, b)); });
    else CK(vpic_hip_synchronize_hydro_self(e, a));
  }
}
double vpic_simulation::x_message(int kind, int axis) {    // normal E / div_b_err ghosts, tang E + norm B averages, one axis
  double err = 0;
  vpic_hip_engine_t *e = engine;
  double *perr = &err;
  plane_exchange(axis, sizeof(float) * (size_t)vpic_hip_face_message_count(e, kind, axis),
                 [e, kind](int d, void *b) { CK(vpic_hip_pack_face_message(e, kind, d, b)); },
                 [e, kind, perr](int d, void *b) { double x = 0; CK(vpic_hip_unpack_face_message(e, kind, d, b, &x)); *perr += x; });
  return err;
}
double vpic_simulation::x_message(int kind) {              // ... every shared axis
  double err = 0;
  for (int a = 0; a < 3; a++) if (shared(a)) err += x_message(kind, a);
  return err;
}
double vpic_simulation::x_synchronize_tang_e_norm_b(void) { // remote.c:298-414
  double err = 0, x;
  if (!multi()) { CK(vpic_hip_synchronize_tang_e_norm_b(engine, &err)); return err; }
  CK(vpic_hip_local_adjust_tang_e_norm_b(engine));
  for (int a = 0; a < 3; a++) {
    if (shared(a)) err += x_message(VPIC_HIP_MSG_TANG_E_NORM_B, a);
    else { CK(vpic_hip_synchronize_tang_e_norm_b_self(engine, a, &x)); err += x; }
  }
  mp_allsum_d(&err, 1);
  return err;
}
double vpic_simulation::x_rms(bool e_field) {              // compute_rms_div_{e,b}_err.c: two sums over all ranks
  double l2[2];
  CK(e_field ? vpic_hip_rms_div_e_err_local(engine, l2) : vpic_hip_rms_div_b_err_local(engine, l2));
  mp_allsum_d(l2, 2);
  return grid->eps0 * sqrt(l2[0] / l2[1]);
}
void vpic_simulation::x_accumulate_rho(void) {             // advance.cxx:155-158
  CK(vpic_hip_clear_rhof(engine));
  for (size_t k = 0; k < species_order.size(); k++) CK(vpic_hip_accumulate_rho_p(engine, (int)k));
  x_synchronize_rho();
}
void vpic_simulation::x_compute_div_e_err(void) { x_message(VPIC_HIP_MSG_NORM_E); CK(vpic_hip_compute_div_e_err(engine)); }
void vpic_simulation::x_compute_rhob(void) { x_message(VPIC_HIP_MSG_NORM_E); CK(vpic_hip_compute_rhob(engine)); }
void vpic_simulation::x_clean_div_b(void) { x_message(VPIC_HIP_MSG_DIV_B); CK(vpic_hip_clean_div_b(engine)); }
void vpic_simulation::x_compute_curl_b(void) { x_tang_b(); CK(vpic_hip_compute_curl_b(engine)); }

enum { MAX_REFLUX_SPECIES = 32 };
// the resident engine for the state the host arrays describe (grid, materials, fields, species)
void vpic_simulation::create_engine(void) {
  vpic_hip_grid_t d;
  describe(d);
  const int ndev = vpic_hip_device_count();
  CK(vpic_hip_create(&engine, &d, (g_mp_nproc > 1 && ndev > 0) ? g_mp_rank % ndev : -1));   // one rank per GPU (shared when there are fewer)
  CK(vpic_hip_set_material_coefficients(engine, &materials[0], (int)materials.size()));
  // Deterministic sums: deposits are summed in fixed point -- two runs of a deck agree bit for bit, as two runs of the reference
  // do (include/vpic_hip.h, vpic_hip_set_accumulation).  VPIC_HIP_DETERMINISTIC=1 / =0 decides; unset, decks that clean div E
  // get them: the Marder pass feeds rhof back into E, and with float-atomic sums the runs of such a deck fall into two groups
  // (tests/test_gpu_deck_host.py) -- the reproducible mode holds the plain deck's tolerance against the reference's run, and
  // since round 4 costs advance_p 1.24 x (integer run sums), not 2 x
  {
    const char *v = getenv("VPIC_HIP_DETERMINISTIC");
    hip_deterministic = v ? atoi(v) != 0 : clean_div_e_interval > 0;
    if (hip_deterministic) CK(vpic_hip_set_accumulation(engine, 1, 0.0));
  }
  for (size_t k = 0; k < species_order.size(); k++) {
    species_t *sp = species_order[k];
    const int id = vpic_hip_species_create(engine, sp->q_m, sp->max_np, sp->max_nm);
    if (id != (int)k) ERROR(("%s", vpic_hip_last_error()));
  }
  for (size_t k = 0; k < reflux_handlers.size(); k++)
    CK(vpic_hip_set_maxwellian_reflux(engine, -(int)k - 3, reflux_handlers[k].ut_para, reflux_handlers[k].ut_perp,
                                      MAX_REFLUX_SPECIES, 0x9e3779b9u * (unsigned)(g_mp_rank + 1)));
  // the transport of the exchanges with other domains (see "exchanges with the neighbouring domains")
  hip_transport = multi() ? chosen_transport() : XPORT_NONE;
  if (hip_transport == XPORT_RCCL) {
    char id[VPIC_HIP_COMM_ID_BYTES];
    memset(id, 0, sizeof(id));
    if (g_mp_rank == 0) CK(vpic_hip_comm_unique_id(id));
    mp_bcast(id, (int)sizeof(id));
    if (vpic_hip_comm_create(&comm, engine, id, g_mp_nproc, g_mp_rank))
      ERROR(("%s\n\t(VPIC_HIP_HOST_TRANSPORT=mpi stages the messages through host memory instead)", vpic_hip_last_error()));
  }
  // which particle exchange: the device-resident one overlapped with the push, unless the deck has custom particle boundary
  // handlers (the engine's resident exchange does not serve them) or asks for the reference's protocol
  const char *xv = getenv("VPIC_HIP_HOST_EXCHANGE");
  hip_resident_exchange = multi() && reflux_handlers.empty() && !(xv && !strcmp(xv, "legacy"));
  CK(vpic_hip_set_sort_order(engine, 1));                  // the order of a sorted species is the engine's business (nothing here reads partition[]); the phased push needs it
  if (multi() && g_mp_rank == 0 && verbose)
    fprintf(stderr, "hip host: %d rank(s)%s, transport: %s, particle exchange: %s\n", g_mp_nproc, self_sends() ? " sending to itself across its periodic axes" : "",
            transport_name(), hip_resident_exchange ? "device-resident, overlapped with the push" : "count-then-payload (boundary_p.c)");
  hip_upload_mirrors();                                   // fields, particles (and a first load_interpolator)
}

// ---- initialize: src/vpic/initialize.cxx:13-100 ----------------------------------------------------
void vpic_simulation::initialize(int argc, char **argv) {
  grid = (grid_t *)calloc(1, sizeof(grid_t));
  for (int k = 0; k < 27; k++) grid->bc[k] = anti_symmetric_fields;
  rng = new mt_rng_t;
  mt_seed(rng, 0);                                        // new_mt_rng(rank)
  user_initialization(argc, argv);
  if (!field_advance) ERROR(("the deck did not call finalize_field_advance"));
  create_engine();
  start_demand_mirrors();
  // consistency checks and derived fields of the user's initial state, initialize.cxx:28-76
  double tmp;
  const bool talk = verbose && g_mp_rank == 0;
  tmp = x_synchronize_tang_e_norm_b();                                            // :32
  if (talk) vpic_host_log("Checking interdomain synchronization: error = %e (arb units)\n", tmp);
  CK(vpic_hip_compute_div_b_err(engine));                                         // :38
  tmp = x_rms(false);                                                             // :39
  if (talk) vpic_host_log("Checking magnetic field divergence: RMS error = %e (charge/volume)\n", tmp);
  x_clean_div_b();                                                                // :44
  x_compute_curl_b();                                                             // :51  radiation damping fields
  x_accumulate_rho();                                                             // :56-59  bound charge density
  x_compute_rhob();                                                               // :60
  x_compute_div_e_err();                                                          // :66
  tmp = x_rms(true);
  if (talk) vpic_host_log("Checking electric field divergence: RMS error = %e (charge/volume)\n", tmp);
  if (tmp > 0) CK(vpic_hip_clean_div_e(engine));                                  // :71
  x_synchronize_tang_e_norm_b();                                                  // :76
  if (!species_order.empty()) CK(vpic_hip_load_interpolator(engine));             // :86
  { species_t *sp;                                        // :88-89 -- species_list only: species a deck took off the list stay as loaded
    LIST_FOR_EACH(sp, species_list) for (size_t k = 0; k < species_order.size(); k++)
      if (species_order[k] == sp) CK(vpic_hip_uncenter_p(engine, (int)k)); }
  if (g_demand) mirrors_stale(); else hip_sync_mirrors();
  user_diagnostics();                                     // initialize.cxx:98
  mirrors_after_user_code();
}

// ---- advance: src/vpic/advance.cxx:13-244 ------------------------------------------------------------
// VPIC_HIP_HOST_TIMING=1: wall-clock split of advance() printed at the end (device work is
// asynchronous, so a phase is charged for the waits it makes; the total is what counts)
static double g_t_step = 0, g_t_mirror = 0, g_t_diag = 0;
static long g_n_step = 0;
static inline double wall_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
void vpic_simulation::finalize(void) {
  if (engine && multi() && vpic_host_mp_rank() == 0 && verbose) {
    int64_t m = 0, b = 0;
    if (comm) vpic_hip_comm_stats(comm, &m, &b);
    fprintf(stderr, "hip host: transport: %s, %ld exchanges posted (%lld messages, %.1f MB over RCCL), %ld host synchronisations for particle exchanges, %ld recovery rounds\n",
            transport_name(), x_messages, (long long)m, b * 1e-6, x_syncs, x_recoveries);
  }
  if (!getenv("VPIC_HIP_HOST_TIMING") || vpic_host_mp_rank() != 0 || !g_n_step) return;
  if (engine) vpic_hip_sync(engine);
  fprintf(stderr, "hip host timing: %ld steps, time step %.3f ms/step, mirror refresh %.3f s, user_diagnostics %.3f s\n",
          g_n_step, 1e3 * g_t_step / g_n_step, g_t_mirror, g_t_diag);
}
int vpic_simulation::advance(void) {
  if (num_step > 0 && step >= num_step) return 0;
  const double t_begin = wall_now();
  CK(vpic_hip_clear_accumulators(engine));                                        // :38
  // only species on species_list: a deck may take species off the list and advance them itself
  // (tracers, decks/trecon-part/tracer.cxx:64-107)
  std::vector<char> listed(species_order.size(), 0);
  sort_pending.resize(species_order.size(), 0);
  { species_t *sp; LIST_FOR_EACH(sp, species_list) for (size_t k = 0; k < species_order.size(); k++) if (species_order[k] == sp) listed[k] = 1; }
  for (size_t k = 0; k < species_order.size(); k++) {                             // :43-51
    const species_t *sp = species_order[k];
    if (!listed[k]) continue;
    int due = sp->sort_interval > 0 && step % sp->sort_interval == 0;
    if (hip_adaptive_sort && sp->sort_interval > 0) CK(vpic_hip_sort_due(engine, (int)k, sp->sort_interval, &due));   // the deck's interval becomes the upper bound
    // a sort that is due is carried out where the species is touched next: by its push (vpic_hip_sort_advance_p: when the push
    // before counted for it, the particles are written straight to their sorted places) or, should a deck hook look at the
    // species first, by the download of its mirror
    if (due) sort_pending[k] = 1;
    // fixed intervals: the push before a sort takes the sort's histogram
    if (!hip_adaptive_sort && sp->sort_interval > 0 && (step + 1) % sp->sort_interval == 0) CK(vpic_hip_species_sort_hint(engine, (int)k));
  }
  mirrors_stale();
  user_particle_collisions();                                                     // :67
  mirrors_after_user_code();
  flush_injected();
  if (multi() && hip_resident_exchange) x_push_and_exchange(listed);               // :70-73 and :94-96, overlapped
  else for (size_t k = 0; k < species_order.size(); k++) if (listed[k]) resident_advance_p((int)k);   // :70-73
  CK(vpic_hip_reduce_accumulators(engine));                                       // :74
  run_emitters();                                                                 // :83-84
  mirrors_stale();
  user_particle_injection();                                                      // :85
  mirrors_after_user_code();
  flush_injected();
  if (multi() && hip_resident_exchange) {                                         // :94-96 happened with the push; what emitters and the
    double pend = movers_pending ? 1 : 0;                                         // injection hook added since (on ANY rank: the exchange is collective)
    mp_allsum_d(&pend, 1);
    if (pend > 0) x_exchange_rounds(num_comm_round);
  } else resident_boundary_p();                                                   // :94-96
  CK(vpic_hip_clear_jf_unload_accumulator(engine));                               // :109-110 in one pass over the mesh
  x_synchronize_jf();                                                             // :112
  mirrors_stale();
  user_current_injection();                                                       // :123
  mirrors_after_user_code();
  CK(vpic_hip_advance_b(engine, 0.5f));                                           // :129
  x_advance_e();                                                                  // :133 (begin/end_remote_ghost_tang_b inside)
  mirrors_stale();
  user_field_injection();                                                         // :141
  mirrors_after_user_code();
  CK(vpic_hip_advance_b(engine, 0.5f));                                           // :147
  const bool talk = verbose && g_mp_rank == 0;
  double err;
  if (clean_div_e_interval > 0 && step % clean_div_e_interval == 0) {               // :151-173
    x_accumulate_rho();
    x_compute_div_e_err();
    err = x_rms(true);
    if (talk) vpic_host_log("Divergence cleaning electric field: initial rms error = %e (charge/volume)\n", err);
    if (err > 0) {
      CK(vpic_hip_clean_div_e(engine));
      x_compute_div_e_err();
      err = x_rms(true);
      if (talk) vpic_host_log("Cleaned rms error = %e (charge/volume)\n", err);
      if (err > 0) CK(vpic_hip_clean_div_e(engine));
    }
  }
  if (clean_div_b_interval > 0 && step % clean_div_b_interval == 0) {               // :177-195
    CK(vpic_hip_compute_div_b_err(engine));
    err = x_rms(false);
    if (talk) vpic_host_log("Divergence cleaning magnetic field: initial rms error = %e (charge/volume)\n", err);
    if (err > 0) {
      x_clean_div_b();
      CK(vpic_hip_compute_div_b_err(engine));
      err = x_rms(false);
      if (talk) vpic_host_log("Cleaned rms error = %e (charge/volume)\n", err);
      if (err > 0) x_clean_div_b();
    }
  }
  if (sync_shared_interval > 0 && step % sync_shared_interval == 0) {               // :199-207
    err = x_synchronize_tang_e_norm_b();
    if (talk) vpic_host_log("Domain desynchronization error = %e (arb units)\n", err);
  }
  CK(vpic_hip_load_interpolator(engine));                                         // :214
  step++;                                                                         // :218
  mirrors_current = false;
  const double t_stepped = wall_now();
  mirrors_stale();
  const double t_mirrored = wall_now();
  user_diagnostics();                                                             // :233
  mirrors_after_user_code();
  g_t_step += t_stepped - t_begin; g_t_mirror += t_mirrored - t_stepped; g_t_diag += wall_now() - t_mirrored; g_n_step++;
  return 1;
}

// ---- dump_energies: src/vpic/dump.cxx:37-77 ----------------------------------------------------------
void vpic_simulation::dump_energies(const char *fname, int append) {
  if (!fname) ERROR(("Invalid file name"));
  double en_f[6];
  CK(vpic_hip_energy_f(engine, en_f));
  mp_allsum_d(en_f, 6);
  std::vector<double> en_p;
  species_t *sp;
  LIST_FOR_EACH(sp, species_list) en_p.push_back(resident_energy_p(sp->p));
  if (!en_p.empty()) mp_allsum_d(&en_p[0], (int)en_p.size());
  if (g_mp_rank != 0) return;
  FILE *f = fopen(fname, append ? "a" : "w");
  if (!f) ERROR(("Could not open \"%s\".", fname));
  if (append == 0) {
    fprintf(f, "%% Layout\n%% step ex ey ez bx by bz");
    LIST_FOR_EACH(sp, species_list) fprintf(f, " \"%s\"", sp->name);
    fprintf(f, "\n%% timestep = %e\n", grid->dt);
  }
  fprintf(f, "%i %e %e %e %e %e %e", step, en_f[0], en_f[1], en_f[2], en_f[3], en_f[4], en_f[5]);
  for (size_t k = 0; k < en_p.size(); k++) fprintf(f, " %e", en_p[k]);
  fprintf(f, "\n");
  fclose(f);
}

// ---- binary dumps: src/vpic/dump.cxx:190-329, header src/vpic/dumpmacros.h:10-48 ---------------------
namespace {
template <class T> void put(FILE *f, T v) { fwrite(&v, sizeof(T), 1, f); }
void write_header_v0(FILE *f, int dump_type, int sp_id, float q_m, int step, const grid_t *g) {
  put<char>(f, (char)CHAR_BIT); put<char>(f, (char)sizeof(short int)); put<char>(f, (char)sizeof(int));
  put<char>(f, (char)sizeof(float)); put<char>(f, (char)sizeof(double));
  put<short int>(f, (short int)0xcafe); put<int>(f, (int)0xdeadbeef); put<float>(f, 1.0f); put<double>(f, 1.0);
  put<int>(f, 0); put<int>(f, dump_type);
  put<int>(f, step); put<int>(f, g->nx); put<int>(f, g->ny); put<int>(f, g->nz);
  put<float>(f, g->dt); put<float>(f, g->dx); put<float>(f, g->dy); put<float>(f, g->dz);
  put<float>(f, g->x0); put<float>(f, g->y0); put<float>(f, g->z0);
  put<float>(f, g->cvac); put<float>(f, g->eps0); put<float>(f, g->damp);
  put<int>(f, vpic_host_mp_rank()); put<int>(f, vpic_host_mp_nproc());
  put<int>(f, sp_id); put<float>(f, q_m);
}
void write_array_header(FILE *f, int elem_size, int ndim, const int *dim) {
  put<int>(f, elem_size); put<int>(f, ndim);
  fwrite(dim, sizeof(int), ndim, f);
}
FILE *open_dump(const char *fbase, int ftag, int step) {
  if (!fbase) ERROR(("Invalid filename"));
  char fname[256];
  if (ftag) snprintf(fname, sizeof(fname), "%s.%i.%i", fbase, step, vpic_host_mp_rank());
  else      snprintf(fname, sizeof(fname), "%s.%i", fbase, vpic_host_mp_rank());
  FILE *f = fopen(fname, "wb");
  if (!f) ERROR(("Could not open \"%s\".", fname));
  return f;
}
}  // namespace

void vpic_simulation::dump_fields(const char *fbase, int ftag) {
  FILE *f = open_dump(fbase, ftag, step);
  if (!g_demand && !mirrors_current) hip_sync_mirrors();   // demand mode: the reads below fault in exactly the arrays they need
  write_header_v0(f, 1 /* dump_type::field_dump */, -1 /* invalid_species_id */, 0, step, grid);
  const int dim[3] = {grid->nx + 2, grid->ny + 2, grid->nz + 2};
  write_array_header(f, (int)sizeof(field_t), 3, dim);
  vpic_host_touch(field, sizeof(field_t));
  fwrite(field, sizeof(field_t), (size_t)dim[0] * dim[1] * dim[2], f);
  fclose(f);
}

void vpic_simulation::dump_hydro(const char *sp_name, const char *fbase, int ftag) {
  species_t *sp = find_species(sp_name);
  if (!sp) ERROR(("Invalid species \"%s\".", sp_name));
  int id = -1;
  for (size_t k = 0; k < species_order.size(); k++) if (species_order[k] == sp) id = (int)k;
  CK(vpic_hip_clear_hydro(engine));                       // dump.cxx:236-238
  CK(vpic_hip_accumulate_hydro_p(engine, id));
  x_synchronize_hydro();
  const int dim[3] = {grid->nx + 2, grid->ny + 2, grid->nz + 2};
  std::vector<vpic_hydro_t> h((size_t)dim[0] * dim[1] * dim[2]);
  CK(vpic_hip_get_hydro(engine, &h[0]));
  FILE *f = open_dump(fbase, ftag, step);
  write_header_v0(f, 2 /* dump_type::hydro_dump */, sp->id, sp->q_m, step, grid);
  write_array_header(f, (int)sizeof(vpic_hydro_t), 3, dim);
  fwrite(&h[0], sizeof(vpic_hydro_t), h.size(), f);
  fclose(f);
}

void vpic_simulation::dump_particles(const char *sp_name, const char *fbase, int ftag) {
  species_t *sp = find_species(sp_name);
  if (!sp) ERROR(("Invalid species name \"%s\".", sp_name));
  if (!g_demand && !mirrors_current) hip_sync_mirrors();   // demand mode: the reads below fault in exactly the arrays they need
  FILE *f = open_dump(fbase, ftag, step);
  write_header_v0(f, 3 /* dump_type::particle_dump */, sp->id, sp->q_m, step, grid);
  const int dim[1] = {sp->np};
  write_array_header(f, (int)sizeof(particle_t), 1, dim);
  // dump.cxx:313-320: a copy of the list is time-centred (center_p) and written, the list itself stays
  std::vector<particle_t> buf(sp->p, sp->p + sp->np);
  if (sp->np) vpic_hip_ref_center_p(&buf[0], sp->np, sp->q_m, interpolator, grid);
  fwrite(buf.data(), sizeof(particle_t), buf.size(), f);
  fclose(f);
}

// ---- dump_species / dump_materials / dump_grid: src/vpic/dump.cxx:82-187 ---------------------------
void vpic_simulation::dump_species(const char *fname) {
  if (vpic_host_mp_rank() != 0) return;
  if (!fname) ERROR(("Invalid file name"));
  FILE *f = fopen(fname, "w");
  if (!f) ERROR(("Could not open \"%s\".", fname));
  species_t *sp;
  LIST_FOR_EACH(sp, species_list) fprintf(f, "%s\n%i\n%e\n", sp->name, sp->id, sp->q_m);
  fclose(f);
}
void vpic_simulation::dump_materials(const char *fname) {
  if (vpic_host_mp_rank() != 0) return;
  if (!fname) ERROR(("Invalid file name"));
  FILE *f = fopen(fname, "w");
  if (!f) ERROR(("Could not open \"%s\".", fname));
  for (size_t k = material_records.size(); k-- > 0;) {      // new_material pushes on the front of the list (material.c)
    const material_rec &m = material_records[k];
    fprintf(f, "%s\n%i\n%e %e %e\n%e %e %e\n%e %e %e\n", m.name.c_str(), (int)k, m.eps[0], m.eps[1], m.eps[2],
            m.mu[0], m.mu[1], m.mu[2], m.sigma[0], m.sigma[1], m.sigma[2]);
  }
  fclose(f);
}
void vpic_simulation::dump_grid(const char *fbase) {
  FILE *f = open_dump(fbase, 0, step);
  write_header_v0(f, 0 /* dump_type::grid_dump */, -1, 0, step, grid);
  int dim[4] = {3, 3, 3, 0};
  write_array_header(f, (int)sizeof(grid->bc[0]), 3, dim);
  fwrite(grid->bc, sizeof(grid->bc[0]), 27, f);
  // range / neighbor in the reference's global numbering (size_grid / join_grid, src/grid/ops.c:52-97,
  // 135-182): every slab has the same number of voxels, ids of rank r start at r * nv
  const int np = vpic_host_mp_nproc(), me = vpic_host_mp_rank();
  const int nx = grid->nx, ny = grid->ny, nz = grid->nz;
  const int64_t sy = nx + 2, sz = sy * (ny + 2), nv = sz * (nz + 2);
  std::vector<int64_t> range((size_t)np + 1), nb(grid->neighbor, grid->neighbor + 6 * nv);
  for (int r = 0; r <= np; r++) range[r] = r * nv;
  if (np > 1) {
    for (size_t k = 0; k < nb.size(); k++) if (nb[k] >= 0) nb[k] += range[me];
    const int n3[3] = {nx, ny, nz};
    for (int f6 = 0; f6 < 6; f6++) {
      if (face_rank[f6] < 0) continue;
      const int a = f6 % 3, l = f6 < 3 ? 1 : n3[a], r = f6 < 3 ? n3[a] : 1;      // this side's boundary plane, the neighbour's
      int lo[3] = {1, 1, 1}, hi[3] = {nx, ny, nz};
      lo[a] = hi[a] = l;
      for (int z = lo[2]; z <= hi[2]; z++) for (int y = lo[1]; y <= hi[1]; y++) for (int x = lo[0]; x <= hi[0]; x++) {
        int c[3] = {x, y, z};
        c[a] = r;
        nb[6 * (x + sy * y + sz * z) + f6] = range[face_rank[f6]] + (c[0] + sy * c[1] + sz * c[2]);
      }
    }
  }
  dim[0] = np + 1;
  write_array_header(f, (int)sizeof(int64_t), 1, dim);
  fwrite(&range[0], sizeof(int64_t), range.size(), f);
  dim[0] = 6; dim[1] = nx + 2; dim[2] = ny + 2; dim[3] = nz + 2;
  write_array_header(f, (int)sizeof(int64_t), 4, dim);
  fwrite(&nb[0], sizeof(int64_t), nb.size(), f);
  fclose(f);
}

// ---- global_header / field_dump / hydro_dump: src/vpic/dump.cxx:899-1552 ---------------------------
int vpic_simulation::dump_mkdir(const char *dname) { return mkdir(dname, S_IRWXU); }      // FileUtils::makeDirectory
int vpic_simulation::dump_cwd(char *dname, size_t size) { return getcwd(dname, size) ? 0 : -1; }

namespace {
struct var_info { const char *name, *degree, *elements, *type; int size; };
// what the .vpc file says about each group of variables (part of the file format, dump.cxx:899-927)
const var_info field_info[12] = {
  {"Electric Field", "VECTOR", "3", "FLOATING_POINT", 4}, {"Electric Field Divergence Error", "SCALAR", "1", "FLOATING_POINT", 4},
  {"Magnetic Field", "VECTOR", "3", "FLOATING_POINT", 4}, {"Magnetic Field Divergence Error", "SCALAR", "1", "FLOATING_POINT", 4},
  {"TCA Field", "VECTOR", "3", "FLOATING_POINT", 4}, {"Bound Charge Density", "SCALAR", "1", "FLOATING_POINT", 4},
  {"Free Current Field", "VECTOR", "3", "FLOATING_POINT", 4}, {"Charge Density", "SCALAR", "1", "FLOATING_POINT", 4},
  {"Edge Material", "VECTOR", "3", "INTEGER", 2}, {"Node Material", "SCALAR", "1", "INTEGER", 2},
  {"Face Material", "VECTOR", "3", "INTEGER", 2}, {"Cell Material", "SCALAR", "1", "INTEGER", 2}};
const var_info hydro_info[5] = {
  {"Current Density", "VECTOR", "3", "FLOATING_POINT", 4}, {"Charge Density", "SCALAR", "1", "FLOATING_POINT", 4},
  {"Momentum Density", "VECTOR", "3", "FLOATING_POINT", 4}, {"Kinetic Energy Density", "SCALAR", "1", "FLOATING_POINT", 4},
  {"Stress Tensor", "TENSOR", "6", "FLOATING_POINT", 4}};
const size_t field_group_bit[12] = {0, 3, 4, 7, 8, 11, 12, 15, 16, 19, 20, 23}, hydro_group_bit[5] = {0, 3, 4, 7, 8};
void hashed(FILE *f, const char *comment) {
  static const char bar[] = "################################################################################\n";
  fprintf(f, "%s# %s\n%s", bar, comment, bar);
}
void item(FILE *f, const char *comment, const char *fmt, double a, double b = 0) {
  hashed(f, comment);
  fprintf(f, fmt, a, b);
}
void variables(FILE *f, const char *key, const DumpParameters &dp, const var_info *info, const size_t *bit, size_t groups) {
  std::vector<size_t> sel;
  for (size_t v = 0; v < groups; v++) if (dp.output_vars.bitset(bit[v])) sel.push_back(v);
  fprintf(f, "%s %d\n", key, (int)sel.size());
  for (size_t k = 0; k < sel.size(); k++) {
    const var_info &i = info[sel[k]];
    fprintf(f, "\"%s\" %s %s %s %d\n", i.name, i.degree, i.elements, i.type, i.size);
  }
}
}  // namespace

void vpic_simulation::global_header(const char *base, std::vector<DumpParameters *> dumpParams) {
  if (vpic_host_mp_rank() != 0) return;
  if (dumpParams.size() < 2) ERROR(("global_header needs the field parameters and at least one species"));
  char filename[256];
  snprintf(filename, sizeof(filename), "%s.vpc", base);
  FILE *f = fopen(filename, "w");
  if (!f) ERROR(("Failed opening file: %s", filename));
  hashed(f, "Header version information");
  fprintf(f, "VPIC_HEADER_VERSION 1.0.0\n\n");
  hashed(f, "Header size for data file headers in bytes");
  fprintf(f, "DATA_HEADER_SIZE 123\n\n");
  item(f, "Time step increment", "GRID_DELTA_T %f\n\n", grid->dt);
  item(f, "GRID_CVAC", "GRID_CVAC %f\n\n", grid->cvac);
  item(f, "GRID_EPS0", "GRID_EPS0 %f\n\n", grid->eps0);
  item(f, "Grid extents in the x-dimension", "GRID_EXTENTS_X %f %f\n\n", grid->x0, grid->x1);
  item(f, "Grid extents in the y-dimension", "GRID_EXTENTS_Y %f %f\n\n", grid->y0, grid->y1);
  item(f, "Grid extents in the z-dimension", "GRID_EXTENTS_Z %f %f\n\n", grid->z0, grid->z1);
  item(f, "Spatial step increment in x-dimension", "GRID_DELTA_X %f\n\n", grid->dx);
  item(f, "Spatial step increment in y-dimension", "GRID_DELTA_Y %f\n\n", grid->dy);
  item(f, "Spatial step increment in z-dimension", "GRID_DELTA_Z %f\n\n", grid->dz);
  hashed(f, "Domain partitions in x-dimension"); fprintf(f, "GRID_TOPOLOGY_X %d\n\n", (int)px);
  hashed(f, "Domain partitions in y-dimension"); fprintf(f, "GRID_TOPOLOGY_Y %d\n\n", (int)py);
  hashed(f, "Domain partitions in z-dimension"); fprintf(f, "GRID_TOPOLOGY_Z %d\n\n", (int)pz);
  hashed(f, "Field data information");
  fprintf(f, "FIELD_DATA_DIRECTORY %s\nFIELD_DATA_BASE_FILENAME %s\n", dumpParams[0]->baseDir, dumpParams[0]->baseFileName);
  variables(f, "FIELD_DATA_VARIABLES", *dumpParams[0], field_info, field_group_bit, 12);
  fprintf(f, "\n");
  hashed(f, "Number of species with output data");
  fprintf(f, "NUM_OUTPUT_SPECIES %d\n\n", (int)dumpParams.size() - 1);
  for (size_t i = 1; i < dumpParams.size(); i++) {
    char comment[128];
    snprintf(comment, sizeof(comment), "Species(%d) data information", (int)i);
    hashed(f, comment);
    fprintf(f, "SPECIES_DATA_DIRECTORY %s\nSPECIES_DATA_BASE_FILENAME %s\n", dumpParams[i]->baseDir, dumpParams[i]->baseFileName);
    variables(f, "HYDRO_DATA_VARIABLES", *dumpParams[i], hydro_info, hydro_group_bit, 5);
    if (i < dumpParams.size() - 1) fprintf(f, "\n");
  }
  fclose(f);
}

// header + array header + payload; the payload comes gathered from the device (vpic_hip_dump_gather)
void vpic_simulation::banded_dump(int what, int dump_type, int sp_id, float q_m, DumpParameters &dp) {
  const int rank = vpic_host_mp_rank();
  char name[512];
  snprintf(name, sizeof(name), "%s/T.%d", dp.baseDir, step);
  dump_mkdir(name);
  snprintf(name, sizeof(name), "%s/T.%d/%s.%d.%d", dp.baseDir, step, dp.baseFileName, step, rank);
  FILE *f = fopen(name, "wb");
  if (!f) ERROR(("Failed opening file: %s", name));
  const size_t s[3] = {dp.stride_x, dp.stride_y, dp.stride_z};
  const int n[3] = {grid->nx, grid->ny, grid->nz};
  static const char *axis = "xyz";
  for (int a = 0; a < 3; a++)
    if (s[a] < 1 || n[a] % (int)s[a]) ERROR(("%c stride must be an integer factor of n%c", axis[a], axis[a]));
  grid_t out = *grid;                                     // what WRITE_HEADER_V0 reports: the strided mesh
  out.nx = n[0] / (int)s[0]; out.ny = n[1] / (int)s[1]; out.nz = n[2] / (int)s[2];
  out.dx = grid->dx * s[0]; out.dy = grid->dy * s[1]; out.dz = grid->dz * s[2];
  write_header_v0(f, dump_type, sp_id, q_m, step, &out);
  const bool hydro_inner = what == VPIC_HIP_DUMP_HYDRO && dp.format != band;     // dump.cxx:1511-1515
  const int extra = hydro_inner ? 0 : 2, dim[3] = {out.nx + extra, out.ny + extra, out.nz + extra};
  const int W = what == VPIC_HIP_DUMP_FIELDS ? 20 : 16;
  write_array_header(f, 4 * W, 3, dim);
  std::vector<int32_t> words;
  const size_t limit = what == VPIC_HIP_DUMP_FIELDS ? total_field_variables : total_hydro_variables;
  for (size_t v = 0; v < limit; v++) if (dp.output_vars.bitset(v)) words.push_back((int32_t)v);
  const int layout = dp.format == band ? VPIC_HIP_DUMP_BAND : hydro_inner ? VPIC_HIP_DUMP_INTERLEAVE_INNER : VPIC_HIP_DUMP_INTERLEAVE;
  const size_t count = (size_t)dim[0] * dim[1] * dim[2] * (layout == VPIC_HIP_DUMP_BAND ? words.size() : (size_t)W);
  std::vector<uint32_t> payload(count);
  if (count) CK(vpic_hip_dump_gather(engine, what, layout, words.empty() ? NULL : &words[0], (int)words.size(),
                                     (int)s[0], (int)s[1], (int)s[2], &payload[0], count * 4));
  fwrite(payload.data(), 4, count, f);
  fclose(f);
}
void vpic_simulation::field_dump(DumpParameters &dumpParams) {
  banded_dump(VPIC_HIP_DUMP_FIELDS, 1 /* dump_type::field_dump */, -1, 0, dumpParams);
}
void vpic_simulation::hydro_dump(const char *speciesname, DumpParameters &dumpParams) {
  species_t *sp = find_species(speciesname);
  if (!sp) ERROR(("Invalide species name: %s", speciesname));
  int id = -1;
  for (size_t k = 0; k < species_order.size(); k++) if (species_order[k] == sp) id = (int)k;
  CK(vpic_hip_clear_hydro(engine));                       // dump.cxx:1403-1405
  CK(vpic_hip_accumulate_hydro_p(engine, id));
  x_synchronize_hydro();
  banded_dump(VPIC_HIP_DUMP_HYDRO, 2 /* dump_type::hydro_dump */, sp->id, sp->q_m, dumpParams);
}

void vpic_simulation::create_field_list(char *strlist, DumpParameters &dp) {      // dump.cxx:929-947
  strlist[0] = 0;
  for (size_t i = 0, any = 0; i < 12; i++) if (dp.output_vars.bitset(field_group_bit[i])) {
    if (any) strcat(strlist, ", ");
    strcat(strlist, field_info[i].name); any = 1;
  }
}
void vpic_simulation::create_hydro_list(char *strlist, DumpParameters &dp) {      // dump.cxx:949-965
  strlist[0] = 0;
  for (size_t i = 0, any = 0; i < 5; i++) if (dp.output_vars.bitset(hydro_group_bit[i])) {
    if (any) strcat(strlist, ", ");
    strcat(strlist, hydro_info[i].name); any = 1;
  }
}
// ---- dump_restart / restart: src/vpic/dump.cxx:332-851 -----------------------------------------------
// Everything a run needs to go on from the current step, one file per rank, named as the reference
// names it (fbase.rank, or fbase.step.rank with a tag).  The V0 header is the reference's
// (dump_type::restart_dump); what follows is this host's own layout -- a restart file is only ever
// read by the executable that wrote it (the reference's holds its function pointers).  Like the
// reference's, it saves the species that are on species_list and the raw bytes of the deck's globals:
// pointers a deck keeps there (tracer lists, DumpParameters vectors) are the deck's to rebuild, as
// decks/trecon-part/tracer.cxx does with its own tracer restart file.
namespace {
const char restart_magic[8] = {'V', 'H', 'I', 'P', 'R', 'S', 'T', '1'};
template <class T> void get(FILE *f, T &v) { if (fread(&v, sizeof(T), 1, f) != 1) ERROR(("restart file is truncated")); }
void put_string(FILE *f, const char *s) { const int n = (int)strlen(s); put<int>(f, n); fwrite(s, 1, n, f); }
std::string get_string(FILE *f) { int n; get(f, n); std::string s((size_t)n, ' '); if (n && fread(&s[0], 1, n, f) != (size_t)n) ERROR(("restart file is truncated")); return s; }
}  // namespace

void vpic_simulation::dump_restart(const char *fbase, int fname_tag) {
  if (vpic_host_mp_rank() == 0) MESSAGE(("Dumping restart to \"%s\"", fbase));
  if (!g_demand && !mirrors_current) hip_sync_mirrors();   // demand mode: the reads below fault in exactly the arrays they need
  FILE *f = open_dump(fbase, fname_tag, step);
  write_header_v0(f, 4 /* dump_type::restart_dump */, -1, 0, step, grid);
  fwrite(restart_magic, 1, 8, f);
  put<int>(f, num_step); put<int>(f, status_interval); put<int>(f, clean_div_e_interval); put<int>(f, clean_div_b_interval);
  put<int>(f, sync_shared_interval); put<double>(f, quota); put<int>(f, restart_interval); put<int>(f, hydro_interval);
  put<int>(f, field_interval); put<int>(f, particle_interval); put<int>(f, num_comm_round); put<int>(f, verbose);
  put<int>(f, hip_mirror_interval); put<int>(f, hip_adaptive_sort);
  put<int>(f, (int)px); put<int>(f, (int)py); put<int>(f, (int)pz);
  fwrite(rng, sizeof(mt_rng_t), 1, f);
  put<int>(f, (int)material_records.size());
  for (size_t k = 0; k < material_records.size(); k++) {
    put_string(f, material_records[k].name.c_str());
    fwrite(material_records[k].eps, sizeof(float), 3, f); fwrite(material_records[k].mu, sizeof(float), 3, f);
    fwrite(material_records[k].sigma, sizeof(float), 3, f);
  }
  const size_t nv = (size_t)(grid->nx + 2) * (grid->ny + 2) * (grid->nz + 2);
  fwrite(grid, sizeof(grid_t), 1, f);                     // the scalars and bc[]; the pointers are rebuilt on the way in
  fwrite(grid->neighbor, sizeof(int64_t), 6 * nv, f);
  fwrite(face_rank, sizeof(int), 6, f); fwrite(topo_index, sizeof(int), 3, f); fwrite(topo_size, sizeof(int), 3, f);
  vpic_host_touch(field, sizeof(field_t));
  fwrite(field, sizeof(field_t), nv, f);
  std::vector<species_t *> listed;
  { species_t *sp; LIST_FOR_EACH(sp, species_list) listed.push_back(sp); }
  put<int>(f, (int)listed.size());
  for (size_t k = listed.size(); k-- > 0;) {              // oldest first, so that reading rebuilds the list in order
    const species_t *sp = listed[k];
    put_string(f, sp->name);
    put<int>(f, sp->id); put<int>(f, sp->max_np); put<int>(f, sp->max_nm); put<float>(f, sp->q_m);
    put<int>(f, sp->sort_interval); put<int>(f, sp->sort_out_of_place); put<int>(f, sp->np);
    vpic_host_touch(sp->p, sizeof(particle_t));
    fwrite(sp->p, sizeof(particle_t), (size_t)sp->np, f);
  }
  fwrite(user_global, 1, sizeof(user_global), f);
  put<int>(f, (int)reflux_handlers.size());
  if (!reflux_handlers.empty()) fwrite(&reflux_handlers[0], sizeof(maxwellian_reflux_t), reflux_handlers.size(), f);
  std::vector<emitter_t *> ems;
  for (emitter_t *em = emitter_list; em; em = em->next) ems.push_back(em);
  put<int>(f, (int)ems.size());
  for (size_t k = ems.size(); k-- > 0;) {                 // oldest first
    const emitter_t *em = ems[k];
    put_string(f, em->name); put_string(f, em->sp->name);
    put<int>(f, em->emission_model == child_langmuir ? 0 : em->emission_model == ccube ? 1 : 2);
    put<int>(f, em->n_component);
    fwrite(em->component, sizeof(int), (size_t)em->n_component, f);
    fwrite(em->model_parameters, 1, MAX_EMISSION_MODEL_SIZE, f);
  }
  fclose(f);
}

void vpic_simulation::restart(const char *fbase) {
  char fname[512];
  snprintf(fname, sizeof(fname), "%s.%i", fbase, vpic_host_mp_rank());
  FILE *f = fopen(fname, "rb");
  if (!f) ERROR(("Could not open \"%s\".", fname));
  if (vpic_host_mp_rank() == 0) MESSAGE(("Restarting from \"%s\"", fbase));
  char head[5 + 2 + 4 + 4 + 8];
  int version, type, saved_step, n[3], rank_nproc[2], sp_id;
  float fl[10], q_m;
  if (fread(head, 1, sizeof(head), f) != sizeof(head)) ERROR(("restart file is truncated"));
  get(f, version); get(f, type); get(f, saved_step); get(f, n[0]); get(f, n[1]); get(f, n[2]);
  if (fread(fl, sizeof(float), 10, f) != 10) ERROR(("restart file is truncated"));
  get(f, rank_nproc[0]); get(f, rank_nproc[1]); get(f, sp_id); get(f, q_m);
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, restart_magic, 8) != 0 || type != 4)
    ERROR(("\"%s\" is not a restart file of this host", fname));
  if (rank_nproc[0] != vpic_host_mp_rank() || rank_nproc[1] != vpic_host_mp_nproc())
    ERROR(("restart file was written by rank %i of %i", rank_nproc[0], rank_nproc[1]));
  step = saved_step;
  int ipx, ipy, ipz;
  get(f, num_step); get(f, status_interval); get(f, clean_div_e_interval); get(f, clean_div_b_interval);
  get(f, sync_shared_interval); get(f, quota); get(f, restart_interval); get(f, hydro_interval);
  get(f, field_interval); get(f, particle_interval); get(f, num_comm_round); get(f, verbose);
  get(f, hip_mirror_interval); get(f, hip_adaptive_sort); get(f, ipx); get(f, ipy); get(f, ipz);
  px = ipx; py = ipy; pz = ipz;
  rng = new mt_rng_t;
  get(f, *rng);
  int nmat;
  get(f, nmat);
  for (int k = 0; k < nmat; k++) {
    const std::string name = get_string(f);
    float v[9];
    if (fread(v, sizeof(float), 9, f) != 9) ERROR(("restart file is truncated"));
    define_material(name.c_str(), v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
  }
  grid = (grid_t *)calloc(1, sizeof(grid_t));
  get(f, *grid);
  const size_t nv = (size_t)(grid->nx + 2) * (grid->ny + 2) * (grid->nz + 2);
  grid->mp = NULL; grid->boundary = NULL; grid->nb = 0;
  grid->range = (int64_t *)malloc(2 * sizeof(int64_t));
  grid->range[0] = 0; grid->range[1] = (int64_t)nv;
  grid->neighbor = (int64_t *)malloc(6 * nv * sizeof(int64_t));
  if (fread(grid->neighbor, sizeof(int64_t), 6 * nv, f) != 6 * nv) ERROR(("restart file is truncated"));
  if (fread(face_rank, sizeof(int), 6, f) != 6 || fread(topo_index, sizeof(int), 3, f) != 3 || fread(topo_size, sizeof(int), 3, f) != 3)
    ERROR(("restart file is truncated"));
  finalize_field_advance(standard_field_advance);
  if (fread(field, sizeof(field_t), nv, f) != nv) ERROR(("restart file is truncated"));
  int nsp;
  get(f, nsp);
  for (int k = 0; k < nsp; k++) {
    const std::string name = get_string(f);
    int id, max_np, max_nm, sort_interval, sort_out_of_place, np;
    float sq_m;
    get(f, id); get(f, max_np); get(f, max_nm); get(f, sq_m); get(f, sort_interval); get(f, sort_out_of_place); get(f, np);
    species_t *sp = define_species(name.c_str(), sq_m, max_np, max_nm, sort_interval, sort_out_of_place);
    sp->id = id; sp->np = np;
    if (np && fread(sp->p, sizeof(particle_t), (size_t)np, f) != (size_t)np) ERROR(("restart file is truncated"));
  }
  if (fread(user_global, 1, sizeof(user_global), f) != sizeof(user_global)) ERROR(("restart file is truncated"));
  int nreflux;
  get(f, nreflux);
  reflux_handlers.resize((size_t)nreflux);
  if (nreflux && fread(&reflux_handlers[0], sizeof(maxwellian_reflux_t), (size_t)nreflux, f) != (size_t)nreflux) ERROR(("restart file is truncated"));
  grid->nb = nreflux;
  int nem;
  get(f, nem);
  for (int k = 0; k < nem; k++) {
    const std::string name = get_string(f), sp_name = get_string(f);
    int model, ncomp;
    get(f, model); get(f, ncomp);
    emitter_t *em = new_emitter(name.c_str(), find_species(sp_name.c_str()), model == 0 ? child_langmuir : model == 1 ? ccube : ivory, ncomp, &emitter_list);
    if (!em || !em->sp) ERROR(("restart file names an emitter this host cannot rebuild"));
    em->n_component = ncomp;
    if (fread(em->component, sizeof(int), (size_t)ncomp, f) != (size_t)ncomp || fread(em->model_parameters, 1, MAX_EMISSION_MODEL_SIZE, f) != MAX_EMISSION_MODEL_SIZE)
      ERROR(("restart file is truncated"));
  }
  fclose(f);
  create_engine();
  mirrors_current = true;
  start_demand_mirrors();
}

// ---- emitters: src/emitter/emitter.c:5-73 ------------------------------------------------------------------
void child_langmuir(void) {}
void ccube(void) {}
void ivory(void) {}
emitter_t *find_emitter_name(const char *name, emitter_t *e_list) {
  for (emitter_t *e = e_list; name && e; e = e->next) if (strcmp(e->name, name) == 0) return e;
  return NULL;
}
emitter_t *new_emitter(const char *name, species_t *sp, emission_model_t emission_model, int max_component, emitter_t **e_list) {
  if (!e_list) ERROR(("Invalid emitter list."));
  if (!name || !name[0]) ERROR(("Cannot create a nameless emitter."));
  if (find_emitter_name(name, *e_list)) ERROR(("There is already a emitter named \"%s\".", name));
  if (emission_model != child_langmuir && emission_model != ccube && emission_model != ivory)
    ERROR(("emitter \"%s\": only the child_langmuir, ccube and ivory emission models are supported by this host", name));
  if (max_component < 1) return NULL;                       // an emitter without components on this rank (emitter.c:24-29)
  emitter_t *e = (emitter_t *)calloc(1, sizeof(emitter_t) + strlen(name));
  e->sp = sp; e->emission_model = emission_model;
  e->component = (int *)calloc((size_t)max_component, sizeof(int));
  e->max_component = max_component;
  strcpy(e->name, name);
  e->next = *e_list;
  *e_list = e;
  return e;
}
void vpic_simulation::run_emitters(void) {                  // advance.cxx:83-84
  for (emitter_t *em = emitter_list; em; em = em->next) {
    int id = -1;
    for (size_t k = 0; k < species_order.size(); k++) if (species_order[k] == em->sp) id = (int)k;
    if (id < 0) ERROR(("emitter \"%s\" emits a species this simulation does not hold", em->name));
    const ccube_t *a = (const ccube_t *)em->model_parameters;     // the three parameter blocks begin alike
    const float coef = em->emission_model == child_langmuir ? (float)(32. / 81.) : em->emission_model == ccube ? 1.f : (float)(1. / 6.);
    const float thresh = em->emission_model == child_langmuir ? 0.f : a->thresh_e_norm;
    if (a->n_emit_per_face < 1 || em->n_component < 1) continue;
    CK(vpic_hip_emit(engine, id, em->component, em->n_component, a->n_emit_per_face, a->ut_perp, a->ut_para, coef, thresh,
                     0x2545f491u * (unsigned)(g_mp_rank + 1)));
    movers_pending = true;
    mirrors_current = false;
  }
}

// ---- custom particle boundaries: add_boundary (src/grid/add_boundary.c:9-35) ------------------------------
void maxwellian_reflux(void) {}
int vpic_host_add_boundary(grid_t *g, boundary_handler_t handler, const void *params, int size) {
  vpic_simulation *sim = vpic_host_current;
  if (!g || !sim || g != sim->grid || !handler || !params) ERROR(("Add boundary encountered invalid boundary!!!"));
  if (handler != (boundary_handler_t)maxwellian_reflux || size != (int)sizeof(maxwellian_reflux_t))
    ERROR(("only the maxwellian_reflux particle boundary handler is supported by this host"));
  sim->reflux_handlers.push_back(*(const maxwellian_reflux_t *)params);
  g->nb = (int)sim->reflux_handlers.size();
  return -((int)sim->reflux_handlers.size() - 1) - 3;
}

// ---- L3 entry points for deck code ------------------------------------------------------------------
species_t *new_species(const char *name, float q_m, int max_local_np, int max_local_nm, int sort_interval,
                       int sort_out_of_place, species_t **sp_list) {               // species_advance.c:21-63
  if (!name || !sp_list || max_local_np < 1 || max_local_nm < 1) ERROR(("Bad species arguments"));
  if (find_species_name(name, *sp_list)) ERROR(("There is already a species named \"%s\".", name));
  species_t *sp = (species_t *)calloc(1, sizeof(species_t) + strlen(name));
  strcpy(sp->name, name);
  sp->id = *sp_list ? (*sp_list)->id + 1 : 0;
  sp->max_np = max_local_np; sp->max_nm = max_local_nm;
  sp->p = (particle_t *)calloc((size_t)max_local_np, sizeof(particle_t));
  sp->pm = (particle_mover_t *)calloc((size_t)max_local_nm, sizeof(particle_mover_t));
  sp->q_m = q_m; sp->sort_interval = sort_interval; sp->sort_out_of_place = sort_out_of_place;
  sp->next = *sp_list;
  *sp_list = sp;
  return sp;
}
species_t *find_species_id(species_id id, species_t *sp_list) {
  species_t *sp;
  LIST_FOR_EACH(sp, sp_list) if (sp->id == id) return sp;
  return NULL;
}
species_t *find_species_name(const char *name, species_t *sp_list) {
  species_t *sp;
  if (!name) return NULL;
  LIST_FOR_EACH(sp, sp_list) if (strcmp(sp->name, name) == 0) return sp;
  return NULL;
}
accumulator_t *new_accumulators(const grid_t *g) {                                // sf_interface.c:56-75
  if (!g) ERROR(("Bad grid."));
  const size_t nv = (size_t)(g->nx + 2) * (g->ny + 2) * (g->nz + 2);
  accumulator_t *a = (accumulator_t *)calloc(nv + 1, sizeof(accumulator_t));
  if (!a) ERROR(("Failed to allocate accumulator."));
  return a;
}
int advance_p(particle_t *p0, int np, const float q_m, particle_mover_t *pm, int max_nm, accumulator_t *a0,
              const interpolator_t *f0, const grid_t *g) {
  vpic_simulation *sim = vpic_host_current;
  const int id = sim ? sim->resident_id(p0) : -1;
  if (id < 0) return vpic_hip_ref_advance_p(p0, np, q_m, pm, max_nm, a0, f0, g);
  sim->resident_advance_p(id);
  return 0;                                               // the movers stay on the device until boundary_p
}
void boundary_p(species_t *sp_list, field_t *f, accumulator_t *a0, const grid_t *g, mt_rng_t *rng) {
  (void)rng;
  vpic_simulation *sim = vpic_host_current;
  species_t *sp;
  bool resident = sim != NULL && sp_list != NULL;
  LIST_FOR_EACH(sp, sp_list) if (!sim || sim->resident_id(sp->p) < 0) resident = false;
  if (resident) { sim->resident_boundary_p(); return; }   // collective, like the reference's: every rank makes the same calls
  vpic_hip_ref_boundary_p(sp_list, f, a0, g, NULL);
}
void sort_p(species_t *sp, const grid_t *g) {
  vpic_simulation *sim = vpic_host_current;
  const int id = sim && sp ? sim->resident_id(sp->p) : -1;
  if (id < 0) { vpic_hip_ref_sort_p(sp, g); return; }
  if (vpic_hip_sort_p(sim->resident_engine(), id)) ERROR(("%s", vpic_hip_last_error()));
}

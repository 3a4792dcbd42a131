// vpic_hip_host.hxx -- a host for the reference's INPUT DECK API on top of the resident HIP engine.
//
// An input deck of the reference is C++ that is compiled INTO member functions of class
// vpic_simulation (src/deck_wrapper.cxx:16-36,541) and uses that class's members and helpers as
// free names (src/vpic/vpic.hxx:126-555).  This header provides a class of the same name with the
// members and helpers such decks use, so that a deck source file compiles unchanged against it:
//
//   hipcc/g++ -DINPUT_DECK=<deck.cxx> host/main.cxx host/deck_wrapper.cxx host/vpic_hip_host.cxx \
//       -Iinclude -Iold-vpic_amd/host -Lold-vpic_amd -lvpic_hip
//
// The time step (src/vpic/advance.cxx:13-244) runs on the GPU; the host arrays the deck can see
// (field, interpolator, species->p) are MIRRORS of the device state, kept coherent ON DEMAND: while the
// engine works they are inaccessible pages; the first access from a deck hook brings that array over,
// a first write marks it dirty, and when the hook returns dirty arrays go back to the device
// (vpic_hip_host.cxx, "host mirrors on demand").  A deck therefore reads and edits fields and particles
// in user_diagnostics / user_field_injection / user_current_injection / user_particle_injection /
// user_particle_collisions exactly as it does in the reference, and pays only for what it touches;
// species->np is current in every hook.  The library announces every host range a HIP copy is about to touch
// (vpic_hip_set_host_access_hook), so a twin handed one of these arrays finds it resident.
//
// Scope: box decks -- periodic, conducting/reflecting or absorbing faces (define_periodic_grid /
// define_reflecting_grid / define_absorbing_grid, set_domain_*_bc) -- on one rank or, built with
// -DVPIC_HIP_HOST_MPI, cut into gpx x gpy x gpz equal bricks over MPI ranks (one GPU each); any number of species, also
// species a deck takes off species_list and advances itself (tracers); any number of materials
// (anisotropic eps / mu / sigma, set_region_material); set_region_field; divergence cleaning; every
// dump of the reference (energies, fields, hydro, particles, grid, species, materials, the strided
// field_dump / hydro_dump with their .vpc header) and restart files (dump_restart, `restart <fbase>`).
// Particles a deck injects while the run is under way (inject_particle / inject_particle_raw from
// user_particle_injection) reach the device at the end of that call.  maxwellian_reflux boundaries (add_boundary); surface emitters with the child_langmuir / ccube / ivory laws.
// Not there: other custom particle boundary handlers, set_region_bc.  Unsupported calls stop with the reference's ERROR convention (message, exit(1)).
// uniform_rand() is the reference's generator (MT19937 + its 53-bit open-interval conversion,
// src/util/mtrand/mtrand.c:69-76,240, mtrand_conv.h:61); maxwellian_rand() uses Box-Muller on it
// instead of the reference's 256-layer ziggurat (whose tables are a data file of the reference), so
// decks that draw normals load statistically equivalent, not identical, particles.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>
#include <map>

#include "vpic_hip.h"
#include "vpic_hip_dropin.h"
#include "FileIO.hxx"

// ---- the reference's type names ---------------------------------------------------------------
typedef vpic_particle_t particle_t;
typedef vpic_particle_mover_t particle_mover_t;
typedef vpic_particle_injector_t particle_injector_t;
typedef vpic_interpolator_t interpolator_t;
typedef vpic_accumulator_t accumulator_t;
typedef vpic_field_t field_t;
typedef vpic_material_coefficient_t material_coefficient_t;
typedef vpic_grid_t grid_t;
typedef vpic_species_t species_t;
typedef uint16_t material_id;
enum { invalid_material_id = 65535 };
typedef int32_t species_id;

enum { anti_symmetric_fields = -1, pec_fields = -1, metal_fields = -1, symmetric_fields = -2,
       pmc_fields = -3, absorb_fields = -4, reflect_particles = -1, absorb_particles = -2 };
#define BOUNDARY(i, j, k) (((i) + 1) + 3 * (((j) + 1) + 3 * ((k) + 1)))
#define INDEX_FORTRAN_3(x, y, z, xl, xh, yl, yh, zl, zh) \
  ((x) - (xl) + ((xh) - (xl) + 1) * ((y) - (yl) + ((yh) - (yl) + 1) * ((z) - (zl))))
#define LIST_FOR_EACH(n, list) for ((n) = (list); (n) != NULL; (n) = (n)->next)

#define ERROR(args) do { fprintf(stderr, "Error at %s(%i):\n\t", __FILE__, __LINE__); vpic_host_log args; fprintf(stderr, "\n"); exit(1); } while (0)
#define WARNING(args) do { fprintf(stderr, "Warning at %s(%i):\n\t", __FILE__, __LINE__); vpic_host_log args; fprintf(stderr, "\n"); } while (0)
#define MESSAGE(args) do { fprintf(stderr, "%s(%i): ", __FILE__, __LINE__); vpic_host_log args; fprintf(stderr, "\n"); } while (0)
void vpic_host_log(const char *fmt, ...);
// demand-mode mirrors: make a block safe to hand to the kernel (write / fwrite) -- see vpic_hip_host.cxx
void vpic_host_touch(const void *p, size_t bytes);
inline size_t vpic_host_fwrite(const void *p, size_t size, size_t n, FILE *f) { vpic_host_touch(p, size * n); return fwrite(p, size, n, f); }
#define BEGIN_PRIMITIVE do
#define END_PRIMITIVE while (0)

// src/field_advance/field_advance.h:185-302: the slots a deck can reach through
// field_advance->method (hot slots are the HIP twins; the rest is not on the path yet)
struct field_advance_methods_t {
  void (*advance_b)(field_t *, const grid_t *, float);
  void (*advance_e)(field_t *, const material_coefficient_t *, const grid_t *);
  void (*energy_f)(double *, const field_t *, const material_coefficient_t *, const grid_t *);
  void (*clear_jf)(field_t *, const grid_t *);
  void (*synchronize_jf)(field_t *, const grid_t *);
};
extern field_advance_methods_t standard_field_advance[1];
struct field_advance_t {
  field_t *f;
  material_coefficient_t *m;
  grid_t *g;
  field_advance_methods_t method[1];
};

// energy_p with the reference's signature (src/species_advance/standard/spa.h:101-106).  When the
// arrays are the simulation's own mirrors it is answered from the resident state (no transfer).
double energy_p(const particle_t *p0, int np, float q_m, const interpolator_t *f0, const grid_t *g);

struct mt_rng_t { uint32_t state[624]; int next; };

// ---- field_dump / hydro_dump vocabulary of the decks (src/vpic/vpic.hxx:44-124, src/util/BitField.hxx) ----
// bit k of a field mask selects 32-bit word k of field_t, of a hydro mask word k of hydro_t
const uint32_t all = 0xffffffff;
const uint32_t electric = 7u << 0, div_e_err = 1u << 3, magnetic = 7u << 4, div_b_err = 1u << 7, tca = 7u << 8,
               rhob = 1u << 11, current = 7u << 12, rhof = 1u << 15, emat = 7u << 16, nmat = 1u << 19,
               fmat = 7u << 20, cmat = 1u << 23;
const uint32_t current_density = 7u << 0, charge_density = 1u << 3, momentum_density = 7u << 4, ke_density = 1u << 7,
               stress_tensor = 63u << 8;
const size_t total_field_variables = 24, total_field_groups = 12, total_hydro_variables = 14, total_hydro_groups = 5;

class BitField {
  uint32_t bits_;
public:
  BitField(uint32_t setbits = 0xffffffff) : bits_(setbits) {}
  uint32_t set(uint32_t mask) { return bits_ |= mask; }
  uint32_t clear(uint32_t mask) { return bits_ &= ~mask; }
  uint32_t setbit(size_t bit) { return bits_ |= (uint32_t)1 << bit; }
  uint32_t clearbit(size_t bit) { return bits_ &= ~((uint32_t)1 << bit); }
  bool bitset(size_t bit) const { return (bits_ >> bit) & 1u; }
  bool bitclear(size_t bit) const { return !bitset(bit); }
  size_t bitsum(const size_t *indeces, size_t size) const { size_t n = 0; for (size_t i = 0; i < size; i++) n += bitset(indeces[i]); return n; }
  size_t bitsum() const { size_t n = 0; for (size_t i = 0; i < 32; i++) n += bitset(i); return n; }
};
enum DumpFormat { band = 0, band_interleave = 1 };
struct DumpParameters {
  void output_variables(uint32_t mask) { output_vars.set(mask); }   // ORs into the default all-ones mask, as the reference does
  BitField output_vars;
  size_t stride_x, stride_y, stride_z;
  DumpFormat format;
  char name[128], baseDir[128], baseFileName[128];
};

// ---- the L3 entry points decks call directly (tracer decks: decks/trecon-part/tracer.cxx:76-107) ----
// Same signatures as the reference (species_advance.h:103-122, spa.h:23-66, sf_interface.h:112-114).
// When the arrays belong to a species of the running simulation the work is done on the resident
// engine (no PCIe traffic): advance_p launches that species' push and returns 0 (the movers stay on
// the device); boundary_p runs the engine's particle exchange for every resident species that has
// movers pending (a no-op when none has: the deck's repeated calls cost nothing); sort_p sorts on
// the device.  For any other arrays the drop-in twins of vpic_hip_dropin.h run (arrays over PCIe).
species_t *new_species(const char *name, float q_m, int max_local_np, int max_local_nm, int sort_interval,
                       int sort_out_of_place, species_t **sp_list);
species_t *find_species_id(species_id id, species_t *sp_list);
species_t *find_species_name(const char *name, species_t *sp_list);
accumulator_t *new_accumulators(const grid_t *g);
int advance_p(particle_t *p0, int np, const float q_m, particle_mover_t *pm, int max_nm, accumulator_t *a0,
              const interpolator_t *f0, const grid_t *g);
void boundary_p(species_t *sp_list, field_t *f, accumulator_t *a0, const grid_t *g, mt_rng_t *rng);
void sort_p(species_t *sp, const grid_t *g);

// Custom particle boundary handlers (src/boundary/boundary.h, src/grid/grid.h:170-190): a deck fills a
// maxwellian_reflux_t, registers it with add_boundary( grid, maxwellian_reflux, &params ) and gives the code it
// gets back (<= -3) to set_domain_particle_bc.  maxwellian_reflux here is only the handler's NAME: the work is
// done on the device (vpic_hip_set_maxwellian_reflux).  Other handlers are refused.
typedef struct maxwellian_reflux { float ut_perp[32], ut_para[32]; } maxwellian_reflux_t;
typedef void (*boundary_handler_t)(void);
void maxwellian_reflux(void);
int vpic_host_add_boundary(grid_t *g, boundary_handler_t handler, const void *params, int size);
#define add_boundary(g, bh, ip) vpic_host_add_boundary((g), (boundary_handler_t)(bh), (ip), (int)sizeof(*(ip)))

// Surface emitters (src/emitter/emitter.h, emitter.c:5-73; deck macros define_surface_emitter / define_volume_emitter):
// an emitter is a list of (voxel, face) components, a species and an emission model with its parameters.  The
// three models of the reference -- child_langmuir, ccube, ivory -- are NAMES here; vpic_simulation::advance
// runs them on the device (vpic_hip_emit) where the reference calls them (advance.cxx:83-84).
#define COMPONENT_ID(local_cell, component_type) (((local_cell) << 5) | (component_type))
#define EXTRACT_LOCAL_CELL(component_id) ((component_id) >> 5)
#define EXTRACT_COMPONENT_TYPE(component_id) ((component_id) & 31)
#define MAX_EMISSION_MODEL_SIZE 1024
typedef void (*emission_model_t)(void);
typedef struct emitter {
  int *component;
  int n_component, max_component;
  species_t *sp;
  emission_model_t emission_model;
  char model_parameters[MAX_EMISSION_MODEL_SIZE];
  struct emitter *next;
  char name[1];
} emitter_t;
typedef struct child_langmuir { int n_emit_per_face; float ut_perp, ut_para; } child_langmuir_t;
typedef struct ccube { int n_emit_per_face; float ut_perp, ut_para, thresh_e_norm; } ccube_t;
typedef struct ivory { int n_emit_per_face; float ut_perp, ut_para, thresh_e_norm; } ivory_t;
void child_langmuir(void);
void ccube(void);
void ivory(void);
emitter_t *new_emitter(const char *name, species_t *sp, emission_model_t emission_model, int max_component, emitter_t **e_list);
emitter_t *find_emitter_name(const char *name, emitter_t *e_list);

// the mp_* calls decks make on grid->mp (src/util/mp/mp.h): elapsed wall clock (max over ranks),
// barrier, finalize, the blocking int send / receive of the turnstile macros
double mp_elapsed(void *mp);
void mp_barrier(void *mp);
void mp_finalize(void *mp);
void mp_send_i(int *buf, int n, int dst, void *mp);
void mp_recv_i(int *buf, int n, int src, void *mp);

// message passing between domains (vpic_hip_host.cxx): with -DVPIC_HIP_HOST_MPI one MPI rank per
// domain / GPU, brick decompositions; without it a single domain
void vpic_host_mp_init(int *argc, char ***argv);
void vpic_host_mp_finalize(void);
int vpic_host_mp_rank(void);
int vpic_host_mp_nproc(void);

class vpic_simulation {
public:
  vpic_simulation();
  ~vpic_simulation();
  void initialize(int argc, char **argv);
  void restart(const char *fbase);          // continue from the files dump_restart wrote (main: `deck.exe restart <fbase>`)
  int advance(void);
  void finalize(void);
  void sync(void) { if (engine) vpic_hip_sync(engine); }   // wait for the device (main: before the wall clock is read)
  inline double rank(void) { return vpic_host_mp_rank(); }
  inline double nproc(void) { return vpic_host_mp_nproc(); }

  // ---- what decks use as free names (src/vpic/vpic.hxx:151-555) ------------------------------
  int verbose, step, num_step, num_comm_round, status_interval;
  int clean_div_e_interval, clean_div_b_interval, sync_shared_interval;
  double quota;
  int restart_interval, hydro_interval, field_interval, particle_interval;
  mt_rng_t *rng;
  grid_t *grid;
  species_t *species_list;
  emitter_t *emitter_list;
  field_advance_t *field_advance;
  field_t *field;
  interpolator_t *interpolator;
  accumulator_t *accumulator;
  char user_global[16384];

  // HIP-specific knobs a deck may set (defaults can also come from the environment:
  // VPIC_HIP_MIRROR_INTERVAL, VPIC_HIP_ADAPTIVE_SORT)
  int hip_mirror_interval;      // kept for restart files of earlier versions (unused)
  int hip_adaptive_sort;        // 1 (default): the engine decides when a species is sorted (vpic_hip_sort_due), the deck's
                                // sort_interval is the upper bound; 0: exactly every sort_interval steps
  void hip_sync_mirrors(void);  // refresh them now
  void hip_upload_mirrors(void);// push host-side edits of field / particles back to the device
  void mirror_download(int kind, int sp);   // demand mode (VPIC_HIP_MIRROR=demand), see vpic_hip_host.cxx

  void define_periodic_grid(double xl, double yl, double zl, double xh, double yh, double zh,
                            double gnx, double gny, double gnz, double gpx, double gpy, double gpz);
  void define_reflecting_grid(double xl, double yl, double zl, double xh, double yh, double zh,
                              double gnx, double gny, double gnz, double gpx, double gpy, double gpz);
  void define_absorbing_grid(double xl, double yl, double zl, double xh, double yh, double zh,
                             double gnx, double gny, double gnz, double gpx, double gpy, double gpz, int pbc);
  void set_domain_field_bc(int boundary, int fbc);
  void set_domain_particle_bc(int boundary, int pbc);
  material_id define_material(const char *name, double eps, double mu = 1, double sigma = 0, double zeta = 0);
  material_id define_material(const char *name, double epsx, double epsy, double epsz, double mux, double muy, double muz,
                              double sigmax, double sigmay, double sigmaz, double zetax = 0, double zetay = 0, double zetaz = 0);
  material_id lookup_material(const char *name);
  void finalize_field_advance(field_advance_methods_t *fam = standard_field_advance);
  species_t *define_species(const char *name, double q_m, double max_local_np, double max_local_nm,
                            double sort_interval, double sort_out_of_place);
  species_t *find_species(const char *name);
  void inject_particle(species_t *sp, double x, double y, double z, double ux, double uy, double uz,
                       double q, int64_t tag, double age = 0, int update_rhob = 1);
  void inject_particle_raw(species_t *sp, float dx, float dy, float dz, int32_t i, float ux, float uy, float uz, float q);
  void seed_rand(double seed);
  double uniform_rand(double low, double high);
  double maxwellian_rand(double dev);
  void dump_energies(const char *fname, int append = 1);
  // binary dumps with the V0 header (dump.cxx:190-329, dumpmacros.h:10-48): same bytes as the
  // reference writes for the same state; particles are time-centred by center_p on the way out
  void dump_fields(const char *fbase, int ftag = 1);
  void dump_hydro(const char *sp_name, const char *fbase, int ftag = 1);
  void dump_particles(const char *sp_name, const char *fbase, int ftag = 1);
  // text / grid dumps (dump.cxx:82-187) and the strided, banded dumps read by the reference's
  // visualisation tools (dump.cxx:929-1552); the field / hydro payloads are gathered on the device
  void dump_species(const char *fname);
  void dump_materials(const char *fname);
  void dump_grid(const char *fbase);
  int dump_mkdir(const char *dname);
  int dump_cwd(char *dname, size_t size);
  void global_header(const char *base, std::vector<DumpParameters *> dumpParams);
  void create_field_list(char *strlist, DumpParameters &dumpParams);
  void create_hydro_list(char *strlist, DumpParameters &dumpParams);
  void dump_restart(const char *fbase, int fname_tag = 1);
  void field_dump(DumpParameters &dumpParams);
  void hydro_dump(const char *speciesname, DumpParameters &dumpParams);
  size_t px, py, pz;            // domain topology (vpic.hxx:171)
  inline double courant_length(double lx, double ly, double lz, double nx, double ny, double nz) {
    double w0, w1 = 0;
    if (nx > 1) w0 = nx / lx, w1 += w0 * w0;
    if (ny > 1) w0 = ny / ly, w1 += w0 * w0;
    if (nz > 1) w0 = nz / lz, w1 += w0 * w0;
    return sqrt(1 / w1);
  }
  inline double trunc_granular(double a, double b) { return b * int(a / b); }

  // resident-engine answers for diagnostics on the simulation's own arrays
  bool owns(const particle_t *p0) const;
  int resident_id(const particle_t *p0) const;      // engine species id of a resident particle array, or -1
  vpic_hip_engine_t *resident_engine(void) { return engine; }
  void resident_advance_p(int id);                  // push one resident species; its movers wait for ...
  void resident_boundary_p(void);                   // ... the exchange of every resident species with pending movers
  double resident_energy_p(const particle_t *p0);
  bool resident_energy_f(double *en, const field_t *f);

private:
  vpic_hip_engine_t *engine;
  std::vector<species_t *> species_order;    // engine species id = position
  std::vector<vpic_material_coefficient_t> materials;
  struct material_rec { std::string name; float eps[3], mu[3], sigma[3]; };
  std::vector<material_rec> material_records;
  void banded_dump(int what, int dump_type, int sp_id, float q_m, DumpParameters &dumpParams);
  bool mirrors_current;
  bool movers_pending;          // a push has run since the last particle exchange
  std::vector<char> sort_pending;   // species whose sort is due and not yet carried out (advance(): done by the push or by the first look at the species)
  void pending_sort(int id);
  std::vector<std::vector<particle_t> > injected;   // particles a deck injects while the run is under way, per species
  std::vector<particle_injector_t> injected_aged; std::vector<int64_t> injected_aged_tags;   // ... with an age (misc.cxx:93-103)
  std::vector<particle_t> injected_rhob;            // ... those whose charge, negated, goes to rhob (update_rhob)
  void flush_injected(void);
  void run_emitters(void);
public:
  std::vector<maxwellian_reflux_t> reflux_handlers;   // handler k answers to particle code -(k+3)
private:
  void queue_injected(species_t *sp, const particle_t &p);
  void box(double xl, double yl, double zl, double xh, double yh, double zh, int nx, int ny, int nz, int pbc, int fbc);
  void slab(double gx0, double gy0, double gz0, double gx1, double gy1, double gz1, int gnx, int gny, int gnz,
            int gpx, int gpy, int gpz, int pbc, int fbc, bool periodic);
  // exchanges with the neighbouring domains (vpic_hip_host.cxx); face_rank[f] = rank sharing face f (this rank itself when
  // it sends to itself across a periodic axis), or -1
  int face_rank[6];
  int hip_transport;                        // XPORT_*: none, MPI with host staging, RCCL on device buffers
  bool hip_deterministic;                   // fixed-point accumulation (VPIC_HIP_DETERMINISTIC; the default of decks that clean div E)
  bool hip_resident_exchange;               // particle exchange: device-resident and overlapped with the push / the reference's protocol
  vpic_hip_comm_t *comm;
  struct XBuf { void *dev; size_t bytes; std::vector<char> host; XBuf() : dev(NULL), bytes(0) {} };
  std::map<int, XBuf> xbufs;                // device message buffers (and their host twins for the staged transport) by (kind, direction, tag)
  void *xbuf(int kind, int d, int tag, size_t bytes);
  std::vector<char> &xhost_of(void *dev);
  struct XferSet { const void *s[6]; size_t ns[6]; void *r[6]; size_t nr[6]; };   // by travel direction; 0 bytes: no message
  int x_start(const XferSet &x);
  void x_finish(int token);
  template <class Pack, class Unpack> void plane_exchange(int axis, size_t bytes, Pack pack, Unpack unpack, int tag = 50);
  struct XRound { int token, tag; void *ms[6], *mr[6]; int cs[6], cr[6]; };
  struct XHeader { int kind, tag, d, count, wanted; };
  std::map<int, int> x_caps;                // message capacities (injectors) by (kind, direction, species or -1)
  std::vector<double> x_np_max;
  int x_mover_cap, x_flags; long x_messages, x_syncs, x_recoveries;
  int &x_cap(int kind, int d, int k);
  XRound x_round(int tag, const int cs[6], const int cr[6], int mover_cap, uint32_t species_mask);
  void x_land(const XRound &r);
  void x_read_back(const std::vector<XRound> &log, std::vector<XHeader> &H);
  void x_recover(std::vector<XHeader> &H);
  void x_make_room(void);
  void x_exchange_rounds(int rounds);
  void x_push_and_exchange(const std::vector<char> &listed);
  void x_advance_e(void);
public:
  const char *transport_name(void) const;
private:
  bool shared(int axis) const;              // the faces of this axis (one or both) belong to other ranks
  bool multi(void) const;
  int topo_index[3], topo_size[3];          // this rank's brick in the gpx x gpy x gpz decomposition
  void x_boundary_p(void);
  void x_tang_b(void);
  void x_synchronize_jf(void);
  void x_synchronize_rho(void);
  void x_synchronize_hydro(void);
  double x_message(int kind);
  double x_message(int kind, int axis);
  double x_synchronize_tang_e_norm_b(void);
  double x_rms(bool e_field);
  void x_accumulate_rho(void);
  void x_compute_div_e_err(void);
  void x_compute_rhob(void);
  void x_clean_div_b(void);
  void x_compute_curl_b(void);
  void describe(vpic_hip_grid_t &d);
  void create_engine(void);
  void start_demand_mirrors(void);
  void mirrors_stale(void);
  void mirrors_after_user_code(void);

  // the deck's bodies (src/deck_wrapper.cxx:16-36)
  void user_initialization(int argc, char **argv);
  void user_particle_injection(void);
  void user_current_injection(void);
  void user_field_injection(void);
  void user_diagnostics(void);
  void user_particle_collisions(void);
};

extern vpic_simulation *vpic_host_current;   // the one simulation of the process

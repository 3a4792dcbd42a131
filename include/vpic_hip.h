/* vpic_hip.h -- C ABI of the MI355X-native VPIC inner-loop engine (libvpic_hip.so).
 *
 * Plain C: pointers and sizes only.  Two groups of entry points:
 *
 *  (1) RESIDENT ENGINE (vpic_hip_*): the production path.  All per-step state lives in HBM in
 *      the engine's own layout (cell-sorted SoA particles, SoA Yee fields, AoS interpolator and
 *      accumulator arrays); host arrays in the reference's AoS layouts are mirrors that are
 *      copied in/out on demand at the seams where the reference lets user code look at them
 *      (src/vpic/advance.cxx:67,83-85,123,141,233).  One engine = one rectangular domain = one
 *      GPU.  Each function names the reference function it replaces.
 *
 *  (2) DROP-IN KERNELS (vpic_hip_ref_*, see vpic_hip_dropin.h): the reference's own L3 C
 *      signatures over host AoS arrays (src/species_advance/standard/spa.h:23-123,
 *      src/sf_interface/sf_interface.h:83-163, field_advance_methods_t in
 *      src/field_advance/field_advance.h:185-302), implemented by upload -> HIP kernel ->
 *      download.  Parity tests call through these so they read like calls of the reference.
 *
 * Return convention of group (1): 0 on success, non-zero on error with a message available from
 * vpic_hip_last_error().  Nothing here ever falls back to a CPU implementation: if no HIP device
 * is usable every entry point fails.  Group (2) follows the reference's convention instead
 * (src/util/util_base.h:213-219: message on stderr, exit(1)).
 */
#ifndef VPIC_HIP_H
#define VPIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define VPIC_HIP_STATIC_ASSERT(c, msg) static_assert(c, msg)
extern "C" {
#else
#define VPIC_HIP_STATIC_ASSERT(c, msg) _Static_assert(c, msg)
#endif

/* ---- the reference's host-visible struct layouts (byte for byte) ---------------------------- */

/* src/species_advance/species_advance.h:28-34 */
typedef struct vpic_particle {
  float dx, dy, dz;      /* cell-relative position on [-1,1] */
  int32_t i;             /* voxel, FORTRAN index over (0:nx+1,0:ny+1,0:nz+1) */
  float ux, uy, uz;      /* normalised momentum */
  float q;               /* macro-particle charge */
  int64_t tag, tag2;     /* never touched by the kernels */
} vpic_particle_t;
/* src/species_advance/species_advance.h:39-42 */
typedef struct vpic_particle_mover { float dispx, dispy, dispz; int32_t i; } vpic_particle_mover_t;
/* src/species_advance/species_advance.h:48-55 */
typedef struct vpic_particle_injector {
  float dx, dy, dz; int32_t i; float ux, uy, uz, q; float dispx, dispy, dispz; int32_t sp_id;
} vpic_particle_injector_t;
/* src/sf_interface/sf_interface.h:45-58 */
typedef struct vpic_interpolator {
  float ex, dexdy, dexdz, d2exdydz;
  float ey, deydz, deydx, d2eydzdx;
  float ez, dezdx, dezdy, d2ezdxdy;
  float cbx, dcbxdx, cby, dcbydy, cbz, dcbzdz;
  float _pad[2];
} vpic_interpolator_t;
/* src/sf_interface/sf_interface.h:68-77 */
typedef struct vpic_accumulator { float jx[4], jy[4], jz[4]; } vpic_accumulator_t;
/* src/sf_interface/sf_interface.h:28-38 */
typedef struct vpic_hydro {
  float jx, jy, jz, rho;     /* <q v f>, <q f> */
  float px, py, pz, ke;      /* <p f>, <m c^2 (gamma-1) f> */
  float txx, tyy, tzz;       /* <p_i v_i f> */
  float tyz, tzx, txy;       /* <p_i v_j f> */
  float _pad[2];
} vpic_hydro_t;
/* src/field_advance/field_advance.h:159-171 */
typedef struct vpic_field {
  float ex, ey, ez, div_e_err;
  float cbx, cby, cbz, div_b_err;
  float tcax, tcay, tcaz, rhob;
  float jfx, jfy, jfz, rhof;
  uint16_t ematx, ematy, ematz, nmat;
  uint16_t fmatx, fmaty, fmatz, cmat;
} vpic_field_t;
/* src/field_advance/standard/sfa_private.h:24-32 */
typedef struct vpic_material_coefficient {
  float decayx, drivex, decayy, drivey, decayz, drivez;
  float rmux, rmuy, rmuz, nonconductive, epsx, epsy, epsz, pad[3];
} vpic_material_coefficient_t;

VPIC_HIP_STATIC_ASSERT(sizeof(vpic_particle_t) == 48 && offsetof(vpic_particle_t, i) == 12 &&
                       offsetof(vpic_particle_t, ux) == 16 && offsetof(vpic_particle_t, q) == 28 &&
                       offsetof(vpic_particle_t, tag) == 32 && offsetof(vpic_particle_t, tag2) == 40,
                       "particle_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_particle_mover_t) == 16, "particle_mover_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_particle_injector_t) == 48, "particle_injector_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_interpolator_t) == 80 && offsetof(vpic_interpolator_t, ey) == 16 &&
                       offsetof(vpic_interpolator_t, ez) == 32 && offsetof(vpic_interpolator_t, cbx) == 48 &&
                       offsetof(vpic_interpolator_t, cby) == 56 && offsetof(vpic_interpolator_t, cbz) == 64,
                       "interpolator_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_accumulator_t) == 48, "accumulator_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_hydro_t) == 64 && offsetof(vpic_hydro_t, txx) == 32, "hydro_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_field_t) == 80 && offsetof(vpic_field_t, cbx) == 16 &&
                       offsetof(vpic_field_t, tcax) == 32 && offsetof(vpic_field_t, jfx) == 48 &&
                       offsetof(vpic_field_t, rhof) == 60 && offsetof(vpic_field_t, ematx) == 64 &&
                       offsetof(vpic_field_t, fmatx) == 72 && offsetof(vpic_field_t, cmat) == 78,
                       "field_t layout");
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_material_coefficient_t) == 64, "material_coefficient layout");

/* Boundary codes: src/grid/grid.h:56-69 */
enum { VPIC_PEC_FIELDS = -1, VPIC_SYMMETRIC_FIELDS = -2, VPIC_PMC_FIELDS = -3, VPIC_ABSORB_FIELDS = -4 };
enum { VPIC_REFLECT_PARTICLES = -1, VPIC_ABSORB_PARTICLES = -2 };

/* ---- (1) resident engine ------------------------------------------------------------------- */

/* One rectangular domain.  Replaces what the kernels read from grid_t (src/grid/grid.h:112-167).
 * The 6*nv neighbor[] table is generated from one code per face (order -x,-y,-z,+x,+y,+z, as in
 * neighbor[6v+f], src/grid/ops.c:74-97), exactly what size_grid + join_grid + set_fbc + set_pbc
 * build for box decks:
 *   fbc[f] >= 0: fields on this face are shared with domain fbc[f] (== rank: periodic onto itself)
 *   fbc[f] <  0: local field boundary code (VPIC_PEC_FIELDS ...)
 *   pbc[f] >= 0: particles crossing go to domain pbc[f] (== rank: wrap locally)
 *   pbc[f] <  0: VPIC_REFLECT_PARTICLES / VPIC_ABSORB_PARTICLES                              */
typedef struct vpic_hip_grid {
  float dt, cvac, eps0, damp;
  float dx, dy, dz, rdx, rdy, rdz;
  int32_t nx, ny, nz;
  int32_t fbc[6], pbc[6];
  int32_t rank;
} vpic_hip_grid_t;

typedef struct vpic_hip_engine vpic_hip_engine_t;

const char *vpic_hip_last_error(void);
/* A caller whose host arrays are demand-paged (protected while the engine owns the data) registers a function that
 * makes [p, p+bytes) resident -- and writable when for_write -- : every entry point given a host array calls it before a
 * HIP copy touches the range (a copy engine that meets a protected page faults the GPU instead of raising SIGSEGV). */
typedef void (*vpic_hip_host_access_fn)(const void *p, size_t bytes, int for_write);
void vpic_hip_set_host_access_hook(vpic_hip_host_access_fn fn);
int  vpic_hip_device_count(void);
/* device < 0: use the current HIP device */
int  vpic_hip_create(vpic_hip_engine_t **e, const vpic_hip_grid_t *g, int device);
void vpic_hip_destroy(vpic_hip_engine_t *e);
int  vpic_hip_sync(vpic_hip_engine_t *e);
void *vpic_hip_stream(vpic_hip_engine_t *e);                 /* hipStream_t the kernels run on */
int  vpic_hip_nv(const vpic_hip_engine_t *e);                /* (nx+2)(ny+2)(nz+2) */

/* host AoS mirrors <-> device (new_field / new_interpolator / new_accumulators arrays of the
 * reference, src/sf_interface/sf_interface.c:13-83, src/field_advance/standard/sfa.c:56-73) */
int vpic_hip_set_fields(vpic_hip_engine_t *e, const vpic_field_t *f);            /* nv entries */
int vpic_hip_get_fields(vpic_hip_engine_t *e, vpic_field_t *f);
int vpic_hip_set_interpolator(vpic_hip_engine_t *e, const vpic_interpolator_t *fi);
int vpic_hip_get_interpolator(vpic_hip_engine_t *e, vpic_interpolator_t *fi);
int vpic_hip_set_accumulator(vpic_hip_engine_t *e, const vpic_accumulator_t *a);
int vpic_hip_get_accumulator(vpic_hip_engine_t *e, vpic_accumulator_t *a);
/* new_material_coefficients (src/field_advance/standard/sfa.c:80-177): table of n materials */
int vpic_hip_set_material_coefficients(vpic_hip_engine_t *e, const vpic_material_coefficient_t *m, int n);

/* species (new_species, src/species_advance/species_advance.c:21-63).  Returns the id >= 0. */
int vpic_hip_species_create(vpic_hip_engine_t *e, float q_m, int64_t max_np, int64_t max_nm);
int vpic_hip_species_set_particles(vpic_hip_engine_t *e, int sp, const vpic_particle_t *p, int64_t np);
/* One emission step of a surface emitter (src/emitter/child-langmuir.c:43-97, ccube.c, ivory.c; advance.cxx:83-84):
 * components[k] = (local voxel << 5) | BOUNDARY code of the emitting face (emitter.h:12-16).  A face whose normal
 * field pulls the species out and reaches thresh_e_norm emits n_emit_per_face particles sharing the charge
 * eps0 dY dZ dt sqrt(coef |q_m E^3| / dX) -- coef 32/81 (child_langmuir), 1 (ccube), 1/6 (ivory) -- with the
 * models' momentum and age distributions; their charge, negated, goes to rhob; they are moved for the rest of
 * the step.  Random numbers from the device's counter-based stream (seed, call). */
int vpic_hip_emit(vpic_hip_engine_t *e, int sp, const int32_t *components, int n, int n_emit_per_face,
                  float ut_perp, float ut_para, float coef, float thresh_e_norm, uint32_t seed);
/* inject_particle with an age (src/vpic/misc.cxx:93-103): n injector records in HOST memory -- position, voxel,
 * momentum, charge, the displacement still to be travelled, species -- are appended to their species and moved
 * (deposits go to the accumulator, a particle stopped by a face becomes a mover); tags: 2 per record or NULL */
int vpic_hip_inject_aged(vpic_hip_engine_t *e, const vpic_particle_injector_t *inj, const int64_t *tags, int n);
/* accumulate_rhob (boundary_p.c:9-71) for n particles handed over in host memory: q_scale * q of each is spread
 * over the 8 nodes of its cell and added to rhob (inject_particle with update_rhob passes -1, misc.cxx:87-91) */
int vpic_hip_accumulate_rhob(vpic_hip_engine_t *e, const vpic_particle_t *p, int64_t n, float q_scale);
/* A custom particle boundary handler of the maxwellian_reflux kind (src/boundary/maxwellian_reflux.c:47-175) for
 * the faces whose particle code is `code` (<= -3: what add_boundary returns, grid.h:68-69): a particle that hits
 * such a face comes back at once with a momentum drawn from the flux of a Maxwellian at the wall -- normal
 * component sqrt(2) ut_para sqrt(-log U), tangential ones ut_perp N(0,1), per species -- and moves on for the
 * rest of its step.  Random numbers come from a counter-based generator on the device (seed, call, species,
 * mover): statistically the reference's handler, not its stream.  A face whose code has no parameters absorbs
 * (boundary_p.c:312-316). */
int vpic_hip_set_maxwellian_reflux(vpic_hip_engine_t *e, int code, const float *ut_para, const float *ut_perp, int n_species, uint32_t seed);
/* Parity tests: the three numbers the handler draws per refluxed particle (mt_frand, mt_frandn, mt_frandn:
 * maxwellian_reflux.c:120-122) from a table indexed by the particle's position in its species' array (3 floats each)
 * instead of the engine's counter-based generator; n_particles = 0 switches back. */
int vpic_hip_set_reflux_draws(vpic_hip_engine_t *e, const float *draws, int64_t n_particles);
/* The same for vpic_hip_emit: six numbers per emitted-particle slot (slot = component index * n_emit_per_face + k), in
 * the model's draw order (child-langmuir.c:60-75: mt_drand_c, mt_drand_c, mt_drandn x 3, mt_drand_c0). */
int vpic_hip_set_emit_draws(vpic_hip_engine_t *e, const double *draws, int64_t n_slots);
/* n more particles at the end of the list (particles a deck injects while the run is under way,
 * vpic.hxx:463-486); pending movers keep their particle indices */
int vpic_hip_species_append_particles(vpic_hip_engine_t *e, int sp, const vpic_particle_t *p, int64_t n);
int vpic_hip_species_get_particles(vpic_hip_engine_t *e, int sp, vpic_particle_t *p, int64_t cap);
int vpic_hip_species_get_particles_range(vpic_hip_engine_t *e, int sp, vpic_particle_t *p, int64_t from, int64_t count);   /* particles [from, from+count) */
/* How advance_p covers a species of np particles when its workgroups take 256*iters particles each: launches of at most
 * 2^30 particles (32-bit byte offsets inside a launch) that start on workgroup boundaries.  Pure host arithmetic;
 * returns the number of segments (< 0: bad arguments). */
int vpic_hip_push_plan(int64_t np, int iters, int64_t *start, int32_t *count, uint32_t *grid, int max_segments);
/* Synthetic loader for benchmark-sized species: ppc particles in every interior cell, uniform in
 * the cell, drifting Maxwellian momenta -- what a deck's `repeat(N) inject_particle(...)` loop does
 * (src/vpic/vpic.hxx:491-505), on the device with a counter-based generator. */
int vpic_hip_species_load_maxwellian(vpic_hip_engine_t *e, int sp, int ppc, uint32_t seed, float q,
                                     float ux, float uy, float uz, float vth);
int64_t vpic_hip_species_np(vpic_hip_engine_t *e, int sp);               /* live particles */
/* extent = array slots in use (live particles + the dead slots the device-resident exchange leaves until the next sort),
 * and the capacities; _reserve enlarges them between steps (boundary_p.c:416-448 grows by 1.3125 when it runs out) */
int vpic_hip_species_capacity(vpic_hip_engine_t *e, int sp, int64_t *extent, int64_t *max_np, int64_t *max_nm);
int vpic_hip_species_reserve(vpic_hip_engine_t *e, int sp, int64_t max_np, int64_t max_nm);
int64_t vpic_hip_species_nm(vpic_hip_engine_t *e, int sp);
/* movers left by the last advance_p, ascending in particle index (src/species_advance/standard/
 * boundary_p.c:168-176 requires that order) */
int vpic_hip_species_get_movers(vpic_hip_engine_t *e, int sp, vpic_particle_mover_t *pm, int64_t cap);
/* hand a mover list to the engine (what a caller of boundary_p holds in sp->pm / sp->nm) */
int vpic_hip_species_set_movers(vpic_hip_engine_t *e, int sp, const vpic_particle_mover_t *pm, int64_t nm);
/* partition[nv+1] as sort_p leaves it (src/species_advance/standard/sort_p.c:48-58) */
int vpic_hip_species_get_partition(vpic_hip_engine_t *e, int sp, int32_t *partition);
/* the same for the engine's own order (vpic_hip_set_sort_order(e, 1), by 4x4x4-cell tile, cell by cell within a tile): where
 * every cell of every tile began at the species' last sort, tpart[64 * tiles + 1] with tiles = ceil(nx/4) ceil(ny/4) ceil(nz/4),
 * key = 64 * tile + (z & 3) << 4 | (y & 3) << 2 | (x & 3) of the cell (x - 1, y - 1, z - 1).  *count: entries written (call
 * with tpart = NULL to learn it).  Valid while vpic_hip_species_sort_order says 2; a species sorted by tile only has its
 * tiles' first entries right and nothing to say about the cells inside. */
int vpic_hip_species_get_tile_partition(vpic_hip_engine_t *e, int sp, int32_t *tpart, int64_t *count);

/* kernels, named after the reference functions they replace */
int vpic_hip_load_interpolator(vpic_hip_engine_t *e);       /* sf_interface/load_interpolator.cxx:284-369 */
int vpic_hip_clear_accumulators(vpic_hip_engine_t *e);      /* sf_interface/clear_accumulators.c:26-49 */
int vpic_hip_reduce_accumulators(vpic_hip_engine_t *e);     /* sf_interface/reduce_accumulators.cxx:143-165: one accumulator here, nothing to reduce */
int vpic_hip_unload_accumulator(vpic_hip_engine_t *e);      /* sf_interface/unload_accumulator.cxx:81-121 */
/* Arithmetic of advance_p.  EXACT (default): the reference's scalar pipeline, operation for operation
 * (advance_p.cxx:68-177 compiled without contraction, true divides and sqrtf): every particle bit-identical.
 * FAST: contracted multiply-adds and 1-ulp v_rsq_f32 / v_rcp_f32 instead of the correctly rounded sequences --
 * what the reference's own V4 pipelines do with rsqrt / rcp estimates (src/util/v4/v4_sse.hxx:914-939, selected
 * by its shipped machine configs); momenta within 8 ulp of the scalar pipeline per step. */
#define VPIC_HIP_PUSH_EXACT 0
#define VPIC_HIP_PUSH_FAST  1
int vpic_hip_set_push_mode(vpic_hip_engine_t *e, int mode);
/* How deposits are summed.  0 (default): float sums (LDS and global float atomics): accumulators agree with the reference's
 * to 2e-6 of the largest entry and differ from run to run at the 1e-7 level (the order of the additions is the machine's).
 * 1: DETERMINISTIC -- every deposit is rounded to 64-bit fixed point (scale 2^k chosen from q_ref, the |charge| of a typical
 * macro-particle; <= 0: the largest charge the host has put into a species so far) and summed as an integer in LDS and in
 * HBM, so accumulators, jf, rhof and everything downstream are bit-identical from run to run whatever the array order,
 * the scheduling or the order messages arrive in.  The reference is reproducible by construction (private accumulators
 * reduced in a fixed order, sf_interface/reduce_accumulators.cxx:37-55); this mode is how the engine gets there.  Particle
 * results are the same in both modes.  Costs the cold decks their in-register run sums (every lane adds for itself).
 * Limits of the guarantee: rhob (the charge absorbing faces and emitters leave behind, which div-E cleaning reads) is summed
 * with float atomics in both modes; a single deposit beyond 2^14 x 4.2 q_ref (|x * scale| >= 2^51) wraps in the fixed-point
 * conversion -- give q_ref the LARGEST macro-particle charge of the run when charges differ by orders of magnitude. */
int vpic_hip_set_accumulation(vpic_hip_engine_t *e, int mode, double q_ref);
int vpic_hip_advance_p(vpic_hip_engine_t *e, int sp);       /* species_advance/standard/advance_p.cxx:399-472 (+move_p.c); movers: vpic_hip_species_nm */
int vpic_hip_sort_p(vpic_hip_engine_t *e, int sp);
/* sort_p followed by advance_p of the same species, as ONE call: when the sort is by tile and cell and finds the counts the
 * push before it took (vpic_hip_species_sort_hint), the push writes every particle straight to its sorted place in the
 * second buffer instead of sorting first (the order is that of the cells before this push: sort_p.c:48-101 applied one step
 * earlier) -- where that is the cheaper way for this species: the engine times both ways (a species whose particles have spread
 * between sorts is better sorted first) and keeps to the cheaper one; the particles and the order are the same either way.
 * Otherwise exactly vpic_hip_sort_p + vpic_hip_advance_p.  vpic_hip_step does the same by itself. */
int vpic_hip_sort_advance_p(vpic_hip_engine_t *e, int sp);
/* the species' next advance_p also takes the histogram of the sort that follows it (the caller knows the next step sorts;
 * vpic_hip_step does this by itself): that sort then starts at its scan.  Anything that changes the species in between
 * makes the sort count for itself again. */
int vpic_hip_species_sort_hint(vpic_hip_engine_t *e, int sp);          /* species_advance/standard/sort_p.c:16-102 */
int vpic_hip_energy_p(vpic_hip_engine_t *e, int sp, double *energy); /* species_advance/standard/energy_p.cxx:124-157 (local part) */
int vpic_hip_center_p(vpic_hip_engine_t *e, int sp);        /* species_advance/standard/center_p.cxx: u(-1/2) -> u(0) */
int vpic_hip_uncenter_p(vpic_hip_engine_t *e, int sp);      /* species_advance/standard/uncenter_p.cxx:154-177: u(0) -> u(-1/2) */
int vpic_hip_clear_jf(vpic_hip_engine_t *e);                /* field_advance/standard/sfa.c:188-211 */
int vpic_hip_clear_jf_unload_accumulator(vpic_hip_engine_t *e);   /* the two calls in one pass (advance.cxx:109-110 makes them back to back): same values, bit for bit */
int vpic_hip_synchronize_jf(vpic_hip_engine_t *e);          /* field_advance/standard/remote.c:416-506: local_adjust_jf + faces shared with itself */
/* the pieces of synchronize_jf for a domain that shares some faces with other domains: the local
 * adjustment (local.c:335-368), then per axis IN ORDER x, y, z (remote.c:284-289) either
 * _synchronize_jf_self (both faces wrap onto this domain) or pack_jf / exchange / unpack_jf */
int vpic_hip_local_adjust_jf(vpic_hip_engine_t *e);
int vpic_hip_synchronize_jf_self(vpic_hip_engine_t *e, int axis);
int vpic_hip_advance_b(vpic_hip_engine_t *e, float frac);   /* field_advance/standard/advance_b.c:74-161 */
int vpic_hip_advance_e(vpic_hip_engine_t *e);               /* field_advance/standard/advance_e.c:87-330 (ghosts of faces shared with other domains must be in place) */
/* advance_e in two launches so that the exchange of the remote tangential-B ghosts overlaps the first
 * (advance_e.c:114,153,191-197 does the same around begin/end_remote_ghost_tang_b): part 1 = local ghosts + the planes
 * x = 2..nx, part 2 = the planes x = 1 and nx+1 + the local adjustment (0 = everything, = vpic_hip_advance_e). */
int vpic_hip_advance_e_part(vpic_hip_engine_t *e, int part);
/* make the engine's stream wait for a HIP event recorded on another stream (the communication stream) */
int vpic_hip_stream_wait_event(vpic_hip_engine_t *e, void *hip_event);
int vpic_hip_energy_f(vpic_hip_engine_t *e, double *en6);   /* field_advance/standard/energy_f.c:139-179 (local part) */
/* boundary_p (species_advance/standard/boundary_p.c:77-505), split at the message boundary:
 *   _pack   : classify the movers of every species; absorbed ones go to rhob and are removed,
 *             emigrants are removed and written to the per-face device buffers
 *   _counts : how many injectors wait in each face's send buffer
 *   _inject : append n injectors (device pointer, e.g. a receive buffer) and finish their moves */
int vpic_hip_boundary_p_pack(vpic_hip_engine_t *e);
/* The same exchange without a round trip to the host per phase (multi-GPU driver): mover counts, particle counts
 * and injector counts stay in device memory; a message to the neighbour across face f is a device buffer
 * { int32 header[4] = {injectors in the payload, injectors wanted, 0, 0}; vpic_particle_injector_t payload[cap] }
 * (16 + 48 cap bytes, _exchange_message_bytes) that is sent WHOLE, so both ends know its size without exchanging
 * counts first (boundary_p.c:333-337 sends the count ahead of the payload).  Per step:
 *   _advance_p_async (every species) ; _exchange_begin ; per round { _exchange_pack -> send / receive ->
 *   _exchange_inject per received message } ; _exchange_finish -- the ONE synchronisation: np / nm of every species
 *   and the headers of the received messages come to the host.  Removed particles leave dead slots (index -1) that every
 *   kernel skips and the next sort drops; *flags: 1 = a round had more movers than its kernels were launched for, 2 = a
 *   message was full -- in both cases the movers concerned are still on their lists (_species_nm > 0) and the caller offers
 *   them again with larger messages (the header's `wanted` > `count` tells both ends of a message); a species out of
 *   particle or mover slots is an error (reserve earlier: _species_reserve).
 * Faces that belong to no other domain behave as in _boundary_p_pack; custom handlers (reflux) are not served. */
int vpic_hip_advance_p_async(vpic_hip_engine_t *e, int sp);
/* The overlap of the exchange with the push (north star; the reference's begin / interior / end pattern,
 * field_advance/standard/advance_e.c:114,153,191-197, applied to boundary_p.c:341-384).  A species in the engine's tile
 * order is pushed in two launches: phase 1 = the tiles on the faces shared with other domains and the particles appended
 * since the sort (every particle that can leave the domain this step, up to stragglers), phase 2 = the other tiles.
 * Between the two the caller packs THAT species' movers (_exchange_pack_species) and starts their transfer on its
 * communication stream; stragglers of phase 2 travel in the later rounds, which carry all species.  A species that is not
 * in tile order is pushed whole by phase 1 (phase 2 then does nothing).  phase 0 = _advance_p_async. */
int vpic_hip_advance_p_phase(vpic_hip_engine_t *e, int sp, int phase);
/* _exchange_pack for the species whose bit is set in species_mask only (messages per species: species k is on the wire
 * while species k+1 is pushed).  A shared face given no message (null / capacity 0) is closed for the round. */
int vpic_hip_exchange_pack_species(vpic_hip_engine_t *e, uint32_t species_mask, void *const dev_msg[6], const int32_t cap[6], int mover_cap);
int vpic_hip_exchange_begin(vpic_hip_engine_t *e);
int vpic_hip_exchange_pack(vpic_hip_engine_t *e, void *const dev_msg[6], const int32_t cap[6], int mover_cap);
int vpic_hip_exchange_inject(vpic_hip_engine_t *e, const void *dev_msg, int cap);
int vpic_hip_exchange_finish(vpic_hip_engine_t *e, const void *const *dev_msgs, int n_msgs /* <= 64: 3 rounds x 6 faces x sent and received, and spare */, int32_t *headers /* 4 per message */, int32_t *flags);
int vpic_hip_boundary_p_counts(vpic_hip_engine_t *e, int32_t ns[6]);
void *vpic_hip_boundary_p_send_buffer(vpic_hip_engine_t *e, int face);          /* device vpic_particle_injector_t[] */
/* copy the injectors waiting on `face` (counts from _counts) into a device buffer of the caller */
int vpic_hip_boundary_p_get_injectors(vpic_hip_engine_t *e, int face, void *dev_dst);
int vpic_hip_boundary_p_inject(vpic_hip_engine_t *e, const void *dev_injectors, int n);
/* face messages for domains that share a face with ANOTHER domain (remote.c:61-134, 416-506);
 * dir = direction of travel 0..5; buf = device float buffer of vpic_hip_face_count floats */
int vpic_hip_face_count(const vpic_hip_engine_t *e, int dir);
int vpic_hip_pack_tang_b(vpic_hip_engine_t *e, int dir, void *dev_buf);
int vpic_hip_unpack_tang_b(vpic_hip_engine_t *e, int dir, const void *dev_buf);
int vpic_hip_pack_jf(vpic_hip_engine_t *e, int dir, void *dev_buf);
int vpic_hip_unpack_jf(vpic_hip_engine_t *e, int dir, const void *dev_buf);

/* ---- divergence cleaning family and charge densities ---------------------------
 * Slots of field_advance_methods_t (src/field_advance/field_advance.h:242-302) and accumulate_rho_p
 * (src/species_advance/standard/spa.h:108-113).  As with synchronize_jf, the plain names handle the
 * local faces and the faces a domain shares with itself; a domain that shares faces with OTHER
 * domains calls the pieces (local adjustment, then per axis in order x, y, z either _self or
 * pack / exchange / unpack). */
int vpic_hip_clear_rhof(vpic_hip_engine_t *e);                       /* sfa.c:213-234 */
int vpic_hip_accumulate_rho_p(vpic_hip_engine_t *e, int sp);         /* species_advance/standard/rho_p.c:23-86 (float atomics: sums in another order) */
int vpic_hip_synchronize_rho(vpic_hip_engine_t *e);                  /* remote.c:533-622 */
int vpic_hip_local_adjust_rho(vpic_hip_engine_t *e);                 /* local.c:368-445: rhof then rhob */
int vpic_hip_synchronize_rho_self(vpic_hip_engine_t *e, int axis);
int vpic_hip_rho_count(const vpic_hip_engine_t *e, int dir);         /* floats of a rho message, remote.c:546 */
int vpic_hip_pack_rho(vpic_hip_engine_t *e, int dir, void *dev_buf);
int vpic_hip_unpack_rho(vpic_hip_engine_t *e, int dir, const void *dev_buf);
int vpic_hip_compute_rhob(vpic_hip_engine_t *e);                     /* compute_rhob.c:74-206 */
int vpic_hip_compute_curl_b(vpic_hip_engine_t *e);                   /* compute_curl_b.c:78-318 */
int vpic_hip_synchronize_tang_e_norm_b(vpic_hip_engine_t *e, double *err);   /* remote.c:298-414; *err = this domain's sum of squared differences */
int vpic_hip_compute_div_e_err(vpic_hip_engine_t *e);                /* compute_div_e_err.c:72-207 */
int vpic_hip_clean_div_e(vpic_hip_engine_t *e);                      /* clean_div_e.c:79-187 */
int vpic_hip_compute_div_b_err(vpic_hip_engine_t *e);                /* compute_div_b_err.c:62-92 */
int vpic_hip_clean_div_b(vpic_hip_engine_t *e);                      /* clean_div_b.c:79-247 */
/* Face messages of the family for faces shared with ANOTHER domain (device buffers of
 * _face_message_count floats).  kind 0: normal E (remote.c:136-207), to be in place before
 * _compute_div_e_err / _compute_rhob; kind 1: div_b_err (remote.c:209-281), before _clean_div_b;
 * kind 2: tangential E and normal B (remote.c:298-414): the receiver averages and *err gets the sum
 * of squared differences.  (_compute_curl_b needs the tang_b ghosts of _pack_tang_b.)  The pieces of
 * synchronize_tang_e_norm_b for such a domain: _local_adjust_tang_e_norm_b, then per axis in order
 * x, y, z either _synchronize_tang_e_norm_b_self or pack / exchange / unpack. */
enum { VPIC_HIP_MSG_NORM_E = 0, VPIC_HIP_MSG_DIV_B = 1, VPIC_HIP_MSG_TANG_E_NORM_B = 2 };
int vpic_hip_face_message_count(const vpic_hip_engine_t *e, int kind, int dir);
int vpic_hip_pack_face_message(vpic_hip_engine_t *e, int kind, int dir, void *dev_buf);
int vpic_hip_unpack_face_message(vpic_hip_engine_t *e, int kind, int dir, const void *dev_buf, double *err);
int vpic_hip_local_adjust_tang_e_norm_b(vpic_hip_engine_t *e);
int vpic_hip_synchronize_tang_e_norm_b_self(vpic_hip_engine_t *e, int axis, double *err);
/* local2[0] = weighted sum of squares * dV, local2[1] = volume of this domain: the two numbers the
 * reference sums over ranks (compute_rms_div_e_err.c:156-158, compute_rms_div_b_err.c:90-92) ... */
int vpic_hip_rms_div_e_err_local(vpic_hip_engine_t *e, double *local2);
int vpic_hip_rms_div_b_err_local(vpic_hip_engine_t *e, double *local2);
/* ... and eps0*sqrt(local2[0]/local2[1]) for a run of one domain */
int vpic_hip_compute_rms_div_e_err(vpic_hip_engine_t *e, double *rms);
int vpic_hip_compute_rms_div_b_err(vpic_hip_engine_t *e, double *rms);

/* ---- hydro moments: sf_interface.h:83-163, spa.h:115-123 ------------------------
 * The engine owns one hydro_t array (allocated on first use); a caller accumulates a species into
 * it, synchronises it and reads it back, as dump.cxx:63-70 does per species. */
int vpic_hip_clear_hydro(vpic_hip_engine_t *e);                      /* sf_interface.c:29-36 */
int vpic_hip_accumulate_hydro_p(vpic_hip_engine_t *e, int sp);       /* species_advance/standard/hydro_p.c:24-176 (uses the loaded interpolator; float atomics) */
int vpic_hip_synchronize_hydro(vpic_hip_engine_t *e);                /* sf_interface/hydro.c:28-163: local adjustment + faces shared with itself */
int vpic_hip_local_adjust_hydro(vpic_hip_engine_t *e);               /* sf_interface/hydro.c:165-200 */
int vpic_hip_synchronize_hydro_self(vpic_hip_engine_t *e, int axis);
int vpic_hip_hydro_count(const vpic_hip_engine_t *e, int dir);       /* floats of a hydro face message (14 per node) */
int vpic_hip_pack_hydro(vpic_hip_engine_t *e, int dir, void *dev_buf);
int vpic_hip_unpack_hydro(vpic_hip_engine_t *e, int dir, const void *dev_buf);
int vpic_hip_set_hydro(vpic_hip_engine_t *e, const vpic_hydro_t *h);
int vpic_hip_get_hydro(vpic_hip_engine_t *e, vpic_hydro_t *h);

/* Payload of the reference's field_dump / hydro_dump (src/vpic/dump.cxx:1116-1364, 1366-1552), gathered
 * on the device so that only the selected, strided bytes cross PCIe.
 *   what   : VPIC_HIP_DUMP_FIELDS (records of 20 32-bit words) or VPIC_HIP_DUMP_HYDRO (16 words; the
 *            state left by accumulate_hydro_p / synchronize_hydro)
 *   layout : VPIC_HIP_DUMP_BAND        out = [nwords][nz/sz+2][ny/sy+2][nx/sx+2] words; word w of a
 *                                      record is taken as the reference does, `((uint32_t*)&rec)[w]`
 *                                      (w up to 23 for fields: 16-19 hold two material ids each, 20-23
 *                                      alias the next record, dump.cxx:1200-1203; past the last
 *                                      record they read as 0 here)
 *            VPIC_HIP_DUMP_INTERLEAVE  out = [nz/sz+2][ny/sy+2][nx/sx+2] whole records (dump.cxx:1331-1357)
 *            VPIC_HIP_DUMP_INTERLEAVE_INNER  hydro_dump's interleaved shape as written: [nz/sz][ny/sy]
 *                                      [nx/sx] records at offsets 0, s-1, 2s-1, ...; with unit strides
 *                                      the first nx*ny*nz records of the array (dump.cxx:1518-1543)
 * Output index i of an axis with n cells, n/s outputs: 0 -> 0, n/s+1 -> n+1, else i when all three
 * strides are 1 and i*s-1 otherwise -- also on an axis whose own stride is 1, as the reference does
 * (dump.cxx:1262-1273).  Strides must divide nx, ny, nz.  `out` is HOST memory of out_bytes. */
enum { VPIC_HIP_DUMP_FIELDS = 0, VPIC_HIP_DUMP_HYDRO = 1 };
enum { VPIC_HIP_DUMP_BAND = 0, VPIC_HIP_DUMP_INTERLEAVE = 1, VPIC_HIP_DUMP_INTERLEAVE_INNER = 2 };
int vpic_hip_dump_gather(vpic_hip_engine_t *e, int what, int layout, const int32_t *words, int nwords,
                         int sx, int sy, int sz, void *out, size_t out_bytes);

/* ---- the GPU-aware transport: RCCL point-to-point over xGMI, one rank per GPU ---------------------------------------
 * What the reference's port layer does with MPI (src/util/mp/dmp/mp_dmp.c:241-266 mp_begin_send / mp_begin_recv;
 * src/grid/grid_comm.c:7-78): a non-blocking send and receive per shared face, all faces posted together, waited for
 * where the data is needed.  Here the buffers are DEVICE memory (the engine's pack_* / exchange_pack_* outputs), the
 * transfers run on a communication stream of the communicator's own and are ordered against the engine's stream with
 * events: _start returns at once, _finish makes the ENGINE'S STREAM (not the host) wait.  The process launcher (MPI,
 * torch.distributed.run ...) only carries the 128-byte id from rank 0 to the others.  A rank may send to itself (the
 * reference self-sends across a periodic axis it owns alone, grid_comm.c:17-19,49): a 1-rank communicator is valid.
 * RCCL is loaded with dlopen by the first of these calls; one rank per device (RCCL refuses to share one). */
#define VPIC_HIP_COMM_ID_BYTES 128
typedef struct vpic_hip_comm vpic_hip_comm_t;
int vpic_hip_comm_unique_id(void *id128);                       /* rank 0; broadcast the bytes to every rank */
int vpic_hip_comm_create(vpic_hip_comm_t **c, vpic_hip_engine_t *e, const void *id128, int nranks, int rank);
int vpic_hip_comm_destroy(vpic_hip_comm_t *c);
/* post n_send + n_recv messages as ONE group behind everything the engine's stream holds so far; messages between a pair of
 * ranks match in the order listed (list by direction 0..5 on every rank); *token names the exchange for _finish (at most
 * 64 exchanges may be pending) */
int vpic_hip_comm_start(vpic_hip_comm_t *c, int n_send, const void *const *sbuf, const size_t *sbytes, const int *speer,
                        int n_recv, void *const *rbuf, const size_t *rbytes, const int *rpeer, int *token);
int vpic_hip_comm_finish(vpic_hip_comm_t *c, int token);        /* the engine's stream waits for that exchange */
int vpic_hip_comm_stats(vpic_hip_comm_t *c, int64_t *messages_sent, int64_t *bytes_sent);
/* timing of the exchanges with events (bench.py's `exchange` block): on = 1 starts and clears, 0 stops, -1 only reads; returns
 * the sums so far -- time on the communication stream, time the engine's stream stood still for the exchanges, their number
 * (call when the device is idle: it waits for the events) */
int vpic_hip_comm_timing(vpic_hip_comm_t *c, int on, double *exchange_ms, double *exposed_ms, int64_t *exchanges);

/* staging helpers for a host whose transport moves host memory (plain MPI): device scratch buffers
 * for the pack / unpack / inject calls above, and copies ordered after / before the engine's work */
void *vpic_hip_device_alloc(vpic_hip_engine_t *e, size_t bytes);
void vpic_hip_device_free(vpic_hip_engine_t *e, void *p);
int vpic_hip_copy_to_host(vpic_hip_engine_t *e, void *host, const void *dev, size_t bytes);
int vpic_hip_copy_from_host(vpic_hip_engine_t *e, void *dev, const void *host, size_t bytes);

/* one vpic_simulation::advance() of a domain that needs no other domain (src/vpic/advance.cxx:
 * 38-214: clear_accumulators, sort when due, advance_p all species, boundary_p, clear_jf, unload,
 * synchronize_jf, advance_b half, advance_e, advance_b half, load_interpolator).
 * sort_interval == 0: never sort; sort_interval < 0: ADAPTIVE, at the latest every -sort_interval
 * steps -- the engine times each species' sorts and pushes with HIP events and sorts a species again
 * as soon as its last push cost at least the average cost per step of the current cycle, the sort
 * included (sorting changes the array order only, not the physics). */
int vpic_hip_step(vpic_hip_engine_t *e, int64_t step, int sort_interval);

/* the adaptive decision for one species, for hosts that drive the steps themselves: *due = 1 when
 * species sp should be sorted before its next advance_p (first call: always; it also switches the
 * event timing of sort_p / advance_p on) */
int vpic_hip_sort_due(vpic_hip_engine_t *e, int sp, int max_interval, int *due);
int vpic_hip_measure_disorder(vpic_hip_engine_t *e, int sp, double *fraction);   /* descents of the voxel index / particle (sampled) */
/* The order the species' array is in: 0 = none known (loaded, uploaded, or moved on since a sort by voxel),
 * 1 = sorted by voxel and partition[] valid (sort_p.c:48-58), 2 = TILE order: grouped by 4x4x4-cell tile, what the
 * engine's own sort policy (vpic_hip_sort_due / vpic_hip_step with sort_interval < 0) gives a species whose particles
 * mostly leave their cell every step; advance_p then runs one workgroup per tile with the tile and its halo as LDS
 * window.  No result depends on the order; partition[] exists for order 1 only. */
int vpic_hip_species_sort_order(vpic_hip_engine_t *e, int sp, int *order);
/* what the engine knows about a species' last advance_p and its sorting, for diagnostics (bench.py, tools): out[0] particles
 * that left their cell, out[1] particles in the fullest tile at the last tile sort, out[2] runs of deposits that missed their
 * tile's LDS window, out[3] sorts so far, out[4] sorts vpic_hip_step made ahead of a fixed interval, out[5] 1 when the species
 * was found too clumped for the tile order, out[6] 1 when it is sorted by tile only, out[7] dead slots.  Values the device
 * publishes without being waited for: a launch or two old. */
int vpic_hip_species_stats(vpic_hip_engine_t *e, int sp, int64_t out[8]);
/* Who chooses the order vpic_hip_sort_p sorts into: 0 (default) = the reference's, by voxel (sort_p.c:48-58; partition[]
 * valid afterwards); 1 = the engine: TILE order for charged species of up to 2^30 particles.  A species whose sorts are
 * asked for by vpic_hip_sort_due's policy is sorted the engine's way in either mode. */
int vpic_hip_set_sort_order(vpic_hip_engine_t *e, int order);

/* HIP-event timing of the advance_p launches on the engine's stream (bench.py roofline leg) */
int vpic_hip_profile_enable(vpic_hip_engine_t *e, int on);
int vpic_hip_profile_read(vpic_hip_engine_t *e, double *advance_p_ms, int64_t *launches, int64_t *particles);
/* The same for the launches of advance_p that SORT the species as they push it (vpic_hip_step, fixed sort interval: a sort
 * that finds the counts the push before it took writes every particle to its sorted place instead of sorting first --
 * sort_p.c:48-101's result, one step stale, for 12 more bytes per particle): booked apart from the plain launches above. */
int vpic_hip_profile_read_sorting(vpic_hip_engine_t *e, double *advance_p_ms, int64_t *launches, int64_t *particles);
/* ... and the plain launches of ONE species (a deck whose species differ: charged against charge-0 tracer copies) */
int vpic_hip_profile_read_species(vpic_hip_engine_t *e, int sp, double *advance_p_ms, int64_t *launches, int64_t *particles);

#ifdef __cplusplus
}
#endif
#endif /* VPIC_HIP_H */

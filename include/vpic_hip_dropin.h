/* vpic_hip_dropin.h -- the reference's own L3 C entry points, served by the HIP engine.
 *
 * Each function below has the signature and argument meaning of the reference function named in
 * its comment and works on the caller's HOST arrays in the reference's array-of-struct layouts:
 * it uploads what the kernel reads, runs the HIP kernel(s) of libvpic_hip.so, and downloads what
 * the kernel writes.  A maintainer of the reference links these in place of the objects listed
 * (see INTEGRATION.md); `#define VPIC_HIP_DROPIN_NAMES` before including this header additionally
 * maps the reference's bare names (advance_p, load_interpolator, ...) onto them.
 *
 * This is the literal drop-in: correct and bit-compatible where the reference is deterministic,
 * but it pays a host<->device round trip per call.  The production path is the resident engine
 * (vpic_hip.h), which keeps all per-step state in HBM.
 *
 * Error convention = the reference's (src/util/util_base.h:213-219): a message on stderr naming
 * the call, then exit(1).  There is no CPU fallback: without a HIP device every call fails so.
 */
#ifndef VPIC_HIP_DROPIN_H
#define VPIC_HIP_DROPIN_H
#include "vpic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* grid_t, byte for byte (src/grid/grid.h:112-167; sizeof 240, offsets pinned below) */
typedef struct vpic_grid {
  void *mp;                       /* mp_handle */
  float dt, cvac, eps0, damp;
  float x0, y0, z0, x1, y1, z1;
  float dx, dy, dz, rdx, rdy, rdz;
  int32_t nx, ny, nz;
  int32_t bc[27];                 /* (-1:1,-1:1,-1:1) FORTRAN; bc[13] = this rank (ops.c:46) */
  int64_t *range;
  int64_t *neighbor;              /* 6*nv: -x,-y,-z,+x,+y,+z neighbours, or particle bc codes */
  int64_t rangel, rangeh;
  int32_t nb;
  void *boundary;
} vpic_grid_t;
VPIC_HIP_STATIC_ASSERT(sizeof(vpic_grid_t) == 240 && offsetof(vpic_grid_t, dt) == 8 &&
                       offsetof(vpic_grid_t, x0) == 24 && offsetof(vpic_grid_t, dx) == 48 &&
                       offsetof(vpic_grid_t, rdx) == 60 && offsetof(vpic_grid_t, nx) == 72 &&
                       offsetof(vpic_grid_t, bc) == 84 && offsetof(vpic_grid_t, range) == 192 &&
                       offsetof(vpic_grid_t, neighbor) == 200 && offsetof(vpic_grid_t, rangel) == 208 &&
                       offsetof(vpic_grid_t, rangeh) == 216 && offsetof(vpic_grid_t, nb) == 224 &&
                       offsetof(vpic_grid_t, boundary) == 232, "grid_t layout");

/* species_t up to the fields the kernels use (src/species_advance/species_advance.h:61-93) */
typedef struct vpic_species {
  int32_t id;
  int32_t np, max_np;
  vpic_particle_t *p;
  int32_t nm, max_nm;
  vpic_particle_mover_t *pm;
  float q_m;
  int32_t sort_interval;
  int32_t sort_out_of_place;
  int32_t *partition;             /* nv+1, allocated by sort_p when NULL (sort_p.c:32) */
  struct vpic_species *next;
  char name[1];
} vpic_species_t;
VPIC_HIP_STATIC_ASSERT(offsetof(vpic_species_t, np) == 4 && offsetof(vpic_species_t, p) == 16 &&
                       offsetof(vpic_species_t, nm) == 24 && offsetof(vpic_species_t, pm) == 32 &&
                       offsetof(vpic_species_t, q_m) == 40 && offsetof(vpic_species_t, sort_interval) == 44 &&
                       offsetof(vpic_species_t, partition) == 56 && offsetof(vpic_species_t, next) == 64 &&
                       offsetof(vpic_species_t, name) == 72, "species_t layout");

/* Grids that share faces with OTHER ranks.  The reference exchanges ghost planes, boundary sums and particles through its
 * port layer (src/grid/grid_comm.c:7-78: begin/end_recv_port, size/begin/end_send_port over src/util/mp).  The twins do not
 * link against that layer; the host hands them a transport once and they run the exchanges of remote.c:61-134,298-414,
 * 416-506,533-622, compute_*_err.c, hydro.c:28-163 and boundary_p.c:341-497 through it (message payloads are the engine's
 * own face messages; both ends of a message are twins).  Without a transport a grid that shares a face is refused.
 *   exchange: for every direction d (0..2: towards -x,-y,-z; 3..5: towards +x,+y,+z) with n_send[d] > 0 the n_send[d] bytes
 *             at send[d] travel through the face in direction d to the rank behind it (g->bc[BOUNDARY(d)]); with
 *             n_recv[d] > 0, n_recv[d] bytes travelling in direction d -- from the rank behind the OPPOSITE face -- arrive
 *             in recv[d].  Returns when everything posted has arrived (exactly what begin_recv_port(d) ... end_send_port(d)
 *             do for the port named d: oracle/dropin_shim.c implements it with those calls).
 *   allsum_d: v[0..n) summed over all ranks in place (mp_allsum_d). */
typedef struct vpic_hip_ref_transport {
  void (*exchange)(void *ctx, const vpic_grid_t *g, const void *const send[6], const size_t n_send[6], void *const recv[6], const size_t n_recv[6]);
  void (*allsum_d)(void *ctx, const vpic_grid_t *g, double *v, int n);
  void *ctx;
} vpic_hip_ref_transport_t;
void vpic_hip_ref_set_transport(const vpic_hip_ref_transport_t *t);

/* number of accumulator copies the caller's array holds, 1 + max n_pipeline (sf_interface.c:66-72);
 * advance_p adds into copy 0, the others stay as they are.  Default 1. */
void vpic_hip_ref_set_accumulator_copies(int n);

/* src/sf_interface/sf_interface.h:108-110 -> load_interpolator.cxx:284-369 */
void vpic_hip_ref_load_interpolator(vpic_interpolator_t *fi, const vpic_field_t *f, const vpic_grid_t *g);
/* src/sf_interface/sf_interface.h:116-118 -> clear_accumulators.c:26-49 */
void vpic_hip_ref_clear_accumulators(vpic_accumulator_t *a, const vpic_grid_t *g);
/* src/sf_interface/sf_interface.h:128-130 -> reduce_accumulators.cxx:143-165 */
void vpic_hip_ref_reduce_accumulators(vpic_accumulator_t *a, const vpic_grid_t *g);
/* src/sf_interface/sf_interface.h:143-146 -> unload_accumulator.cxx:81-121 */
void vpic_hip_ref_unload_accumulator(vpic_field_t *f, const vpic_accumulator_t *a, const vpic_grid_t *g);
/* src/species_advance/standard/spa.h:58-66 -> advance_p.cxx:399-472 (+ move_p.c).  Returns nm;
 * pm[0..nm) ascending in particle index as boundary_p.c:168-176 requires. */
int vpic_hip_ref_advance_p(vpic_particle_t *p0, int np, const float q_m, vpic_particle_mover_t *pm, int max_nm,
                           vpic_accumulator_t *a0, const vpic_interpolator_t *f0, const vpic_grid_t *g);
/* src/species_advance/standard/spa.h:101-106 -> energy_p.cxx:124-157 (this rank's share; the
 * caller's mp_allsum_d stays where it is) */
double vpic_hip_ref_energy_p(const vpic_particle_t *p0, int np, float q_m, const vpic_interpolator_t *f0,
                             const vpic_grid_t *g);
/* src/species_advance/standard/spa.h:75-93 -> center_p.cxx, uncenter_p.cxx:154-177 */
void vpic_hip_ref_center_p(vpic_particle_t *p0, int np, const float q_m, const vpic_interpolator_t *f0, const vpic_grid_t *g);
void vpic_hip_ref_uncenter_p(vpic_particle_t *p0, int np, const float q_m, const vpic_interpolator_t *f0, const vpic_grid_t *g);
/* src/species_advance/standard/spa.h:50-54 -> move_p.c:20-136: finish the move of particle pm->i;
 * returns 1 (and leaves the remaining displacement in pm) when it stopped on a face boundary_p
 * must handle */
int vpic_hip_ref_move_p(vpic_particle_t *p0, vpic_particle_mover_t *pm, vpic_accumulator_t *a0, const vpic_grid_t *g);
/* src/species_advance/standard/spa.h:35-43 -> boundary_p.c:77-505, grids of one rank (absorbing
 * faces: rhob + removal); rng is unused without custom boundary handlers */
void vpic_hip_ref_boundary_p(vpic_species_t *sp_list, vpic_field_t *f, vpic_accumulator_t *a0, const vpic_grid_t *g, void *rng);
/* src/species_advance/standard/spa.h:23-25 -> sort_p.c:16-102 */
void vpic_hip_ref_sort_p(vpic_species_t *sp, const vpic_grid_t *g);
/* field_advance_methods_t slots (src/field_advance/field_advance.h:185-302), standard solver:
 * advance_b.c:74-161, advance_e.c:87-330, sfa.c:188-211, remote.c:416-506, energy_f.c:139-179.
 * Faces shared with other ranks are served through the registered transport (vpic_hip_ref_set_transport). */
void vpic_hip_ref_advance_b(vpic_field_t *f, const vpic_grid_t *g, float frac);
void vpic_hip_ref_advance_e(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g);
void vpic_hip_ref_clear_jf(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_synchronize_jf(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_energy_f(double *energy6, const vpic_field_t *f, const vpic_material_coefficient_t *m,
                           const vpic_grid_t *g);
/* the remaining slots of the table (field_advance.h:242-302) and accumulate_rho_p (spa.h:108-113):
 * sfa.c:213-234, rho_p.c:23-86, remote.c:533-622, compute_rhob.c, compute_curl_b.c, remote.c:298-414,
 * compute_div_e_err.c, compute_rms_div_e_err.c, clean_div_e.c, compute_div_b_err.c,
 * compute_rms_div_b_err.c, clean_div_b.c.  The rms values, the synchronisation error and energy_f are global sums when a
 * transport is registered (the reference's mp_allsum_d), the rank's own otherwise. */
void vpic_hip_ref_clear_rhof(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_accumulate_rho_p(vpic_field_t *f, const vpic_particle_t *p0, int np, const vpic_grid_t *g);
/* spa.h:30-33 / boundary_p.c:9-71: one particle's charge into rhob (vpic.hxx:483-484 calls it inline) */
void vpic_hip_ref_accumulate_rhob(vpic_field_t *f, const vpic_particle_t *p, const vpic_grid_t *g);
void vpic_hip_ref_synchronize_rho(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_compute_rhob(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g);
void vpic_hip_ref_compute_curl_b(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g);
double vpic_hip_ref_synchronize_tang_e_norm_b(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_compute_div_e_err(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g);
double vpic_hip_ref_compute_rms_div_e_err(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_clean_div_e(vpic_field_t *f, const vpic_material_coefficient_t *m, const vpic_grid_t *g);
void vpic_hip_ref_compute_div_b_err(vpic_field_t *f, const vpic_grid_t *g);
double vpic_hip_ref_compute_rms_div_b_err(vpic_field_t *f, const vpic_grid_t *g);
void vpic_hip_ref_clean_div_b(vpic_field_t *f, const vpic_grid_t *g);
/* hydro: sf_interface.h:90-92,158-163 -> sf_interface.c:29-36, hydro.c:28-200; spa.h:115-123 -> hydro_p.c:24-176 */
void vpic_hip_ref_clear_hydro(vpic_hydro_t *h, const vpic_grid_t *g);
void vpic_hip_ref_accumulate_hydro_p(vpic_hydro_t *h0, const vpic_particle_t *p0, int np, float q_m,
                                     const vpic_interpolator_t *f0, const vpic_grid_t *g);
void vpic_hip_ref_synchronize_hydro(vpic_hydro_t *h, const vpic_grid_t *g);
void vpic_hip_ref_local_adjust_hydro(vpic_hydro_t *h, const vpic_grid_t *g);
/* number of materials in the table behind `m` (the reference passes an opaque pointer whose
 * length only new_material_coefficients knows); default 1 */
void vpic_hip_ref_set_material_count(int n);

/* ---- the allocation slots, so that the whole method table can be this library's ----------------------
 * material_t (src/material/material.h:43-52); new_* return zeroed, 128-byte aligned blocks that the
 * matching delete_* of THIS library frees (sf_interface.c:29-75, sfa.c:30-78,80-186). */
typedef struct vpic_material {
  uint16_t id;
  float epsx, epsy, epsz, mux, muy, muz, sigmax, sigmay, sigmaz, zetax, zetay, zetaz;
  struct vpic_material *next;
  char name[1];
} vpic_material_t;
vpic_field_t *vpic_hip_ref_new_field(vpic_grid_t *g);
void vpic_hip_ref_delete_field(vpic_field_t *f);
/* sfa.c:80-177; also records the material count for the other entry points (vpic_hip_ref_set_material_count) */
vpic_material_coefficient_t *vpic_hip_ref_new_material_coefficients(vpic_grid_t *g, vpic_material_t *m_list);
void vpic_hip_ref_delete_material_coefficients(vpic_material_coefficient_t *mc);
vpic_hydro_t *vpic_hip_ref_new_hydro(vpic_grid_t *g);
void vpic_hip_ref_delete_hydro(vpic_hydro_t *h);
vpic_interpolator_t *vpic_hip_ref_new_interpolator(vpic_grid_t *g);
void vpic_hip_ref_delete_interpolator(vpic_interpolator_t *fi);
/* (1 + the copies announced with vpic_hip_ref_set_accumulator_copies) arrays of POW2_CEIL(nv, 2) records */
vpic_accumulator_t *vpic_hip_ref_new_accumulators(vpic_grid_t *g);
void vpic_hip_ref_delete_accumulators(vpic_accumulator_t *a);
/* The 20 slots of field_advance_methods_t in the reference's order (field_advance.h:185-302), all of them
 * entry points of this library: `(field_advance_methods_t *)vpic_hip_ref_field_advance_methods` is what a
 * deck hands to finalize_field_advance. */
extern void *const vpic_hip_ref_field_advance_methods[20];

#ifdef VPIC_HIP_DROPIN_NAMES
#define load_interpolator   vpic_hip_ref_load_interpolator
#define clear_accumulators  vpic_hip_ref_clear_accumulators
#define reduce_accumulators vpic_hip_ref_reduce_accumulators
#define unload_accumulator  vpic_hip_ref_unload_accumulator
#define advance_p           vpic_hip_ref_advance_p
#define energy_p            vpic_hip_ref_energy_p
#define sort_p              vpic_hip_ref_sort_p
#endif

#ifdef __cplusplus
}
#endif
#endif /* VPIC_HIP_DROPIN_H */

/* dropin_shim.c -- TEST INFRASTRUCTURE: the translation unit a maintainer of the reference would add to run
 * the REFERENCE ITSELF on libvpic_hip.so.  It is compiled against the reference's
 * own headers and defines the hot-path symbols of the objects left out of the link -- advance_p, move_p,
 * boundary_p, sort_p, energy_p, center_p, uncenter_p, accumulate_rho_p, accumulate_hydro_p,
 * load_interpolator, clear/reduce/unload accumulators, synchronize/local_adjust hydro and the standard
 * field advance's method table -- by calling the twins of include/vpic_hip_dropin.h.
 *   make -C oracle dropin DECK=... OUT=name  ->  oracle/_ref/name.dropin.exe
 * Everything else in that executable (main, vpic_simulation, grids, species lists, MPI layer, dumps) is the
 * reference's own code, compiled from where it lies.  On several ranks the twins' face messages travel through the
 * reference's own port layer (src/grid/grid_comm.c): the transport below, registered on first use. */
#include "spa.h"
#include "sf_interface.h"
#include "field_advance.h"
#include "pipelines.h"
#include <vpic_hip_dropin.h>

#include <string.h>

#define G(g) ((const vpic_grid_t *)(g))

/* the transport of include/vpic_hip_dropin.h over the reference's ports: a message travelling in direction d leaves through
 * the port of that name (grid_comm.c:7-78: the receiver posts begin_recv_port with the SAME (i,j,k)) */
static const int dir_ijk[6][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
static void port_exchange(void *ctx, const vpic_grid_t *vg, const void *const send[6], const size_t n_send[6],
                          void *const recv[6], const size_t n_recv[6]) {
  const grid_t *g = (const grid_t *)vg;
  int d;
  (void)ctx;
  for (d = 0; d < 6; d++) if (n_recv[d]) begin_recv_port(dir_ijk[d][0], dir_ijk[d][1], dir_ijk[d][2], (int)n_recv[d], g);
  for (d = 0; d < 6; d++) if (n_send[d]) {
    void *buf = size_send_port(dir_ijk[d][0], dir_ijk[d][1], dir_ijk[d][2], (int)n_send[d], g);
    if (buf) { memcpy(buf, send[d], n_send[d]); begin_send_port(dir_ijk[d][0], dir_ijk[d][1], dir_ijk[d][2], (int)n_send[d], g); }
  }
  for (d = 0; d < 6; d++) if (n_recv[d]) {
    const void *buf = end_recv_port(dir_ijk[d][0], dir_ijk[d][1], dir_ijk[d][2], g);
    if (buf) memcpy(recv[d], buf, n_recv[d]);
  }
  for (d = 0; d < 6; d++) if (n_send[d]) end_send_port(dir_ijk[d][0], dir_ijk[d][1], dir_ijk[d][2], g);
}
static void port_allsum_d(void *ctx, const vpic_grid_t *vg, double *v, int n) {
  const grid_t *g = (const grid_t *)vg;
  double tmp[16];
  (void)ctx;
  if (n > 16) n = 16;
  mp_allsum_d(v, tmp, n, g->mp);
  memcpy(v, tmp, sizeof(double) * (size_t)n);
}
/* (registered when the executable is loaded: initialize() reaches the field table's slots -- which point straight at the
 * twins -- before any of the wrappers below runs) */
__attribute__((constructor)) static void announce_transport(void) {
  static int done = 0;
  if (!done) { const vpic_hip_ref_transport_t t = {port_exchange, port_allsum_d, 0}; vpic_hip_ref_set_transport(&t); done = 1; }
}

static void announce_copies(void) {
  int n;
  announce_transport();
  n = serial.n_pipeline > thread.n_pipeline ? serial.n_pipeline : thread.n_pipeline;   /* sf_interface.c:61-66 */
  vpic_hip_ref_set_accumulator_copies(1 + n);
}

int advance_p(particle_t *p0, int np, const float q_m, particle_mover_t *pm, int max_nm, accumulator_t *a0,
              const interpolator_t *f0, const grid_t *g) {
  announce_copies();
  return vpic_hip_ref_advance_p((vpic_particle_t *)p0, np, q_m, (vpic_particle_mover_t *)pm, max_nm,
                                (vpic_accumulator_t *)a0, (const vpic_interpolator_t *)f0, G(g));
}
int move_p(particle_t *p0, particle_mover_t *m, accumulator_t *a0, const grid_t *g) {
  return vpic_hip_ref_move_p((vpic_particle_t *)p0, (vpic_particle_mover_t *)m, (vpic_accumulator_t *)a0, G(g));
}
void boundary_p(species_t *sp_list, field_t *f, accumulator_t *a0, const grid_t *g, mt_rng_t *rng) {
  announce_transport();
  vpic_hip_ref_boundary_p((vpic_species_t *)sp_list, (vpic_field_t *)f, (vpic_accumulator_t *)a0, G(g), rng);
}
void accumulate_rhob(field_t *f, const particle_t *p, const grid_t *g) {
  vpic_hip_ref_accumulate_rhob((vpic_field_t *)f, (const vpic_particle_t *)p, G(g));
}
void sort_p(species_t *sp, const grid_t *g) { vpic_hip_ref_sort_p((vpic_species_t *)sp, G(g)); }
double energy_p(const particle_t *p0, int np, float q_m, const interpolator_t *f0, const grid_t *g) {
  double local = vpic_hip_ref_energy_p((const vpic_particle_t *)p0, np, q_m, (const vpic_interpolator_t *)f0, G(g)), global;
  mp_allsum_d(&local, &global, 1, g->mp);                  /* energy_p.cxx:155 */
  return global;
}
void center_p(particle_t *p0, int np, const float q_m, const interpolator_t *f0, const grid_t *g) {
  vpic_hip_ref_center_p((vpic_particle_t *)p0, np, q_m, (const vpic_interpolator_t *)f0, G(g));
}
void uncenter_p(particle_t *p0, int np, const float q_m, const interpolator_t *f0, const grid_t *g) {
  vpic_hip_ref_uncenter_p((vpic_particle_t *)p0, np, q_m, (const vpic_interpolator_t *)f0, G(g));
}
void accumulate_rho_p(field_t *f, const particle_t *p0, int np, const grid_t *g) {
  vpic_hip_ref_accumulate_rho_p((vpic_field_t *)f, (const vpic_particle_t *)p0, np, G(g));
}
void accumulate_hydro_p(hydro_t *h0, const particle_t *p0, int np, float q_m, const interpolator_t *f0, const grid_t *g) {
  vpic_hip_ref_accumulate_hydro_p((vpic_hydro_t *)h0, (const vpic_particle_t *)p0, np, q_m, (const vpic_interpolator_t *)f0, G(g));
}
void load_interpolator(interpolator_t *fi, const field_t *f, const grid_t *g) {
  vpic_hip_ref_load_interpolator((vpic_interpolator_t *)fi, (const vpic_field_t *)f, G(g));
}
void clear_accumulators(accumulator_t *a, const grid_t *g) { announce_copies(); vpic_hip_ref_clear_accumulators((vpic_accumulator_t *)a, G(g)); }
void reduce_accumulators(accumulator_t *a, const grid_t *g) { announce_copies(); vpic_hip_ref_reduce_accumulators((vpic_accumulator_t *)a, G(g)); }
void unload_accumulator(field_t *f, const accumulator_t *a, const grid_t *g) {
  vpic_hip_ref_unload_accumulator((vpic_field_t *)f, (const vpic_accumulator_t *)a, G(g));
}
void synchronize_hydro(hydro_t *h, const grid_t *g) { announce_transport(); vpic_hip_ref_synchronize_hydro((vpic_hydro_t *)h, G(g)); }
void local_adjust_hydro(hydro_t *h, const grid_t *g) { vpic_hip_ref_local_adjust_hydro((vpic_hydro_t *)h, G(g)); }

/* field_advance.h:334-347: `standard_field_advance` is this symbol; all 20 slots are the library's */
field_advance_methods_t _standard_field_advance[1] = {{
  (field_t *(*)(grid_t *))vpic_hip_ref_new_field,
  (void (*)(field_t *))vpic_hip_ref_delete_field,
  (material_coefficient_t *(*)(grid_t *, material_t *))vpic_hip_ref_new_material_coefficients,
  (void (*)(material_coefficient_t *))vpic_hip_ref_delete_material_coefficients,
  (void (*)(field_t *, const grid_t *, float))vpic_hip_ref_advance_b,
  (void (*)(field_t *, const material_coefficient_t *, const grid_t *))vpic_hip_ref_advance_e,
  (void (*)(double *, const field_t *, const material_coefficient_t *, const grid_t *))vpic_hip_ref_energy_f,
  (void (*)(field_t *, const grid_t *))vpic_hip_ref_clear_jf,
  (void (*)(field_t *, const grid_t *))vpic_hip_ref_synchronize_jf,
  (void (*)(field_t *, const grid_t *))vpic_hip_ref_clear_rhof,
  (void (*)(field_t *, const grid_t *))vpic_hip_ref_synchronize_rho,
  (void (*)(field_t *, const material_coefficient_t *, const grid_t *))vpic_hip_ref_compute_rhob,
  (void (*)(field_t *, const material_coefficient_t *, const grid_t *))vpic_hip_ref_compute_curl_b,
  (double (*)(field_t *, const grid_t *))vpic_hip_ref_synchronize_tang_e_norm_b,
  (void (*)(field_t *, const material_coefficient_t *, const grid_t *))vpic_hip_ref_compute_div_e_err,
  (double (*)(field_t *, const grid_t *))vpic_hip_ref_compute_rms_div_e_err,
  (void (*)(field_t *, const material_coefficient_t *, const grid_t *))vpic_hip_ref_clean_div_e,
  (void (*)(field_t *, const grid_t *))vpic_hip_ref_compute_div_b_err,
  (double (*)(field_t *, const grid_t *))vpic_hip_ref_compute_rms_div_b_err,
  (void (*)(field_t *, const grid_t *))vpic_hip_ref_clean_div_b }};

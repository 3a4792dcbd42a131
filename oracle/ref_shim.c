/* ref_shim.c -- TEST INFRASTRUCTURE.  Ours, not the reference's: a few helpers linked into
 * oracle/_ref/libvpic_ref.so so that the reference's own C functions (advance_p, move_p,
 * load_interpolator, advance_b, advance_e, ... compiled from /root/reference, untouched) can be
 * driven from Python/ctypes by oracle/gen_golden.py and tests/test_oracle_vs_ref.py.
 * It only calls the reference's public API (src/main.cxx:72-79 boot order, grid/grid.h,
 * material/material.h, field_advance/field_advance.h).
 */
#include "spa.h"        /* pulls species_advance.h -> sf_interface.h -> field_advance.h -> grid.h */
#include <stddef.h>

static int booted = 0;

/* src/main.cxx:72-79: pipeline dispatchers first, then message passing */
int ref_boot(int tpp) {
  static char arg0[] = "ref_shim";
  static char *argv[] = {arg0, NULL};
  if (booted) return thread.n_pipeline;
  thread.boot(tpp, 1);
  serial.boot(tpp, 1);
  mp_init(1, argv);
  booted = 1;
  return thread.n_pipeline;
}

int ref_n_pipeline(void) { return thread.n_pipeline; }

/* A single-rank periodic box (what a deck's define_periodic_grid does, vpic/vpic.hxx:253-263) */
grid_t *ref_new_periodic_grid(float dt, float cvac, float eps0, float damp,
                              double lx, double ly, double lz, int nx, int ny, int nz) {
  grid_t *g = new_grid();
  g->dt = dt; g->cvac = cvac; g->eps0 = eps0; g->damp = damp;
  partition_periodic_box(g, 0, 0, 0, lx, ly, lz, nx, ny, nz, 1, 1, 1);
  return g;
}

/* face = 0..5 (-x,-y,-z,+x,+y,+z); what set_domain_field_bc / set_domain_particle_bc do
 * (vpic/vpic.hxx:331-357) */
void ref_set_face_bc(grid_t *g, int face, int fbc, int pbc) {
  static const int b[6] = { BOUNDARY(-1,0,0), BOUNDARY(0,-1,0), BOUNDARY(0,0,-1),
                            BOUNDARY( 1,0,0), BOUNDARY(0, 1,0), BOUNDARY(0,0, 1) };
  set_fbc(g, b[face], fbc);
  set_pbc(g, b[face], pbc);
}

void ref_grid_info(const grid_t *g, float *out10, int *n3) {
  out10[0] = g->dt; out10[1] = g->cvac; out10[2] = g->eps0; out10[3] = g->damp;
  out10[4] = g->dx; out10[5] = g->dy; out10[6] = g->dz;
  out10[7] = g->rdx; out10[8] = g->rdy; out10[9] = g->rdz;
  n3[0] = g->nx; n3[1] = g->ny; n3[2] = g->nz;
}

/* One vacuum material -> coefficient table (field_advance/standard/sfa.c:80-177) */
material_coefficient_t *ref_new_vacuum_coefficients(grid_t *g) {
  material_t *m_list = NULL;
  new_material("vacuum", 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, &m_list);
  return _standard_field_advance->new_material_coefficients(g, m_list);
}

/* Three materials -- vacuum, an anisotropic dielectric / magnetic one, an anisotropic conductor -- with
 * ids 0, 1, 2 in this order; props9[k] = {epsx,epsy,epsz, mux,muy,muz, sigmax,sigmay,sigmaz} */
material_coefficient_t *ref_new_coefficients(grid_t *g, const float *props9, int n) {
  material_t *m_list = NULL;
  char name[16];
  for (int k = 0; k < n; k++) {
    const float *p = props9 + 9 * k;
    sprintf(name, "m%d", k);
    new_material(name, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], 0, 0, 0, &m_list);
  }
  return _standard_field_advance->new_material_coefficients(g, m_list);
}

/* The standard field advance kernels are reached through the method table
 * (field_advance/field_advance.h:185-302); expose the ones on the hot path by name. */
void ref_advance_b(field_t *f, const grid_t *g, float frac) { _standard_field_advance->advance_b(f, g, frac); }
void ref_advance_e(field_t *f, const material_coefficient_t *m, const grid_t *g) { _standard_field_advance->advance_e(f, m, g); }
void ref_clear_jf(field_t *f, const grid_t *g) { _standard_field_advance->clear_jf(f, g); }
void ref_synchronize_jf(field_t *f, const grid_t *g) { _standard_field_advance->synchronize_jf(f, g); }
void ref_energy_f(double *en, const field_t *f, const material_coefficient_t *m, const grid_t *g) {
  _standard_field_advance->energy_f(en, f, m, g);
}

/* divergence cleaning family and charge densities (field_advance.h:242-302) */
void ref_clear_rhof(field_t *f, const grid_t *g) { _standard_field_advance->clear_rhof(f, g); }
void ref_synchronize_rho(field_t *f, const grid_t *g) { _standard_field_advance->synchronize_rho(f, g); }
void ref_compute_rhob(field_t *f, const material_coefficient_t *m, const grid_t *g) { _standard_field_advance->compute_rhob(f, m, g); }
void ref_compute_curl_b(field_t *f, const material_coefficient_t *m, const grid_t *g) { _standard_field_advance->compute_curl_b(f, m, g); }
double ref_synchronize_tang_e_norm_b(field_t *f, const grid_t *g) { return _standard_field_advance->synchronize_tang_e_norm_b(f, g); }
void ref_compute_div_e_err(field_t *f, const material_coefficient_t *m, const grid_t *g) { _standard_field_advance->compute_div_e_err(f, m, g); }
double ref_compute_rms_div_e_err(field_t *f, const grid_t *g) { return _standard_field_advance->compute_rms_div_e_err(f, g); }
void ref_clean_div_e(field_t *f, const material_coefficient_t *m, const grid_t *g) { _standard_field_advance->clean_div_e(f, m, g); }
void ref_compute_div_b_err(field_t *f, const grid_t *g) { _standard_field_advance->compute_div_b_err(f, g); }
double ref_compute_rms_div_b_err(field_t *f, const grid_t *g) { return _standard_field_advance->compute_rms_div_b_err(f, g); }
void ref_clean_div_b(field_t *f, const grid_t *g) { _standard_field_advance->clean_div_b(f, g); }

/* sort_p / boundary_p take a species_t; build one around caller-owned arrays.  sort_p may
 * replace sp->p (out-of-place variant frees and mallocs, sort_p.c:69-77), so the species owns
 * reference-allocated copies and results are copied back out. */
species_t *ref_new_species(float q_m, int max_np, int max_nm, int sort_out_of_place) {
  species_t *sp_list = NULL;
  return new_species("s", q_m, max_np, max_nm, 1, sort_out_of_place, &sp_list);
}
particle_t *ref_species_p(species_t *sp) { return sp->p; }
particle_mover_t *ref_species_pm(species_t *sp) { return sp->pm; }
int *ref_species_partition(species_t *sp) { return sp->partition; }
void ref_species_set_counts(species_t *sp, int np, int nm) { sp->np = np; sp->nm = nm; }
int ref_species_np(species_t *sp) { return sp->np; }
int ref_species_nm(species_t *sp) { return sp->nm; }

/* struct layout probe for include/vpic_hip.h static asserts */
void ref_layout(int *out) {
  int k = 0;
  out[k++] = sizeof(particle_t);  out[k++] = sizeof(particle_mover_t); out[k++] = sizeof(particle_injector_t);
  out[k++] = sizeof(interpolator_t); out[k++] = sizeof(accumulator_t); out[k++] = sizeof(field_t);
  out[k++] = sizeof(grid_t); out[k++] = sizeof(species_t); out[k++] = sizeof(material_coefficient_t*);
  out[k++] = offsetof(grid_t, dt); out[k++] = offsetof(grid_t, x0); out[k++] = offsetof(grid_t, dx);
  out[k++] = offsetof(grid_t, rdx); out[k++] = offsetof(grid_t, nx); out[k++] = offsetof(grid_t, bc);
  out[k++] = offsetof(grid_t, range); out[k++] = offsetof(grid_t, neighbor); out[k++] = offsetof(grid_t, rangel);
  out[k++] = offsetof(grid_t, rangeh); out[k++] = offsetof(grid_t, nb); out[k++] = offsetof(grid_t, boundary);
  out[k++] = offsetof(species_t, np); out[k++] = offsetof(species_t, p); out[k++] = offsetof(species_t, nm);
  out[k++] = offsetof(species_t, pm); out[k++] = offsetof(species_t, q_m); out[k++] = offsetof(species_t, sort_interval);
  out[k++] = offsetof(species_t, partition); out[k++] = offsetof(species_t, next); out[k++] = offsetof(species_t, name);
}

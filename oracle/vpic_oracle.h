/* vpic_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the VPIC inner loop of pdlfs/old-vpic, written
 * from the reference's algorithm (each function cites the reference file:line it follows; paths
 * are relative to the reference's src/).  It is the checker the HIP engine is compared with.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning: the reference ships no golden vectors for this path, so this oracle is
 * pinned against the reference ITSELF, compiled from /root/reference by oracle/Makefile (target
 * `ref` -> oracle/_ref/libvpic_ref.so) and driven by oracle/gen_golden.py; the vectors it wrote
 * are committed under tests/golden/ and tests/test_oracle_golden.py checks this file against them
 * bit for bit.
 *
 * Struct layouts are the reference's (sizes/offsets asserted in vpic_oracle.c):
 *   particle 48 B   species_advance/species_advance.h:28-34
 *   mover 16 B      species_advance/species_advance.h:39-42
 *   injector 48 B   species_advance/species_advance.h:48-55
 *   interpolator 80 sf_interface/sf_interface.h:45-58
 *   accumulator 48  sf_interface/sf_interface.h:68-77
 *   field 80 B      field_advance/field_advance.h:159-171
 *   material coefficient 64 B  field_advance/standard/sfa_private.h:24-32
 */
#ifndef VPIC_ORACLE_H
#define VPIC_ORACLE_H
#include <stdint.h>

typedef struct { float dx, dy, dz; int32_t i; float ux, uy, uz, q; int64_t tag, tag2; } orc_particle_t;
typedef struct { float dispx, dispy, dispz; int32_t i; } orc_mover_t;
typedef struct { float dx, dy, dz; int32_t i; float ux, uy, uz, q; float dispx, dispy, dispz; int32_t sp_id; } orc_injector_t;
typedef struct {
  float ex, dexdy, dexdz, d2exdydz;
  float ey, deydz, deydx, d2eydzdx;
  float ez, dezdx, dezdy, d2ezdxdy;
  float cbx, dcbxdx, cby, dcbydy, cbz, dcbzdz;
  float _pad[2];
} orc_interpolator_t;
typedef struct { float jx[4], jy[4], jz[4]; } orc_accumulator_t;
typedef struct {
  float ex, ey, ez, div_e_err;
  float cbx, cby, cbz, div_b_err;
  float tcax, tcay, tcaz, rhob;
  float jfx, jfy, jfz, rhof;
  uint16_t ematx, ematy, ematz, nmat;
  uint16_t fmatx, fmaty, fmatz, cmat;
} orc_field_t;
typedef struct {
  float decayx, drivex, decayy, drivey, decayz, drivez;
  float rmux, rmuy, rmuz, nonconductive, epsx, epsy, epsz, pad[3];
} orc_material_coefficient_t;

/* sf_interface/sf_interface.h:28-38 (64 B) */
typedef struct {
  float jx, jy, jz, rho;
  float px, py, pz, ke;
  float txx, tyy, tzz;
  float tyz, tzx, txy;
  float _pad[2];
} orc_hydro_t;

/* Field boundary codes of a face that is not shared with a domain (grid/grid.h:56-66). */
enum { ORC_PEC_FIELDS = -1, ORC_SYMMETRIC_FIELDS = -2, ORC_PMC_FIELDS = -3, ORC_ABSORB_FIELDS = -4 };
/* Particle boundary codes (grid/grid.h:68-69). */
enum { ORC_REFLECT_PARTICLES = -1, ORC_ABSORB_PARTICLES = -2 };

/* A single rectangular domain.  Instead of the reference's 6*nv neighbor table (grid/ops.c:74-97,
 * 135-182, 199-231) the faces are described one code each, which generates exactly the tables
 * that size_grid + join_grid + set_fbc + set_pbc produce for box decks:
 *   fbc[f] >= 0 : field face shared with domain fbc[f] (== rank: periodic onto itself)
 *   fbc[f] <  0 : local field boundary code
 *   pbc[f] >= 0 : particles crossing go to domain pbc[f] (== rank: wrap locally)
 *   pbc[f] <  0 : reflect / absorb
 * Faces are ordered -x,-y,-z,+x,+y,+z like neighbor[6v+f].                                   */
typedef struct {
  float dt, cvac, eps0, damp;
  float dx, dy, dz, rdx, rdy, rdz;
  int   nx, ny, nz;
  int   fbc[6], pbc[6];
  int   rank;
} orc_grid_t;

#ifdef __cplusplus
extern "C" {
#endif

int  orc_nv(const orc_grid_t *g);                                        /* (nx+2)(ny+2)(nz+2) */
int  orc_accumulator_stride(const orc_grid_t *g);                        /* POW2_CEIL(nv,2) */

void orc_load_interpolator(orc_interpolator_t *fi, const orc_field_t *f, const orc_grid_t *g);
void orc_clear_accumulators(orc_accumulator_t *a, const orc_grid_t *g, int n_pipeline);
void orc_reduce_accumulators(orc_accumulator_t *a, const orc_grid_t *g, int n_pipeline);
void orc_unload_accumulator(orc_field_t *f, const orc_accumulator_t *a, const orc_grid_t *g);

int  orc_move_p(orc_particle_t *p0, orc_mover_t *pm, orc_accumulator_t *a0, const orc_grid_t *g);
/* n_pipeline >= 1 reproduces the reference's per-pipeline particle split, private accumulators
 * and mover segments (a0 must hold 1+n_pipeline copies); n_pipeline == 0 is the plain
 * sequential loop into one accumulator. */
int  orc_advance_p(orc_particle_t *p0, int np, float q_m, orc_mover_t *pm, int max_nm,
                   orc_accumulator_t *a0, const orc_interpolator_t *f0, const orc_grid_t *g,
                   int n_pipeline);

/* species_advance/standard/center_p.cxx, uncenter_p.cxx: half E kick + half Boris rotation forward
 * (center) or their inverse (uncenter); momenta only. */
void orc_center_p(orc_particle_t *p0, int np, float q_m, const orc_interpolator_t *f0, const orc_grid_t *g);
void orc_uncenter_p(orc_particle_t *p0, int np, float q_m, const orc_interpolator_t *f0, const orc_grid_t *g);

void orc_sort_p(orc_particle_t *p, int np, int *partition, const orc_grid_t *g, int out_of_place);

double orc_energy_p(const orc_particle_t *p0, int np, float q_m, const orc_interpolator_t *f0,
                    const orc_grid_t *g);
void   orc_energy_f(double *en6, const orc_field_t *f, const orc_material_coefficient_t *m,
                    const orc_grid_t *g);

void orc_vacuum_coefficients(orc_material_coefficient_t *m);             /* eps=mu=1, sigma=0 */
void orc_material_coefficients(orc_material_coefficient_t *mc, const float *props9, float dt, float eps0);   /* sfa.c:145-177 */
void orc_clear_jf(orc_field_t *f, const orc_grid_t *g);
void orc_advance_b(orc_field_t *f, const orc_grid_t *g, float frac);
/* advance_e = local ghosts + self-periodic ghost copy + all E updates + local_adjust_tang_e.
 * Faces shared with ANOTHER domain must have had their ghosts filled by the caller first
 * (orc_pack_tang_b on the neighbour, orc_unpack_tang_b here).                                */
void orc_advance_e(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g);
void orc_local_ghost_tang_b(orc_field_t *f, const orc_grid_t *g);
void orc_local_adjust_tang_e(orc_field_t *f, const orc_grid_t *g);
void orc_local_adjust_norm_b(orc_field_t *f, const orc_grid_t *g);
void orc_local_adjust_jf(orc_field_t *f, const orc_grid_t *g);
/* synchronize_jf for faces shared with this same domain (periodic onto itself) + local_adjust_jf;
 * faces shared with other domains go through orc_pack_jf / orc_unpack_jf, axis by axis.        */
void orc_synchronize_jf_local(orc_field_t *f, const orc_grid_t *g);

/* Face messages (without the leading cell-size float of the reference: uniform meshes only).
 * dir = 0..5 is the direction of travel (-x,-y,-z,+x,+y,+z).  Return = number of floats.     */
int  orc_tang_b_count(const orc_grid_t *g, int dir);
int  orc_pack_tang_b(float *buf, const orc_field_t *f, const orc_grid_t *g, int dir);
int  orc_unpack_tang_b(orc_field_t *f, const float *buf, const orc_grid_t *g, int dir);
int  orc_pack_jf(float *buf, const orc_field_t *f, const orc_grid_t *g, int dir);
int  orc_unpack_jf(orc_field_t *f, const float *buf, const orc_grid_t *g, int dir);

/* boundary_p split at the message boundary.  pack: classify movers (processed in reverse, holes
 * back-filled), absorbed particles go to rhob, emigrants are appended to out[face] (capacity
 * cap each); returns new np, writes counts to ns[6].  remote_nx/ny/nz[face] = the receiving
 * domain's cell counts (to form its local voxel index).                                       */
void orc_accumulate_rhob(orc_field_t *f, const orc_particle_t *p, const orc_grid_t *g);
int  orc_boundary_p_pack(orc_particle_t *p0, int np, const orc_mover_t *pm, int nm, int sp_id,
                         orc_field_t *f, const orc_grid_t *g,
                         orc_injector_t *out[6], int ns[6], int cap);
/* inject: append n injectors (processed in reverse) and finish their moves; returns new np and
 * writes the new mover count to *nm.                                                          */
int  orc_boundary_p_inject(orc_particle_t *p0, int np, orc_mover_t *pm, int *nm,
                           const orc_injector_t *in, int n, orc_accumulator_t *a0,
                           const orc_grid_t *g);

/* boundary/maxwellian_reflux.c:116-175 with the handler's three random draws passed in (uniform, normal, normal) */
void orc_maxwellian_reflux(const float draws[3], const orc_particle_t *r, const orc_mover_t *pm,
                           const orc_grid_t *g, float ut_para, float ut_perp, int face, int sp_id,
                           orc_injector_t *pi);

/* emitter/child-langmuir.c:15-120 with the model's six draws per emitted particle passed in (see vpic_oracle.c) */
int orc_child_langmuir(orc_particle_t *p0, int np, int max_np, orc_mover_t *pm0, int *nm_io, int max_nm,
                       const int *component, int n_component, int n_emit_per_face, float ut_perp, float ut_para,
                       float q_m, const orc_interpolator_t *fi, orc_field_t *f, orc_accumulator_t *a,
                       const orc_grid_t *g, const double *draws);

/* Divergence cleaning family and charge densities.  The *_local functions
 * handle local boundary faces and faces this domain shares with itself (periodic onto itself). */
void orc_clear_rhof(orc_field_t *f, const orc_grid_t *g);
void orc_accumulate_rho_p(orc_field_t *f, const orc_particle_t *p, int n, const orc_grid_t *g);
int  orc_rho_count(const orc_grid_t *g, int dir);
int  orc_pack_rho(float *buf, const orc_field_t *f, const orc_grid_t *g, int dir);
int  orc_unpack_rho(orc_field_t *f, const float *buf, const orc_grid_t *g, int dir);
void orc_synchronize_rho_local(orc_field_t *f, const orc_grid_t *g);
void orc_compute_div_e_err(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g);
void orc_compute_rhob(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g);
void orc_rms_div_e_err_local(double *local2, const orc_field_t *f, const orc_grid_t *g);
void orc_clean_div_e(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g);
void orc_compute_div_b_err(orc_field_t *f, const orc_grid_t *g);
void orc_rms_div_b_err_local(double *local2, const orc_field_t *f, const orc_grid_t *g);
void orc_clean_div_b(orc_field_t *f, const orc_grid_t *g);
void orc_compute_curl_b(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g);
double orc_synchronize_tang_e_norm_b_local(orc_field_t *f, const orc_grid_t *g);

/* face messages of the family for faces shared with another domain; kind 0 normal E (-> ghost),
 * 1 div_b_err (-> ghost), 2 tang E + norm B (-> averaged on the shared plane; returns the squared
 * difference sum).  Before orc_compute_div_e_err / orc_compute_rhob the kind-0 ghosts of such faces
 * must be in place, before orc_clean_div_b the kind-1 ghosts, before orc_compute_curl_b the tang_b
 * ghosts. */
int    orc_msg_count(const orc_grid_t *g, int kind, int dir);
int    orc_pack_msg(float *buf, const orc_field_t *f, const orc_grid_t *g, int kind, int dir);
double orc_unpack_msg(orc_field_t *f, const float *buf, const orc_grid_t *g, int kind, int dir);
void   orc_local_adjust_tang_e_norm_b(orc_field_t *f, const orc_grid_t *g);
double orc_synchronize_tang_e_norm_b_self(orc_field_t *f, const orc_grid_t *g, int axis);
void   orc_local_adjust_rho(orc_field_t *f, const orc_grid_t *g);
void   orc_synchronize_rho_self(orc_field_t *f, const orc_grid_t *g, int axis);

/* Hydro moments */
void orc_clear_hydro(orc_hydro_t *h, const orc_grid_t *g);
void orc_accumulate_hydro_p(orc_hydro_t *h, const orc_particle_t *p, int n, float q_m,
                            const orc_interpolator_t *f0, const orc_grid_t *g);
void orc_local_adjust_hydro(orc_hydro_t *h, const orc_grid_t *g);
int  orc_hydro_count(const orc_grid_t *g, int dir);
int  orc_pack_hydro(float *buf, const orc_hydro_t *h, const orc_grid_t *g, int dir);
int  orc_unpack_hydro(orc_hydro_t *h, const float *buf, const orc_grid_t *g, int dir);
void orc_synchronize_hydro_local(orc_hydro_t *h, const orc_grid_t *g);

#ifdef __cplusplus
}
#endif
#endif

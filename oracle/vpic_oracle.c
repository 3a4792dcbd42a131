/* vpic_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See vpic_oracle.h.
 *
 * Plain-C restatement of the reference's scalar pipelines.  Operation order and parenthesisation
 * follow the reference so that, compiled like the reference (-O2 -ffp-contract=off, SSE scalar
 * fp32), results are bit-identical to oracle/_ref on the same inputs.  Citations: file:line under
 * the reference's src/.
 */
#include "vpic_oracle.h"
#include <math.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define STATIC_ASSERT(c, n) typedef char orc_static_assert_##n[(c) ? 1 : -1]
STATIC_ASSERT(sizeof(orc_particle_t) == 48, particle);
STATIC_ASSERT(sizeof(orc_mover_t) == 16, mover);
STATIC_ASSERT(sizeof(orc_injector_t) == 48, injector);
STATIC_ASSERT(sizeof(orc_interpolator_t) == 80, interpolator);
STATIC_ASSERT(sizeof(orc_accumulator_t) == 48, accumulator);
STATIC_ASSERT(sizeof(orc_field_t) == 80, field);
STATIC_ASSERT(sizeof(orc_material_coefficient_t) == 64, matcoef);

#define DIE(msg) do { fprintf(stderr, "vpic_oracle: %s (%s:%d)\n", msg, __FILE__, __LINE__); exit(1); } while (0)

/* util/util_base.h:158-159 INDEX_FORTRAN_3 over (0:nx+1,0:ny+1,0:nz+1) */
#define VOXEL(x, y, z) ((x) + (nx + 2) * ((y) + (ny + 2) * (z)))

int orc_nv(const orc_grid_t *g) { return (g->nx + 2) * (g->ny + 2) * (g->nz + 2); }
/* util/util_base.h:113 POW2_CEIL(nv,2); sf_interface/sf_interface.c:66-72 */
int orc_accumulator_stride(const orc_grid_t *g) { return (orc_nv(g) + 1) & ~1; }

/* util/util_base.h:141-147 DISTRIBUTE */
static void distribute(int N, int b, int p, int P, int *i, int *n) {
  double t = (double)(N / b) / (double)P;
  int _i = b * (int)(t * (double)p + 0.5);
  *n = (p == P) ? (N % b) : (b * (int)(t * (double)(p + 1) + 0.5) - _i);
  *i = _i;
}

/* ------------------------------------------------------------------------------------------
 * sf_interface/load_interpolator.cxx:72-140 (interior voxels 1..n)                           */
void orc_load_interpolator(orc_interpolator_t *fi, const orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const float fourth = 0.25f, half = 0.5f;
  for (int z = 1; z <= nz; z++)
    for (int y = 1; y <= ny; y++)
      for (int x = 1; x <= nx; x++) {
        orc_interpolator_t *pi = &fi[VOXEL(x, y, z)];
        const orc_field_t *pf0 = &f[VOXEL(x, y, z)];
        const orc_field_t *pfx = &f[VOXEL(x + 1, y, z)], *pfy = &f[VOXEL(x, y + 1, z)];
        const orc_field_t *pfz = &f[VOXEL(x, y, z + 1)], *pfyz = &f[VOXEL(x, y + 1, z + 1)];
        const orc_field_t *pfzx = &f[VOXEL(x + 1, y, z + 1)], *pfxy = &f[VOXEL(x + 1, y + 1, z)];
        float w0, w1, w2, w3;
        w0 = pf0->ex; w1 = pfy->ex; w2 = pfz->ex; w3 = pfyz->ex;
        pi->ex       = fourth * ((w3 + w0) + (w1 + w2));
        pi->dexdy    = fourth * ((w3 - w0) + (w1 - w2));
        pi->dexdz    = fourth * ((w3 - w0) - (w1 - w2));
        pi->d2exdydz = fourth * ((w3 + w0) - (w1 + w2));
        w0 = pf0->ey; w1 = pfz->ey; w2 = pfx->ey; w3 = pfzx->ey;
        pi->ey       = fourth * ((w3 + w0) + (w1 + w2));
        pi->deydz    = fourth * ((w3 - w0) + (w1 - w2));
        pi->deydx    = fourth * ((w3 - w0) - (w1 - w2));
        pi->d2eydzdx = fourth * ((w3 + w0) - (w1 + w2));
        w0 = pf0->ez; w1 = pfx->ez; w2 = pfy->ez; w3 = pfxy->ez;
        pi->ez       = fourth * ((w3 + w0) + (w1 + w2));
        pi->dezdx    = fourth * ((w3 - w0) + (w1 - w2));
        pi->dezdy    = fourth * ((w3 - w0) - (w1 - w2));
        pi->d2ezdxdy = fourth * ((w3 + w0) - (w1 + w2));
        w0 = pf0->cbx; w1 = pfx->cbx; pi->cbx = half * (w1 + w0); pi->dcbxdx = half * (w1 - w0);
        w0 = pf0->cby; w1 = pfy->cby; pi->cby = half * (w1 + w0); pi->dcbydy = half * (w1 - w0);
        w0 = pf0->cbz; w1 = pfz->cbz; pi->cbz = half * (w1 + w0); pi->dcbzdz = half * (w1 - w0);
      }
}

/* sf_interface/clear_accumulators.c:26-49 */
void orc_clear_accumulators(orc_accumulator_t *a, const orc_grid_t *g, int n_pipeline) {
  memset(a, 0, sizeof(*a) * (size_t)(1 + n_pipeline) * (size_t)orc_accumulator_stride(g));
}

/* sf_interface/reduce_accumulators.cxx:37-55 (interior voxels; copies summed in order 1..na-1) */
void orc_reduce_accumulators(orc_accumulator_t *a, const orc_grid_t *g, int n_pipeline) {
  const int nx = g->nx, ny = g->ny, nz = g->nz, na = 1 + n_pipeline;
  const int stride = orc_accumulator_stride(g);
  for (int z = 1; z <= nz; z++)
    for (int y = 1; y <= ny; y++)
      for (int x = 1; x <= nx; x++) {
        float *da = (float *)&a[VOXEL(x, y, z)];
        for (int n = 1; n < na; n++) {
          const float *sa = da + 12 * (size_t)stride * n;
          for (int k = 0; k < 12; k++) da[k] += sa[k];
        }
      }
}

/* sf_interface/unload_accumulator.cxx:30-52 (voxels 1..n+1) */
void orc_unload_accumulator(orc_field_t *f, const orc_accumulator_t *a, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const float cx = 0.25 * g->rdy * g->rdz / g->dt;
  const float cy = 0.25 * g->rdz * g->rdx / g->dt;
  const float cz = 0.25 * g->rdx * g->rdy / g->dt;
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_accumulator_t *a0 = &a[VOXEL(x, y, z)];
        const orc_accumulator_t *ax = &a[VOXEL(x - 1, y, z)], *ay = &a[VOXEL(x, y - 1, z)];
        const orc_accumulator_t *az = &a[VOXEL(x, y, z - 1)], *ayz = &a[VOXEL(x, y - 1, z - 1)];
        const orc_accumulator_t *azx = &a[VOXEL(x - 1, y, z - 1)], *axy = &a[VOXEL(x - 1, y - 1, z)];
        f0->jfx += cx * (a0->jx[0] + ay->jx[1] + az->jx[2] + ayz->jx[3]);
        f0->jfy += cy * (a0->jy[0] + az->jy[1] + ax->jy[2] + azx->jy[3]);
        f0->jfz += cz * (a0->jz[0] + ax->jz[1] + ay->jz[2] + axy->jz[3]);
      }
}

/* ------------------------------------------------------------------------------------------
 * species_advance/standard/move_p.c:20-136.  The neighbor[] lookup of :123 is replaced by the
 * per-face codes of orc_grid_t (same outcome as grid/ops.c:74-97 + join_grid/set_pbc tables). */
int orc_move_p(orc_particle_t *p0, orc_mover_t *pm, orc_accumulator_t *a0, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  float s_midx, s_midy, s_midz, s_dispx, s_dispy, s_dispz, s_dir[3];
  float v0, v1, v2, v3, v4, v5;
  int type;
  float *a;
  orc_particle_t *p = p0 + pm->i;

  for (;;) {
    s_midx = p->dx; s_midy = p->dy; s_midz = p->dz;
    s_dispx = pm->dispx; s_dispy = pm->dispy; s_dispz = pm->dispz;
    s_dir[0] = (s_dispx > 0) ? 1 : -1;
    s_dir[1] = (s_dispy > 0) ? 1 : -1;
    s_dir[2] = (s_dispz > 0) ? 1 : -1;

    v0 = (s_dispx == 0) ? 3.4e38 : (s_dir[0] - s_midx) / s_dispx;
    v1 = (s_dispy == 0) ? 3.4e38 : (s_dir[1] - s_midy) / s_dispy;
    v2 = (s_dispz == 0) ? 3.4e38 : (s_dir[2] - s_midz) / s_dispz;

    /**/         v3 = 2,  type = 3;
    if (v0 < v3) v3 = v0, type = 0;
    if (v1 < v3) v3 = v1, type = 1;
    if (v2 < v3) v3 = v2, type = 2;
    v3 *= 0.5;

    s_dispx *= v3; s_dispy *= v3; s_dispz *= v3;
    s_midx += s_dispx; s_midy += s_dispy; s_midz += s_dispz;

    v5 = p->q * s_dispx * s_dispy * s_dispz * (1. / 3.);
    a = (float *)(a0 + p->i);
#   define accumulate_j(X, Y, Z)                                                  \
    v4 = p->q * s_disp##X;                                                        \
    v1 = v4 * s_mid##Y;                                                           \
    v0 = v4 - v1;                                                                 \
    v1 += v4;                                                                     \
    v4 = 1 + s_mid##Z;                                                            \
    v2 = v0 * v4;                                                                 \
    v3 = v1 * v4;                                                                 \
    v4 = 1 - s_mid##Z;                                                            \
    v0 *= v4;                                                                     \
    v1 *= v4;                                                                     \
    v0 += v5;                                                                     \
    v1 -= v5;                                                                     \
    v2 -= v5;                                                                     \
    v3 += v5;                                                                     \
    a[0] += v0; a[1] += v1; a[2] += v2; a[3] += v3
    accumulate_j(x, y, z); a += 4;
    accumulate_j(y, z, x); a += 4;
    accumulate_j(z, x, y);
#   undef accumulate_j

    pm->dispx -= s_dispx; pm->dispy -= s_dispy; pm->dispz -= s_dispz;
    p->dx += s_dispx + s_dispx; p->dy += s_dispy + s_dispy; p->dz += s_dispz + s_dispz;

    if (type == 3) return 0;

    v0 = s_dir[type];
    {
      /* which cell coordinate and which face */
      int i = p->i, c[3], n[3] = {nx, ny, nz}, stride[3] = {1, nx + 2, (nx + 2) * (ny + 2)};
      int face = ((v0 > 0) ? 3 : 0) + type;
      c[0] = i % (nx + 2); c[1] = (i / (nx + 2)) % (ny + 2); c[2] = i / ((nx + 2) * (ny + 2));
      int at_edge = (v0 > 0) ? (c[type] == n[type]) : (c[type] == 1);
      if (at_edge && g->pbc[face] != g->rank) {              /* neighbor outside [rangel,rangeh] */
        (&(p->dx))[type] = v0;
        if (g->pbc[face] != ORC_REFLECT_PARTICLES) return 1;
        (&(p->ux))[type] = -(&(p->ux))[type];
        (&(pm->dispx))[type] = -(&(pm->dispx))[type];
      } else {
        if (at_edge) p->i = i + ((v0 > 0) ? -(n[type] - 1) : (n[type] - 1)) * stride[type]; /* wrap */
        else         p->i = i + ((v0 > 0) ? stride[type] : -stride[type]);
        (&(p->dx))[type] = -v0;
      }
    }
  }
  return 0;
}

/* species_advance/standard/advance_p.cxx:68-177, one pipeline's particle range */
static void advance_p_range(orc_particle_t *p0, int first, int n, orc_mover_t *pm, int max_nm,
                            int *nm_out, int *n_ignored, orc_accumulator_t *a0,
                            const orc_interpolator_t *f0, const orc_grid_t *g,
                            float qdt_2mc, float cdt_dx, float cdt_dy, float cdt_dz) {
  const float one = 1., one_third = 1. / 3., two_fifteenths = 2. / 15.;
  float dx, dy, dz, ux, uy, uz, q, hax, hay, haz, cbx, cby, cbz, v0, v1, v2, v3, v4, v5;
  int ii, nm = 0, itmp = 0;
  orc_particle_t *p = p0 + first;
  const orc_interpolator_t *f;
  float *a;
  orc_mover_t local_pm[1];

  for (; n; n--, p++) {
    dx = p->dx; dy = p->dy; dz = p->dz; ii = p->i;
    f = f0 + ii;
    hax = qdt_2mc * ((f->ex + dy * f->dexdy) + dz * (f->dexdz + dy * f->d2exdydz));
    hay = qdt_2mc * ((f->ey + dz * f->deydz) + dx * (f->deydx + dz * f->d2eydzdx));
    haz = qdt_2mc * ((f->ez + dx * f->dezdx) + dy * (f->dezdy + dx * f->d2ezdxdy));
    cbx = f->cbx + dx * f->dcbxdx;
    cby = f->cby + dy * f->dcbydy;
    cbz = f->cbz + dz * f->dcbzdz;
    ux = p->ux; uy = p->uy; uz = p->uz; q = p->q;
    ux += hax; uy += hay; uz += haz;
    v0 = qdt_2mc / sqrtf(one + (ux * ux + (uy * uy + uz * uz)));
    v1 = cbx * cbx + (cby * cby + cbz * cbz);
    v2 = (v0 * v0) * v1;
    v3 = v0 * (one + v2 * (one_third + v2 * two_fifteenths));
    v4 = v3 / (one + v1 * (v3 * v3));
    v4 += v4;
    v0 = ux + v3 * (uy * cbz - uz * cby);
    v1 = uy + v3 * (uz * cbx - ux * cbz);
    v2 = uz + v3 * (ux * cby - uy * cbx);
    ux += v4 * (v1 * cbz - v2 * cby);
    uy += v4 * (v2 * cbx - v0 * cbz);
    uz += v4 * (v0 * cby - v1 * cbx);
    ux += hax; uy += hay; uz += haz;
    p->ux = ux; p->uy = uy; p->uz = uz;
    v0 = one / sqrtf(one + (ux * ux + (uy * uy + uz * uz)));
    ux *= cdt_dx; uy *= cdt_dy; uz *= cdt_dz;
    ux *= v0; uy *= v0; uz *= v0;
    v0 = dx + ux; v1 = dy + uy; v2 = dz + uz;
    v3 = v0 + ux; v4 = v1 + uy; v5 = v2 + uz;

    if (v3 <= one && v4 <= one && v5 <= one && -v3 <= one && -v4 <= one && -v5 <= one) {
      p->dx = v3; p->dy = v4; p->dz = v5;
      dx = v0; dy = v1; dz = v2;
      v5 = q * ux * uy * uz * one_third;
      a = (float *)(a0 + ii);
#     define ACCUMULATE_J(X, Y, Z, offset)                                        \
      v4 = q * u##X;                                                              \
      v1 = v4 * d##Y;                                                             \
      v0 = v4 - v1;                                                               \
      v1 += v4;                                                                   \
      v4 = one + d##Z;                                                            \
      v2 = v0 * v4;                                                               \
      v3 = v1 * v4;                                                               \
      v4 = one - d##Z;                                                            \
      v0 *= v4;                                                                   \
      v1 *= v4;                                                                   \
      v0 += v5;                                                                   \
      v1 -= v5;                                                                   \
      v2 -= v5;                                                                   \
      v3 += v5;                                                                   \
      a[offset + 0] += v0; a[offset + 1] += v1; a[offset + 2] += v2; a[offset + 3] += v3
      ACCUMULATE_J(x, y, z, 0);
      ACCUMULATE_J(y, z, x, 4);
      ACCUMULATE_J(z, x, y, 8);
#     undef ACCUMULATE_J
    } else {
      local_pm->dispx = ux; local_pm->dispy = uy; local_pm->dispz = uz;
      local_pm->i = (int32_t)(p - p0);
      if (orc_move_p(p0, local_pm, a0, g)) {
        if (nm < max_nm) pm[nm++] = local_pm[0];
        else itmp++;
      }
    }
  }
  *nm_out = nm;
  *n_ignored = itmp;
}

/* species_advance/standard/advance_p.cxx:399-472 (+ per-pipeline setup :41-64) */
int orc_advance_p(orc_particle_t *p0, int np, float q_m, orc_mover_t *pm, int max_nm,
                  orc_accumulator_t *a0, const orc_interpolator_t *f0, const orc_grid_t *g,
                  int n_pipeline) {
  if (!p0 || np < 0 || !pm || max_nm < 0 || !a0 || !f0 || !g) DIE("bad advance_p argument");
  const float qdt_2mc = 0.5 * q_m * g->dt / g->cvac;
  const float cdt_dx = g->cvac * g->dt * g->rdx;
  const float cdt_dy = g->cvac * g->dt * g->rdy;
  const float cdt_dz = g->cvac * g->dt * g->rdz;
  int nm = 0, ign;

  if (n_pipeline <= 0) {
    advance_p_range(p0, 0, np, pm, max_nm, &nm, &ign, a0, f0, g, qdt_2mc, cdt_dx, cdt_dy, cdt_dz);
    return nm;
  }
  const int stride = orc_accumulator_stride(g);
  for (int rank = 0; rank <= n_pipeline; rank++) {
    int first, n, mfirst, mmax, seg_nm;
    distribute(np, 16, rank, n_pipeline, &first, &n);
    mmax = max_nm - (np & 15);
    if (mmax < 0) mmax = 0;
    distribute(mmax, 8, rank, n_pipeline, &mfirst, &mmax);
    if (rank == n_pipeline) mmax = max_nm - mfirst;
    orc_accumulator_t *a = a0;
    if (rank != n_pipeline) a += (size_t)(1 + rank) * stride;
    advance_p_range(p0, first, n, pm + mfirst, mmax, &seg_nm, &ign, a, f0, g,
                    qdt_2mc, cdt_dx, cdt_dy, cdt_dz);
    if (ign) fprintf(stderr, "vpic_oracle: pipeline %d ran out of storage for %d movers\n", rank, ign);
    if (nm != mfirst) memmove(pm + nm, pm + mfirst, sizeof(*pm) * (size_t)seg_nm);
    nm += seg_nm;
  }
  return nm;
}

/* ------------------------------------------------------------------------------------------
 * species_advance/standard/center_p.cxx:9-71 (uncenter=0) and uncenter_p.cxx:5-71 (uncenter=1);
 * host constants uncenter_p.cxx:154-177 (qdt_2mc = 0.5*q_m*dt/cvac)                            */
static void center_uncenter(orc_particle_t *p, int np, float q_m, const orc_interpolator_t *f0,
                            const orc_grid_t *g, int uncenter) {
  const float args_qdt_2mc = 0.5 * q_m * g->dt / g->cvac;
  const float qdt_2mc = uncenter ? -args_qdt_2mc : args_qdt_2mc;
  const float qdt_4mc = uncenter ? -0.5 * args_qdt_2mc : 0.5 * args_qdt_2mc;
  const float one = 1., one_third = 1. / 3., two_fifteenths = 2. / 15.;
  for (; np; np--, p++) {
    const float dx = p->dx, dy = p->dy, dz = p->dz;
    const orc_interpolator_t *f = f0 + p->i;
    const float hax = qdt_2mc * ((f->ex + dy * f->dexdy) + dz * (f->dexdz + dy * f->d2exdydz));
    const float hay = qdt_2mc * ((f->ey + dz * f->deydz) + dx * (f->deydx + dz * f->d2eydzdx));
    const float haz = qdt_2mc * ((f->ez + dx * f->dezdx) + dy * (f->dezdy + dx * f->d2ezdxdy));
    const float cbx = f->cbx + dx * f->dcbxdx, cby = f->cby + dy * f->dcbydy, cbz = f->cbz + dz * f->dcbzdz;
    float ux = p->ux, uy = p->uy, uz = p->uz, v0, v1, v2, v3, v4;
    if (!uncenter) { ux += hax; uy += hay; uz += haz; }
    v0 = qdt_4mc / (float)sqrt(one + (ux * ux + (uy * uy + uz * uz)));
    v1 = cbx * cbx + (cby * cby + cbz * cbz);
    v2 = (v0 * v0) * v1;
    v3 = v0 * (one + v2 * (one_third + v2 * two_fifteenths));
    v4 = v3 / (one + v1 * (v3 * v3));
    v4 += v4;
    v0 = ux + v3 * (uy * cbz - uz * cby);
    v1 = uy + v3 * (uz * cbx - ux * cbz);
    v2 = uz + v3 * (ux * cby - uy * cbx);
    ux += v4 * (v1 * cbz - v2 * cby);
    uy += v4 * (v2 * cbx - v0 * cbz);
    uz += v4 * (v0 * cby - v1 * cbx);
    if (uncenter) { ux += hax; uy += hay; uz += haz; }
    p->ux = ux; p->uy = uy; p->uz = uz;
  }
}
void orc_center_p(orc_particle_t *p0, int np, float q_m, const orc_interpolator_t *f0, const orc_grid_t *g) {
  center_uncenter(p0, np, q_m, f0, g, 0);
}
void orc_uncenter_p(orc_particle_t *p0, int np, float q_m, const orc_interpolator_t *f0, const orc_grid_t *g) {
  center_uncenter(p0, np, q_m, f0, g, 1);
}

/* ------------------------------------------------------------------------------------------
 * species_advance/standard/sort_p.c:16-102                                                   */
void orc_sort_p(orc_particle_t *p, int np, int *partition, const orc_grid_t *g, int out_of_place) {
  const int nc = orc_nv(g), nc1 = nc + 1;
  int i, j;
  if (np == 0) return;
  int *next = (int *)calloc((size_t)nc1, sizeof(int));
  for (i = 0; i < np; i++) next[p[i].i]++;
  j = 0;
  for (i = 0; i < nc1; i++) { partition[i] = j; j += next[i]; next[i] = partition[i]; }
  if (out_of_place) {
    orc_particle_t *new_p = (orc_particle_t *)malloc(sizeof(*p) * (size_t)np);
    for (i = 0; i < np; i++) new_p[next[p[i].i]++] = p[i];
    memcpy(p, new_p, sizeof(*p) * (size_t)np);
    free(new_p);
  } else {
    orc_particle_t save_p, *src, *dest;
    i = 0;
    while (i < nc) {
      if (next[i] >= partition[i + 1]) i++;
      else {
        src = &p[next[i]];
        for (;;) {
          dest = &p[next[src->i]++];
          if (src == dest) break;
          save_p = *dest; *dest = *src; *src = save_p;
        }
      }
    }
  }
  free(next);
}

/* ------------------------------------------------------------------------------------------
 * species_advance/standard/energy_p.cxx:31-47,124-157 (sequential; per-pipeline partial sums of
 * the reference differ from this only in the last bits of a double)                           */
double orc_energy_p(const orc_particle_t *p0, int np, float q_m, const orc_interpolator_t *f0,
                    const orc_grid_t *g) {
  const float qdt_2mc = 0.5 * q_m * g->dt / g->cvac, one = 1.;
  double en = 0;
  for (int n = 0; n < np; n++) {
    const orc_particle_t *p = p0 + n;
    const orc_interpolator_t *f = f0 + p->i;
    float dx = p->dx, dy = p->dy, dz = p->dz, v0, v1, v2;
    v0 = p->ux + qdt_2mc * ((f->ex + dy * f->dexdy) + dz * (f->dexdz + dy * f->d2exdydz));
    v1 = p->uy + qdt_2mc * ((f->ey + dz * f->deydz) + dx * (f->deydx + dz * f->d2eydzdx));
    v2 = p->uz + qdt_2mc * ((f->ez + dx * f->dezdx) + dy * (f->dezdy + dx * f->d2ezdxdy));
    v0 = v0 * v0 + v1 * v1 + v2 * v2;
    v0 /= (float)sqrt(one + v0) + one;
    en += (double)v0 * (double)p->q;
  }
  return (double)g->cvac * (double)g->cvac * en / (double)q_m;
}

/* field_advance/standard/energy_f.c:50-82,158-178 */
void orc_energy_f(double *global, const orc_field_t *f, const orc_material_coefficient_t *m,
                  const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  double en_ex = 0, en_ey = 0, en_ez = 0, en_bx = 0, en_by = 0, en_bz = 0;
  for (int z = 1; z <= nz; z++)
    for (int y = 1; y <= ny; y++)
      for (int x = 1; x <= nx; x++) {
        const orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x + 1, y, z)], *fy = &f[VOXEL(x, y + 1, z)];
        const orc_field_t *fz = &f[VOXEL(x, y, z + 1)], *fyz = &f[VOXEL(x, y + 1, z + 1)];
        const orc_field_t *fzx = &f[VOXEL(x + 1, y, z + 1)], *fxy = &f[VOXEL(x + 1, y + 1, z)];
        en_ex += 0.25 * (m[f0->ematx].epsx * f0->ex * f0->ex + m[fy->ematx].epsx * fy->ex * fy->ex +
                         m[fz->ematx].epsx * fz->ex * fz->ex + m[fyz->ematx].epsx * fyz->ex * fyz->ex);
        en_ey += 0.25 * (m[f0->ematy].epsy * f0->ey * f0->ey + m[fz->ematy].epsy * fz->ey * fz->ey +
                         m[fx->ematy].epsy * fx->ey * fx->ey + m[fzx->ematy].epsy * fzx->ey * fzx->ey);
        en_ez += 0.25 * (m[f0->ematz].epsz * f0->ez * f0->ez + m[fx->ematz].epsz * fx->ez * fx->ez +
                         m[fy->ematz].epsz * fy->ez * fy->ez + m[fxy->ematz].epsz * fxy->ez * fxy->ez);
        en_bx += 0.5 * (m[f0->fmatx].rmux * f0->cbx * f0->cbx + m[fx->fmatx].rmux * fx->cbx * fx->cbx);
        en_by += 0.5 * (m[f0->fmaty].rmuy * f0->cby * f0->cby + m[fy->fmaty].rmuy * fy->cby * fy->cby);
        en_bz += 0.5 * (m[f0->fmatz].rmuz * f0->cbz * f0->cbz + m[fz->fmatz].rmuz * fz->cbz * fz->cbz);
      }
  double v0 = 0.5 * g->eps0 * g->dx * g->dy * g->dz;
  global[0] = en_ex * v0; global[1] = en_ey * v0; global[2] = en_ez * v0;
  global[3] = en_bx * v0; global[4] = en_by * v0; global[5] = en_bz * v0;
}

/* field_advance/standard/sfa.c:145-177 for eps=mu=1, sigma=0 */
void orc_vacuum_coefficients(orc_material_coefficient_t *m) {
  memset(m, 0, sizeof(*m));
  m->decayx = m->decayy = m->decayz = 1;
  m->drivex = m->drivey = m->drivez = 1;
  m->rmux = m->rmuy = m->rmuz = 1;
  m->nonconductive = 1;
  m->epsx = m->epsy = m->epsz = 1;
}

/* field_advance/standard/sfa.c:145-177: the coefficient record of one material.  props9 = {eps xyz, mu xyz,
 * sigma xyz} as floats (material_t holds floats); the arithmetic is the reference's: double exp / sinh on
 * float operands, stored as float.  decay = exp(-sigma dt / (eps eps0)); drive = the exactly integrated
 * source weight, 1/eps without conductivity, 0 for a perfect conductor to numerical precision. */
void orc_material_coefficients(orc_material_coefficient_t *mc, const float *props9, float dt, float eps0) {
  const float *eps = props9, *mu = props9 + 3, *sigma = props9 + 6;
  float a[3], decay[3], drive[3];
  memset(mc, 0, sizeof(*mc));
  for (int k = 0; k < 3; k++) {
    a[k] = (sigma[k] * dt) / (eps[k] * eps0);
    decay[k] = exp(-a[k]);
    if (a[k] == 0) drive[k] = 1. / eps[k];
    else if (decay[k] == 0) drive[k] = 0;
    else drive[k] = 2. * exp(-0.5 * a[k]) * sinh(0.5 * a[k]) / (a[k] * eps[k]);
  }
  mc->decayx = decay[0]; mc->decayy = decay[1]; mc->decayz = decay[2];
  mc->drivex = drive[0]; mc->drivey = drive[1]; mc->drivez = drive[2];
  mc->rmux = 1. / mu[0]; mc->rmuy = 1. / mu[1]; mc->rmuz = 1. / mu[2];
  mc->nonconductive = (a[0] == 0 && a[1] == 0 && a[2] == 0) ? 1. : 0.;
  mc->epsx = eps[0]; mc->epsy = eps[1]; mc->epsz = eps[2];
}

/* field_advance/standard/sfa.c:188-211 */
void orc_clear_jf(orc_field_t *f, const orc_grid_t *g) {
  const int nv = orc_nv(g);
  for (int v = 0; v < nv; v++) f[v].jfx = f[v].jfy = f[v].jfz = 0;
}

/* ------------------------------------------------------------------------------------------
 * Plane boxes.  A component directed along axis `ca` lives, on a plane normal to `axis`:
 *   edge mesh (E, tca, jf): 1..n along its own axis, 1..n+1 along the others
 *   face mesh (cB)        : 1..n+1 along its own axis, 1..n along the others
 * (field_advance/field_advance.h:60-70; the *_EDGE_LOOP / *_FACE_LOOP macros of
 * field_advance/standard/local.c:26-46 and remote.c:17-41 are exactly these boxes.)          */
typedef struct { int lo[3], hi[3]; } box_t;
static box_t plane_box(const orc_grid_t *g, int axis, int plane, int ca, int edge_mesh) {
  const int n[3] = {g->nx, g->ny, g->nz};
  box_t b;
  for (int d = 0; d < 3; d++) {
    b.lo[d] = 1;
    b.hi[d] = (d == ca) ? (edge_mesh ? n[d] : n[d] + 1) : (edge_mesh ? n[d] + 1 : n[d]);
  }
  b.lo[axis] = b.hi[axis] = plane;
  return b;
}
#define BOX_LOOP(b) for (int z = (b).lo[2]; z <= (b).hi[2]; z++) \
                    for (int y = (b).lo[1]; y <= (b).hi[1]; y++) \
                    for (int x = (b).lo[0]; x <= (b).hi[0]; x++)

static int is_local_bc(int bc) { return bc < 0; }  /* local.c:72 `bc<0 || bc>nproc` */

/* float-index of a component inside orc_field_t */
enum { F_EX = 0, F_CBX = 4, F_TCAX = 8, F_JFX = 12 };
#define FC(f, v, base, comp) (((float *)&(f)[v])[(base) + (comp)])

/* field_advance/standard/local.c:50-122.  The absorbing case (local.c:84-108) is the 2nd-order
 * accurate 1st-order Higdon condition: the ghost cB relaxes towards the first interior value with
 * the two E differences of Faraday's law at the face added.                                     */
void orc_local_ghost_tang_b(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz, n[3] = {g->nx, g->ny, g->nz};
  const int stride[3] = {1, nx + 2, (nx + 2) * (ny + 2)};
  const float cdt_d[3] = {g->cvac * g->dt * g->rdx, g->cvac * g->dt * g->rdy, g->cvac * g->dt * g->rdz};
  const float higend = (nx > 1 || ny > 1 || nz > 1) ? 1.03527618 : 1.;
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc)) continue;
    int ghost = hi ? n[axis] + 1 : 0, off = hi ? -stride[axis] : stride[axis];
    if (bc == ORC_ABSORB_FIELDS) {
      const int aY = (axis + 1) % 3, aZ = (axis + 2) % 3;
      const int to_face = ((hi ? n[axis] + 1 : 1) - ghost) * stride[axis];   /* ghost voxel -> face voxel */
      float drive = cdt_d[axis] * higend, decay = (1 - drive) / (1 + drive), t1, t2;
      drive = 2 * drive / (1 + drive);
      box_t b = plane_box(g, axis, ghost, aY, 0);                 /* cbY over the Z-oriented... ZY_EDGE_LOOP box */
      BOX_LOOP(b) {
        int v = VOXEL(x, y, z), vf = v + to_face;
        t1 = cdt_d[axis] * (FC(f, vf + off, F_EX, aZ) - FC(f, vf, F_EX, aZ));
        t1 = hi ? -t1 : t1;
        t2 = FC(f, v + off + stride[aZ], F_EX, axis);
        t2 = cdt_d[aZ] * (t2 - FC(f, v + off, F_EX, axis));
        FC(f, v, F_CBX, aY) = decay * FC(f, v, F_CBX, aY) + drive * FC(f, v + off, F_CBX, aY) - t1 + t2;
      }
      b = plane_box(g, axis, ghost, aZ, 0);
      BOX_LOOP(b) {
        int v = VOXEL(x, y, z), vf = v + to_face;
        t1 = cdt_d[axis] * (FC(f, vf + off, F_EX, aY) - FC(f, vf, F_EX, aY));
        t1 = hi ? -t1 : t1;
        t2 = FC(f, v + off + stride[aY], F_EX, axis);
        t2 = cdt_d[aY] * (t2 - FC(f, v + off, F_EX, axis));
        FC(f, v, F_CBX, aZ) = decay * FC(f, v, F_CBX, aZ) + drive * FC(f, v + off, F_CBX, aZ) + t1 - t2;
      }
      continue;
    }
    float sign;
    if (bc == ORC_PEC_FIELDS) sign = 1;
    else if (bc == ORC_SYMMETRIC_FIELDS || bc == ORC_PMC_FIELDS) sign = -1;
    else DIE("Bad boundary condition encountered.");
    for (int t = 1; t <= 2; t++) {
      int ca = (axis + t) % 3;
      box_t b = plane_box(g, axis, ghost, ca, 0);
      BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_CBX, ca) = sign * FC(f, v + off, F_CBX, ca); }
    }
  }
}

/* field_advance/standard/local.c:224-264 */
void orc_local_adjust_tang_e(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc) || bc != ORC_PEC_FIELDS) continue;
    int plane = hi ? n[axis] + 1 : 1;
    for (int t = 1; t <= 2; t++) {
      int ca = (axis + t) % 3;
      box_t b = plane_box(g, axis, plane, ca, 1);
      BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_EX, ca) = 0; FC(f, v, F_TCAX, ca) = 0; }
    }
  }
}

/* field_advance/standard/local.c:266-296 */
void orc_local_adjust_norm_b(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc) || bc != ORC_SYMMETRIC_FIELDS) continue;
    box_t b = plane_box(g, axis, hi ? n[axis] + 1 : 1, axis, 0);
    BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_CBX, axis) = 0; }
  }
}

/* field_advance/standard/local.c:335-368 */
void orc_local_adjust_jf(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc)) continue;
    int plane = hi ? n[axis] + 1 : 1;
    for (int t = 1; t <= 2; t++) {
      int ca = (axis + t) % 3;
      box_t b = plane_box(g, axis, plane, ca, 1);
      if (bc == ORC_PEC_FIELDS) { BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_JFX, ca) = 0; } }
      else                      { BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_JFX, ca) *= 2.; } }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Face messages: field_advance/standard/remote.c:61-134 (tang_b) and :416-506 (jf).
 * dir = direction of travel.  Message order within a face follows the reference: for tang_b the
 * (axis+1) component then the (axis+2) component (remote.c:83-85); for jf likewise (remote.c:
 * 442-444).  Uniform meshes: the cell-size weights are rw=1, lw=0 (tang_b, remote.c:108-110) and
 * lw=rw=1 (jf, remote.c:452-457); the arithmetic keeps the reference's form.                   */
int orc_tang_b_count(const orc_grid_t *g, int dir) {
  const int n[3] = {g->nx, g->ny, g->nz};
  int a = dir % 3, nY = n[(a + 1) % 3], nZ = n[(a + 2) % 3];
  return nY * (nZ + 1) + nZ * (nY + 1);
}

int orc_pack_tang_b(float *buf, const orc_field_t *f, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, plane = dir < 3 ? 1 : n[axis], k = 0;
  for (int t = 1; t <= 2; t++) {
    int ca = (axis + t) % 3;
    box_t b = plane_box(g, axis, plane, ca, 0);
    BOX_LOOP(b) buf[k++] = FC(f, VOXEL(x, y, z), F_CBX, ca);
  }
  return k;
}

int orc_unpack_tang_b(orc_field_t *f, const float *buf, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  const int stride[3] = {1, nx + 2, (nx + 2) * (ny + 2)};
  int axis = dir % 3, ghost = dir < 3 ? n[axis] + 1 : 0, k = 0;
  int in = dir < 3 ? -stride[axis] : stride[axis];       /* f(x+i,y+j,z+k): one step along travel */
  const float rw = 1, lw = 0;
  for (int t = 1; t <= 2; t++) {
    int ca = (axis + t) % 3;
    box_t b = plane_box(g, axis, ghost, ca, 0);
    BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_CBX, ca) = rw * buf[k++] + lw * FC(f, v + in, F_CBX, ca); }
  }
  return k;
}

int orc_pack_jf(float *buf, const orc_field_t *f, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, plane = dir < 3 ? 1 : n[axis] + 1, k = 0;
  for (int t = 1; t <= 2; t++) {
    int ca = (axis + t) % 3;
    box_t b = plane_box(g, axis, plane, ca, 1);
    BOX_LOOP(b) buf[k++] = FC(f, VOXEL(x, y, z), F_JFX, ca);
  }
  return k;
}

int orc_unpack_jf(orc_field_t *f, const float *buf, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, plane = dir < 3 ? n[axis] + 1 : 1, k = 0;
  const float lw = 1, rw = 1;
  for (int t = 1; t <= 2; t++) {
    int ca = (axis + t) % 3;
    box_t b = plane_box(g, axis, plane, ca, 1);
    BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_JFX, ca) = lw * FC(f, v, F_JFX, ca) + rw * buf[k++]; }
  }
  return k;
}

/* remote.c:416-506 restricted to faces this domain shares with itself */
void orc_synchronize_jf_local(orc_field_t *f, const orc_grid_t *g) {
  orc_local_adjust_jf(f, g);
  for (int axis = 0; axis < 3; axis++) {
    if (g->fbc[axis] != g->rank || g->fbc[axis + 3] != g->rank) continue;
    int cnt = orc_tang_b_count(g, axis);
    float *lo = (float *)malloc(sizeof(float) * (size_t)cnt), *hi = (float *)malloc(sizeof(float) * (size_t)cnt);
    orc_pack_jf(lo, f, g, axis);       /* travelling -axis: plane 1     */
    orc_pack_jf(hi, f, g, axis + 3);   /* travelling +axis: plane n+1   */
    orc_unpack_jf(f, lo, g, axis);     /* lands on plane n+1            */
    orc_unpack_jf(f, hi, g, axis + 3); /* lands on plane 1              */
    free(lo); free(hi);
  }
}

/* ------------------------------------------------------------------------------------------
 * field_advance/standard/advance_b.c:12-14,38-40,56-68,122-158                                */
void orc_advance_b(orc_field_t *f, const orc_grid_t *g, float frac) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const float px = (nx > 1) ? frac * g->cvac * g->dt * g->rdx : 0;
  const float py = (ny > 1) ? frac * g->cvac * g->dt * g->rdy : 0;
  const float pz = (nz > 1) ? frac * g->cvac * g->dt * g->rdz : 0;
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x + 1, y, z)], *fy = &f[VOXEL(x, y + 1, z)], *fz = &f[VOXEL(x, y, z + 1)];
        if (y <= ny && z <= nz) f0->cbx -= (py * (fy->ez - f0->ez) - pz * (fz->ey - f0->ey));
        if (z <= nz && x <= nx) f0->cby -= (pz * (fz->ex - f0->ex) - px * (fx->ez - f0->ez));
        if (x <= nx && y <= ny) f0->cbz -= (px * (fx->ey - f0->ey) - py * (fy->ex - f0->ex));
      }
  orc_local_adjust_norm_b(f, g);
}

/* field_advance/standard/advance_e.c:8-25,104-108,114-115,153-329 */
void orc_advance_e(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const float damp = g->damp;
  const float px = (nx > 1) ? (1 + damp) * g->cvac * g->dt * g->rdx : 0;
  const float py = (ny > 1) ? (1 + damp) * g->cvac * g->dt * g->rdy : 0;
  const float pz = (nz > 1) ? (1 + damp) * g->cvac * g->dt * g->rdz : 0;
  const float cj = g->dt / g->eps0;

  /* ghosts: faces shared with this same domain (the reference sends to itself, grid_comm.c:17-49) */
  for (int dir = 0; dir < 6; dir++) {
    if (g->fbc[dir] != g->rank) continue;
    float *buf = (float *)malloc(sizeof(float) * (size_t)orc_tang_b_count(g, dir));
    orc_pack_tang_b(buf, f, g, dir);
    orc_unpack_tang_b(f, buf, g, dir);
    free(buf);
  }
  orc_local_ghost_tang_b(f, g);

  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x - 1, y, z)], *fy = &f[VOXEL(x, y - 1, z)], *fz = &f[VOXEL(x, y, z - 1)];
        if (x <= nx) {
          f0->tcax = (py * (f0->cbz * m[f0->fmatz].rmuz - fy->cbz * m[fy->fmatz].rmuz) -
                      pz * (f0->cby * m[f0->fmaty].rmuy - fz->cby * m[fz->fmaty].rmuy)) - damp * f0->tcax;
          f0->ex = m[f0->ematx].decayx * f0->ex + m[f0->ematx].drivex * (f0->tcax - cj * f0->jfx);
        }
        if (y <= ny) {
          f0->tcay = (pz * (f0->cbx * m[f0->fmatx].rmux - fz->cbx * m[fz->fmatx].rmux) -
                      px * (f0->cbz * m[f0->fmatz].rmuz - fx->cbz * m[fx->fmatz].rmuz)) - damp * f0->tcay;
          f0->ey = m[f0->ematy].decayy * f0->ey + m[f0->ematy].drivey * (f0->tcay - cj * f0->jfy);
        }
        if (z <= nz) {
          f0->tcaz = (px * (f0->cby * m[f0->fmaty].rmuy - fx->cby * m[fx->fmaty].rmuy) -
                      py * (f0->cbx * m[f0->fmatx].rmux - fy->cbx * m[fy->fmatx].rmux)) - damp * f0->tcaz;
          f0->ez = m[f0->ematz].decayz * f0->ez + m[f0->ematz].drivez * (f0->tcaz - cj * f0->jfz);
        }
      }
  orc_local_adjust_tang_e(f, g);
}

/* ------------------------------------------------------------------------------------------
 * species_advance/standard/boundary_p.c:9-71                                                 */
void orc_accumulate_rhob(orc_field_t *f, const orc_particle_t *p, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny;
  float w0, w1, w2, w3, w4, w5, w6, w7, t;
  int i, j, k;
  t = p->dx; w0 = 0.125 * p->q * g->rdx * g->rdy * g->rdz;
  t *= w0; w1 = w0 + t; w0 -= t;
  t = p->dy; w3 = 1 + t; w2 = w0 * w3; w3 *= w1; t = 1 - t; w0 *= t; w1 *= t;
  t = p->dz; w7 = 1 + t; w4 = w0 * w7; w5 = w1 * w7; w6 = w2 * w7; w7 *= w3;
  t = 1 - t; w0 *= t; w1 *= t; w2 *= t; w3 *= t;
  i = p->i; j = i / (g->nx + 2); i -= j * (g->nx + 2); k = j / (g->ny + 2); j -= k * (g->ny + 2);
  if (i == 1)     w0 += w0, w2 += w2, w4 += w4, w6 += w6;
  if (i == g->nx) w1 += w1, w3 += w3, w5 += w5, w7 += w7;
  if (j == 1)     w0 += w0, w1 += w1, w4 += w4, w5 += w5;
  if (j == g->ny) w2 += w2, w3 += w3, w6 += w6, w7 += w7;
  if (k == 1)     w0 += w0, w1 += w1, w2 += w2, w3 += w3;
  if (k == g->nz) w4 += w4, w5 += w5, w6 += w6, w7 += w7;
  int v = p->i, sy = nx + 2, sz = (nx + 2) * (ny + 2);
  f[v].rhob += w0; f[v + 1].rhob += w1; f[v + sy].rhob += w2; f[v + sy + 1].rhob += w3;
  f[v + sz].rhob += w4; f[v + sz + 1].rhob += w5; f[v + sz + sy].rhob += w6; f[v + sz + sy + 1].rhob += w7;
}

/* species_advance/standard/boundary_p.c:194-320: movers in reverse, face tests, back-fill */
int orc_boundary_p_pack(orc_particle_t *p0, int np, const orc_mover_t *pm0, int nm, int sp_id,
                        orc_field_t *f, const orc_grid_t *g,
                        orc_injector_t *out[6], int ns[6], int cap) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  const int stride[3] = {1, nx + 2, (nx + 2) * (ny + 2)};
  for (const orc_mover_t *pm = pm0 + nm - 1; nm; pm--, nm--) {
    orc_particle_t *r = p0 + pm->i;
    const float d[3] = {r->dx, r->dy, r->dz}, u[3] = {r->ux, r->uy, r->uz};
    int handled = 0;
    for (int face = 0; face < 6 && !handled; face++) {
      int axis = face % 3, hi = face >= 3;
      int cond = hi ? ((d[axis] == 1) & (u[axis] > 0)) : ((d[axis] == -1) & (u[axis] < 0));
      if (!cond) continue;
      int code = g->pbc[face];
      if (code == ORC_ABSORB_PARTICLES) {
        orc_accumulate_rhob(f, r, g);
        r[0] = p0[--np];
        handled = 1;
      } else if (code >= 0 && code != g->rank) {
        if (ns[face] >= cap) DIE("injector buffer overflow");
        orc_injector_t *pi = &out[face][ns[face]++];
        pi->dx = (axis == 0) ? -r->dx : r->dx;
        pi->dy = (axis == 1) ? -r->dy : r->dy;
        pi->dz = (axis == 2) ? -r->dz : r->dz;
        /* receiver's local voxel: same transverse coords, first/last cell along the axis
         * (grid/ops.c:157-171; equal-sized neighbours) */
        pi->i = r->i + (hi ? -(n[axis] - 1) : (n[axis] - 1)) * stride[axis];
        pi->ux = r->ux; pi->uy = r->uy; pi->uz = r->uz; pi->q = r->q;
        pi->dispx = pm->dispx; pi->dispy = pm->dispy; pi->dispz = pm->dispz;
        pi->sp_id = sp_id;
        r[0] = p0[--np];
        handled = 1;
      }
    }
    if (!handled) {   /* boundary_p.c:312-316 "Unknown boundary interaction ... using absorption" */
      orc_accumulate_rhob(f, r, g);
      r[0] = p0[--np];
    }
  }
  return np;
}

/* species_advance/standard/boundary_p.c:457-497 */
int orc_boundary_p_inject(orc_particle_t *p0, int np, orc_mover_t *pm0, int *nm_io,
                          const orc_injector_t *in, int n, orc_accumulator_t *a0,
                          const orc_grid_t *g) {
  int nm = *nm_io;
  for (const orc_injector_t *pi = in + n - 1; n; pi--, n--) {
    orc_particle_t *p = p0 + np;
    orc_mover_t *pm = pm0 + nm;
    p->dx = pi->dx; p->dy = pi->dy; p->dz = pi->dz; p->i = pi->i;
    p->ux = pi->ux; p->uy = pi->uy; p->uz = pi->uz; p->q = pi->q;
    pm->dispx = pi->dispx; pm->dispy = pi->dispy; pm->dispz = pi->dispz;
    pm->i = np;
    np++;
    nm += orc_move_p(p0, pm, a0, g);
  }
  *nm_io = nm;
  return np;
}

/* boundary/maxwellian_reflux.c:116-175 -- the custom particle boundary handler: a particle that reaches the face is
 * re-emitted with the momentum of the flux of a Maxwellian at the wall and travels what is left of the step with it.
 * The three random numbers the handler draws -- mt_frand (uniform on (0,1)), then two mt_frandn (normals) -- are
 * passed in, in that order, so that the restatement can be checked particle by particle against the reference's
 * own handler fed from a generator in the same state (oracle/gen_reflux.py).  face: 0..5 = -x,-y,-z,+x,+y,+z.   */
void orc_maxwellian_reflux(const float draws[3], const orc_particle_t *r, const orc_mover_t *pm,
                           const orc_grid_t *g, float ut_para, float ut_perp, int face, int sp_id,
                           orc_injector_t *pi) {
  static const int perm[6][3] = {{0, 1, 2}, {2, 0, 1}, {1, 2, 0}, {0, 1, 2}, {2, 0, 1}, {1, 2, 0}};   /* :70-75 */
  static const float scale[6] = {M_SQRT2, M_SQRT2, M_SQRT2, -M_SQRT2, -M_SQRT2, -M_SQRT2};           /* :76-77 */
  float u[3], ux, uy, uz, dispx, dispy, dispz, ratio;
  u[0] = ut_para * scale[face] * sqrtf(-logf(draws[0]));                                              /* :120 */
  u[1] = ut_perp * draws[1];
  u[2] = ut_perp * draws[2];
  ux = u[perm[face][0]]; uy = u[perm[face][1]]; uz = u[perm[face][2]];
  dispx = g->dx * pm->dispx; dispy = g->dy * pm->dispy; dispz = g->dz * pm->dispz;                     /* :146-148 */
  ratio = r->ux * r->ux + r->uy * r->uy + r->uz * r->uz;
  ratio = sqrtf(((1 + ratio) * (dispx * dispx + dispy * dispy + dispz * dispz)) /
                ((1 + (ux * ux + uy * uy + uz * uz)) * (FLT_MIN + ratio)));
  dispx = ux * ratio * g->rdx; dispy = uy * ratio * g->rdy; dispz = uz * ratio * g->rdz;
  pi->dx = r->dx; pi->dy = r->dy; pi->dz = r->dz; pi->i = r->i;                                         /* :163-174 */
  pi->ux = ux; pi->uy = uy; pi->uz = uz; pi->q = r->q;
  pi->dispx = dispx; pi->dispy = dispy; pi->dispz = dispz; pi->sp_id = sp_id;
}

/* emitter/child-langmuir.c:15-120 -- the surface emission model: every listed cell face whose normal field pulls the
 * species out of it emits n_emit_per_face particles that share the Child-law charge, start on the face with a
 * half-Maxwellian normal momentum, a Maxwellian tangential one and a uniformly random age, leave their charge,
 * negated, in rhob, and make the rest of their first step through move_p.  The six numbers the model draws per
 * particle -- mt_drand_c, mt_drand_c (position on the face), mt_drandn x 3 (momentum: normal, then the two
 * tangential components in cyclic order), mt_drand_c0 (age) -- are passed in, in emission order, so that the
 * restatement can be checked against the reference's own function fed from a generator in the same state
 * (oracle/gen_reflux.py).  component[n] = 32 * voxel + BOUNDARY(i,j,k) (emitter.h:18-21).  Returns the new np. */
int orc_child_langmuir(orc_particle_t *p0, int np, int max_np, orc_mover_t *pm0, int *nm_io, int max_nm,
                       const int *component, int n_component, int n_emit_per_face, float ut_perp, float ut_para,
                       float q_m, const orc_interpolator_t *fi, orc_field_t *f, orc_accumulator_t *a,
                       const orc_grid_t *g, const double *draws) {
  int nm = *nm_io;
  for (int n = 0; n < n_component; n++) {
    const int i = component[n] >> 5, type = component[n] & 31;
    /* BOUNDARY(i,j,k) = (i+1) + 3*((j+1) + 3*(k+1)): -x 12, -y 10, -z 4, +x 14, +y 16, +z 22; dir = the sign in EMIT_PARTICLES */
    int axis; float dir;
    if (type == 12) { axis = 0; dir = 1; } else if (type == 10) { axis = 1; dir = 1; } else if (type == 4) { axis = 2; dir = 1; }
    else if (type == 14) { axis = 0; dir = -1; } else if (type == 16) { axis = 1; dir = -1; } else if (type == 22) { axis = 2; dir = -1; }
    else continue;                                                                                      /* :113 */
    const float e_n = axis == 0 ? fi[i].ex : axis == 1 ? fi[i].ey : fi[i].ez;
    const float dX = axis == 0 ? g->dx : axis == 1 ? g->dy : g->dz;
    const float dY = axis == 0 ? g->dy : axis == 1 ? g->dz : g->dx;
    const float dZ = axis == 0 ? g->dz : axis == 1 ? g->dx : g->dy;
    if (!(q_m * (dir * e_n) > 0)) continue;                                                             /* :44 */
    float qp, age;
    int m = n_emit_per_face;
    qp = g->eps0 * dY * dZ * g->dt * sqrt((32. / 81.) * fabs(q_m * e_n * e_n * e_n) / dX) / (float)m;   /* :49-51 */
    if (q_m < 0) qp = -qp;
    if (np + m >= max_np) { m = max_np - np; if (m < 0) m = 0; }                                        /* :53-57 */
    for (; m; m--) {
      orc_particle_t *p = p0 + np++;
      float *d[3] = {&p->dx, &p->dy, &p->dz}, *u[3] = {&p->ux, &p->uy, &p->uz};
      const int aX = axis, aY = (axis + 1) % 3, aZ = (axis + 2) % 3;
      *d[aX] = -(dir * 1);
      *d[aY] = 2 * draws[0] - 1;
      *d[aZ] = 2 * draws[1] - 1;
      p->i = i;
      *u[aX] = dir * fabs(ut_para * draws[2]);
      *u[aY] = ut_perp * draws[3];
      *u[aZ] = ut_perp * draws[4];
      p->q = -qp; orc_accumulate_rhob(f, p, g); p->q = qp;                                              /* :67 */
      if (nm >= max_nm) { draws += 6; continue; }                                                       /* (the age is drawn after this test: :68-72) */
      orc_mover_t *pm = pm0 + nm;
      age = draws[5];
      age *= g->cvac * g->dt / sqrt(*u[aX] * *u[aX] + *u[aY] * *u[aY] + *u[aZ] * *u[aZ] + 1);
      float *disp[3] = {&pm->dispx, &pm->dispy, &pm->dispz};
      *disp[aX] = *u[aX] * age / dX;
      *disp[aY] = *u[aY] * age / dY;
      *disp[aZ] = *u[aZ] * age / dZ;
      pm->i = np - 1;
      nm += orc_move_p(p0, pm, a, g);
      draws += 6;
    }
  }
  *nm_io = nm;
  return np;
}

/* ==========================================================================================
 * Divergence cleaning family and charge densities.
 * Float-index of the scalar members of orc_field_t.                                          */
enum { F_DIV_E_ERR = 3, F_DIV_B_ERR = 7, F_RHOB = 11, F_RHOF = 15 };

/* X_NODE_LOOP / X_FACE_LOOP of local.c:38-46, remote.c:33-41 */
static box_t node_box(const orc_grid_t *g, int axis, int plane) {
  const int n[3] = {g->nx, g->ny, g->nz};
  box_t b;
  for (int d = 0; d < 3; d++) { b.lo[d] = 1; b.hi[d] = n[d] + 1; }
  b.lo[axis] = b.hi[axis] = plane;
  return b;
}
static box_t face_box(const orc_grid_t *g, int axis, int plane) { return plane_box(g, axis, plane, axis, 0); }

/* field_advance/standard/sfa.c:213-234 */
void orc_clear_rhof(orc_field_t *f, const orc_grid_t *g) {
  const int nv = orc_nv(g);
  for (int v = 0; v < nv; v++) f[v].rhof = 0;
}

/* species_advance/standard/rho_p.c:23-86 */
void orc_accumulate_rho_p(orc_field_t *f0, const orc_particle_t *p, int n, const orc_grid_t *g) {
  const int sy = g->nx + 2, sz = sy * (g->ny + 2);
  float w0, w1, w2, w3, w4, w5, w6, w7, t;
  const float r8V = 0.125 * g->rdx * g->rdy * g->rdz;
  for (; n; n--, p++) {
    t = p->dx; w0 = r8V * p->q; t *= w0; w1 = w0 + t; w0 -= t;
    t = p->dy; w3 = 1 + t; w2 = w0 * w3; w3 *= w1; t = 1 - t; w0 *= t; w1 *= t;
    t = p->dz; w7 = 1 + t; w4 = w0 * w7; w5 = w1 * w7; w6 = w2 * w7; w7 *= w3;
    t = 1 - t; w0 *= t; w1 *= t; w2 *= t; w3 *= t;
    orc_field_t *f = f0 + p->i;
    f[0].rhof += w0; f[1].rhof += w1; f[sy].rhof += w2; f[sy + 1].rhof += w3;
    f[sz].rhof += w4; f[sz + 1].rhof += w5; f[sz + sy].rhof += w6; f[sz + sy + 1].rhof += w7;
  }
}

/* field_advance/standard/local.c:368-445: all six faces for rhof, then all six for rhob */
void orc_local_adjust_rho(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  for (int comp = 0; comp < 2; comp++)
    for (int face = 0; face < 6; face++) {
      int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
      if (!is_local_bc(bc)) continue;
      box_t b = node_box(g, axis, hi ? n[axis] + 1 : 1);
      if (bc == ORC_PEC_FIELDS) { BOX_LOOP(b) FC(f, VOXEL(x, y, z), comp ? F_RHOB : F_RHOF, 0) = 0; }
      else if (!comp)           { BOX_LOOP(b) FC(f, VOXEL(x, y, z), F_RHOF, 0) *= 2; }
    }
}

/* remote.c:548-583: (rhof, rhob) pairs over the node plane 1 (travelling -axis) or n+1 (+axis);
 * they land on the shared plane at the other end: rhof = lw*rhof + rw*recv, rhob = hlw*rhob +
 * hrw*recv with the cell-size weights of a uniform mesh.                                      */
int orc_rho_count(const orc_grid_t *g, int dir) {
  const int n[3] = {g->nx, g->ny, g->nz};
  int a = dir % 3;
  return 2 * (n[(a + 1) % 3] + 1) * (n[(a + 2) % 3] + 1);
}
int orc_pack_rho(float *buf, const orc_field_t *f, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, k = 0;
  box_t b = node_box(g, axis, dir < 3 ? 1 : n[axis] + 1);
  BOX_LOOP(b) { const orc_field_t *f0 = &f[VOXEL(x, y, z)]; buf[k++] = f0->rhof; buf[k++] = f0->rhob; }
  return k;
}
int orc_unpack_rho(orc_field_t *f, const float *buf, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, k = 0;
  const float d = axis == 0 ? g->dx : axis == 1 ? g->dy : g->dz;
  float hrw = d, hlw = hrw + d, lw, rw;     /* remote.c:566-571 with the remote cell size == ours */
  hrw /= hlw; hlw = d / hlw; lw = hlw + hlw; rw = hrw + hrw;
  box_t b = node_box(g, axis, dir < 3 ? n[axis] + 1 : 1);
  BOX_LOOP(b) {
    orc_field_t *f0 = &f[VOXEL(x, y, z)];
    f0->rhof = lw * f0->rhof + rw * buf[k]; k++;
    f0->rhob = hlw * f0->rhob + hrw * buf[k]; k++;
  }
  return k;
}
/* remote.c:533-622 restricted to faces this domain shares with itself */
void orc_synchronize_rho_self(orc_field_t *f, const orc_grid_t *g, int axis) {
  if (g->fbc[axis] != g->rank || g->fbc[axis + 3] != g->rank) return;
  int cnt = orc_rho_count(g, axis);
  float *lo = (float *)malloc(sizeof(float) * (size_t)cnt), *hi = (float *)malloc(sizeof(float) * (size_t)cnt);
  orc_pack_rho(lo, f, g, axis); orc_pack_rho(hi, f, g, axis + 3);
  orc_unpack_rho(f, lo, g, axis); orc_unpack_rho(f, hi, g, axis + 3);
  free(lo); free(hi);
}
void orc_synchronize_rho_local(orc_field_t *f, const orc_grid_t *g) {
  orc_local_adjust_rho(f, g);
  for (int axis = 0; axis < 3; axis++) orc_synchronize_rho_self(f, g, axis);
}

/* Normal-E ghosts: remote.c:136-207 for faces shared with this same domain (plane 1 -> ghost
 * n+1, plane n -> ghost 0, over the node box), local.c:128-180 for local faces.               */
static void ghost_norm_e(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  const int stride[3] = {1, nx + 2, (nx + 2) * (ny + 2)};
  for (int dir = 0; dir < 6; dir++) {
    int axis = dir % 3;
    if (g->fbc[dir] == g->rank) {
      int from = dir < 3 ? 1 : n[axis], to = dir < 3 ? n[axis] + 1 : 0, off = (from - to) * stride[axis];
      box_t b = node_box(g, axis, to);
      BOX_LOOP(b) { int v = VOXEL(x, y, z); FC(f, v, F_EX, axis) = FC(f, v + off, F_EX, axis); }
    }
  }
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc)) continue;
    int ghost = hi ? n[axis] + 1 : 0, in = hi ? -stride[axis] : stride[axis];
    float sign;
    box_t b = node_box(g, axis, ghost);
    if (bc == ORC_ABSORB_FIELDS) {                      /* local.c:162-170: linear extrapolation */
      BOX_LOOP(b) {
        int v = VOXEL(x, y, z);
        FC(f, v, F_EX, axis) = 2 * FC(f, v + in, F_EX, axis) - FC(f, v + 2 * in, F_EX, axis);
        FC(f, v, F_TCAX, axis) = 2 * FC(f, v + in, F_TCAX, axis) - FC(f, v + 2 * in, F_TCAX, axis);
      }
      continue;
    }
    if (bc == ORC_PEC_FIELDS) sign = 1;
    else if (bc == ORC_SYMMETRIC_FIELDS || bc == ORC_PMC_FIELDS) sign = -1;
    else DIE("Bad boundary condition encountered.");
    BOX_LOOP(b) {
      int v = VOXEL(x, y, z);
      FC(f, v, F_EX, axis) = sign * FC(f, v + in, F_EX, axis);
      FC(f, v, F_TCAX, axis) = sign * FC(f, v + in, F_TCAX, axis);
    }
  }
}

/* compute_div_e_err.c:6-11 / compute_rhob.c:8-12 at every node 1..n+1 */
static void div_e_like(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g, int rhob) {
  const int nx = g->nx, ny = g->ny, nz = g->nz, n[3] = {g->nx, g->ny, g->nz};
  const float px = (nx > 1) ? (rhob ? g->eps0 * g->rdx : g->rdx) : 0;
  const float py = (ny > 1) ? (rhob ? g->eps0 * g->rdy : g->rdy) : 0;
  const float pz = (nz > 1) ? (rhob ? g->eps0 * g->rdz : g->rdz) : 0;
  const float cj = 1. / g->eps0;
  ghost_norm_e(f, g);
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x - 1, y, z)], *fy = &f[VOXEL(x, y - 1, z)], *fz = &f[VOXEL(x, y, z - 1)];
        if (rhob)
          f0->rhob = m[f0->nmat].nonconductive *
            (px * (m[f0->ematx].epsx * f0->ex - m[fx->ematx].epsx * fx->ex) +
             py * (m[f0->ematy].epsy * f0->ey - m[fy->ematy].epsy * fy->ey) +
             pz * (m[f0->ematz].epsz * f0->ez - m[fz->ematz].epsz * fz->ez) - f0->rhof);
        else
          f0->div_e_err = m[f0->nmat].nonconductive *
            (px * (m[f0->ematx].epsx * f0->ex - m[fx->ematx].epsx * fx->ex) +
             py * (m[f0->ematy].epsy * f0->ey - m[fy->ematy].epsy * fy->ey) +
             pz * (m[f0->ematz].epsz * f0->ez - m[fz->ematz].epsz * fz->ez) - cj * (f0->rhof + f0->rhob));
      }
  /* local.c:298-330 (div_e_err: zero on PEC faces) / local.c:414-445 (rhob: zero on PEC faces) */
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc) || !(bc == ORC_PEC_FIELDS || (!rhob && bc == ORC_ABSORB_FIELDS))) continue;
    box_t b = node_box(g, axis, hi ? n[axis] + 1 : 1);
    BOX_LOOP(b) FC(f, VOXEL(x, y, z), rhob ? F_RHOB : F_DIV_E_ERR, 0) = 0;
  }
}
void orc_compute_div_e_err(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g) { div_e_like(f, m, g, 0); }
void orc_compute_rhob(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g) { div_e_like(f, m, g, 1); }

/* compute_rms_div_e_err.c:36-48,92-156: this domain's two sums (local[0], local[1]); the caller
 * adds them over domains and forms eps0*sqrt(sum0/sum1).  Interior products are float products
 * added to a double, the weighted exterior ones double products, as there.                    */
void orc_rms_div_e_err_local(double *local2, const orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  double err = 0;
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        const float e = f[VOXEL(x, y, z)].div_e_err;
        const int nb = (x == 1 || x == nx + 1) + (y == 1 || y == ny + 1) + (z == 1 || z == nz + 1);
        if (nb == 0) err += e * e;
        else err += (nb == 1 ? 0.5 : nb == 2 ? 0.25 : 0.125) * (double)e * (double)e;
      }
  local2[0] = err * g->dx * g->dy * g->dz;
  local2[1] = g->nx * g->ny * g->nz * g->dx * g->dy * g->dz;
}

/* clean_div_e.c:6-13,39-46,127-179 */
static void marder_p(const orc_grid_t *g, float *px, float *py, float *pz) {
  float alphadt;
  *px = (g->nx > 1) ? g->rdx : 0; *py = (g->ny > 1) ? g->rdy : 0; *pz = (g->nz > 1) ? g->rdz : 0;
  alphadt = 0.3888889 / (*px * *px + *py * *py + *pz * *pz);
  *px *= alphadt; *py *= alphadt; *pz *= alphadt;
}
void orc_clean_div_e(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  float px, py, pz;
  marder_p(g, &px, &py, &pz);
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x + 1, y, z)], *fy = &f[VOXEL(x, y + 1, z)], *fz = &f[VOXEL(x, y, z + 1)];
        if (x <= nx) f0->ex += m[f0->ematx].drivex * px * (fx->div_e_err - f0->div_e_err);
        if (y <= ny) f0->ey += m[f0->ematy].drivey * py * (fy->div_e_err - f0->div_e_err);
        if (z <= nz) f0->ez += m[f0->ematz].drivez * pz * (fz->div_e_err - f0->div_e_err);
      }
  orc_local_adjust_tang_e(f, g);
}

/* compute_div_b_err.c:44-48 over cells 1..n */
void orc_compute_div_b_err(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const float px = (nx > 1) ? g->rdx : 0, py = (ny > 1) ? g->rdy : 0, pz = (nz > 1) ? g->rdz : 0;
  for (int z = 1; z <= nz; z++)
    for (int y = 1; y <= ny; y++)
      for (int x = 1; x <= nx; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x + 1, y, z)], *fy = &f[VOXEL(x, y + 1, z)], *fz = &f[VOXEL(x, y, z + 1)];
        f0->div_b_err = px * (fx->cbx - f0->cbx) + py * (fy->cby - f0->cby) + pz * (fz->cbz - f0->cbz);
      }
}
/* compute_rms_div_b_err.c:36-48,88-93 */
void orc_rms_div_b_err_local(double *local2, const orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  double err = 0;
  for (int z = 1; z <= nz; z++)
    for (int y = 1; y <= ny; y++)
      for (int x = 1; x <= nx; x++) { const float e = f[VOXEL(x, y, z)].div_b_err; err += e * e; }
  local2[0] = err * g->dx * g->dy * g->dz;
  local2[1] = g->nx * g->ny * g->nz * g->dx * g->dy * g->dz;
}
/* clean_div_b.c:6-8,31-38,95-247; div_b ghosts remote.c:209-281 (self) and local.c:182-216 */
void orc_clean_div_b(orc_field_t *f, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz, n[3] = {g->nx, g->ny, g->nz};
  const int stride[3] = {1, nx + 2, (nx + 2) * (ny + 2)};
  float px, py, pz;
  marder_p(g, &px, &py, &pz);
  for (int dir = 0; dir < 6; dir++) {
    int axis = dir % 3;
    if (g->fbc[dir] != g->rank) continue;
    int from = dir < 3 ? 1 : n[axis], to = dir < 3 ? n[axis] + 1 : 0, off = (from - to) * stride[axis];
    box_t b = face_box(g, axis, to);
    BOX_LOOP(b) { int v = VOXEL(x, y, z); f[v].div_b_err = f[v + off].div_b_err; }
  }
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3, bc = g->fbc[face];
    if (!is_local_bc(bc)) continue;
    int ghost = hi ? n[axis] + 1 : 0, in = hi ? -stride[axis] : stride[axis];
    box_t b = face_box(g, axis, ghost);
    if (bc == ORC_PEC_FIELDS) { BOX_LOOP(b) { int v = VOXEL(x, y, z); f[v].div_b_err = f[v + in].div_b_err; } }
    else if (bc == ORC_SYMMETRIC_FIELDS || bc == ORC_PMC_FIELDS) { BOX_LOOP(b) { int v = VOXEL(x, y, z); f[v].div_b_err = -f[v + in].div_b_err; } }
    else { BOX_LOOP(b) f[VOXEL(x, y, z)].div_b_err = 0; }
  }
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x - 1, y, z)], *fy = &f[VOXEL(x, y - 1, z)], *fz = &f[VOXEL(x, y, z - 1)];
        if (y <= ny && z <= nz) f0->cbx += px * (f0->div_b_err - fx->div_b_err);
        if (z <= nz && x <= nx) f0->cby += py * (f0->div_b_err - fy->div_b_err);
        if (x <= nx && y <= ny) f0->cbz += pz * (f0->div_b_err - fz->div_b_err);
      }
  orc_local_adjust_norm_b(f, g);
}

/* compute_curl_b.c:8-18,94-97,103-104: tca = curl(cB/mu)*c*dt over the E extents */
void orc_compute_curl_b(orc_field_t *f, const orc_material_coefficient_t *m, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const float px = (nx > 1) ? g->cvac * g->dt * g->rdx : 0;
  const float py = (ny > 1) ? g->cvac * g->dt * g->rdy : 0;
  const float pz = (nz > 1) ? g->cvac * g->dt * g->rdz : 0;
  for (int dir = 0; dir < 6; dir++) {
    if (g->fbc[dir] != g->rank) continue;
    float *buf = (float *)malloc(sizeof(float) * (size_t)orc_tang_b_count(g, dir));
    orc_pack_tang_b(buf, f, g, dir); orc_unpack_tang_b(f, buf, g, dir);
    free(buf);
  }
  orc_local_ghost_tang_b(f, g);
  for (int z = 1; z <= nz + 1; z++)
    for (int y = 1; y <= ny + 1; y++)
      for (int x = 1; x <= nx + 1; x++) {
        orc_field_t *f0 = &f[VOXEL(x, y, z)];
        const orc_field_t *fx = &f[VOXEL(x - 1, y, z)], *fy = &f[VOXEL(x, y - 1, z)], *fz = &f[VOXEL(x, y, z - 1)];
        if (x <= nx) f0->tcax = py * (f0->cbz * m[f0->fmatz].rmuz - fy->cbz * m[fy->fmatz].rmuz) -
                                pz * (f0->cby * m[f0->fmaty].rmuy - fz->cby * m[fz->fmaty].rmuy);
        if (y <= ny) f0->tcay = pz * (f0->cbx * m[f0->fmatx].rmux - fz->cbx * m[fz->fmatx].rmux) -
                                px * (f0->cbz * m[f0->fmatz].rmuz - fx->cbz * m[fx->fmatz].rmuz);
        if (z <= nz) f0->tcaz = px * (f0->cby * m[f0->fmaty].rmuy - fx->cby * m[fx->fmaty].rmuy) -
                                py * (f0->cbx * m[f0->fmatx].rmux - fy->cbx * m[fy->fmatx].rmux);
      }
}

/* remote.c:298-414 restricted to faces this domain shares with itself: per axis, the values on
 * plane 1 and plane n+1 (normal cB over the face box, then (e,tca) of the two tangential
 * components over their edge boxes) are replaced by their average; returns the sum of squared
 * differences of cB and e (double), this domain's share of the reference's allsum.            */
double orc_synchronize_tang_e_norm_b_self(orc_field_t *f, const orc_grid_t *g, int axis);
void orc_local_adjust_tang_e_norm_b(orc_field_t *f, const orc_grid_t *g);
double orc_synchronize_tang_e_norm_b_local(orc_field_t *f, const orc_grid_t *g) {
  double err = 0;
  orc_local_adjust_tang_e_norm_b(f, g);
  for (int axis = 0; axis < 3; axis++) err += orc_synchronize_tang_e_norm_b_self(f, g, axis);
  return err;
}

/* ==========================================================================================
 * Hydro moments: species_advance/standard/hydro_p.c:24-176,
 * sf_interface/sf_interface.c:29-36 (clear_hydro), sf_interface/hydro.c:28-200.               */
void orc_clear_hydro(orc_hydro_t *h, const orc_grid_t *g) { memset(h, 0, sizeof(*h) * (size_t)orc_nv(g)); }

void orc_accumulate_hydro_p(orc_hydro_t *h0, const orc_particle_t *p0, int n, float q_m,
                            const orc_interpolator_t *f0, const orc_grid_t *g) {
  float dx, dy, dz; int ii;
  float ux, uy, uz, q;
  float vx, vy, vz, ke_mc;
  float w0, w1, w2, w3, w4, w5, w6, w7;
  float qdt_2mc, qdt_4mc2, c, r8V, mc_q;
  const orc_particle_t *p;
  const orc_interpolator_t *f;
  orc_hydro_t *h;
  const int stride_10 = 1, stride_21 = (g->nx + 2) - 1, stride_43 = (g->nx + 2) * (g->ny + 2) - (g->nx + 2) - 1;

  qdt_2mc  = 0.5*q_m*g->dt/g->cvac;
  qdt_4mc2 = 0.25*q_m*g->dt/(g->cvac*g->cvac);
  c = g->cvac;
  r8V = 0.125*g->rdx*g->rdy*g->rdz;
  mc_q = g->cvac/q_m;

  for( p=p0; n; n--, p++ ) {
    dx = p->dx; dy = p->dy; dz = p->dz; ii = p->i;
    ux = p->ux; uy = p->uy; uz = p->uz; q = p->q;
    f  = f0 + ii;
    ux += qdt_2mc*((f->ex+dy*f->dexdy) + dz*(f->dexdz+dy*f->d2exdydz));
    uy += qdt_2mc*((f->ey+dz*f->deydz) + dx*(f->deydx+dz*f->d2eydzdx));
    uz += qdt_2mc*((f->ez+dx*f->dezdx) + dy*(f->dezdy+dx*f->d2ezdxdy));
    w5 = f->cbx + dx*f->dcbxdx;
    w6 = f->cby + dy*f->dcbydy;
    w7 = f->cbz + dz*f->dcbzdz;
    ke_mc = ux*ux + uy*uy + uz*uz;
    vz = sqrt(1+ke_mc);            /* double sqrt of a float sum, rounded to float (hydro_p.c:86) */
    ke_mc *= c/(vz+1);
    vz = c/vz;
    w0 = qdt_4mc2*vz;
    w1 = w5*w5 + w6*w6 + w7*w7;
    w2 = w0*w0*w1;
    w3 = w0*(1+(1./3.)*w2*(1+0.4*w2));   /* double arithmetic, rounded to float (hydro_p.c:92) */
    w4 = w3/(1 + w1*w3*w3); w4 += w4;
    w0 = ux + w3*( uy*w7 - uz*w6 );
    w1 = uy + w3*( uz*w5 - ux*w7 );
    w2 = uz + w3*( ux*w6 - uy*w5 );
    ux += w4*( w1*w7 - w2*w6 );
    uy += w4*( w2*w5 - w0*w7 );
    uz += w4*( w0*w6 - w1*w5 );
    vx  = ux*vz; vy  = uy*vz; vz *= uz;
    w0  = r8V*q; dx *= w0; w1  = w0+dx; w0 -= dx;
    w3  = 1+dy; w2  = w0*w3; w3 *= w1; dy  = 1-dy; w0 *= dy; w1 *= dy;
    w7  = 1+dz; w4  = w0*w7; w5  = w1*w7; w6  = w2*w7; w7 *= w3;
    dz  = 1-dz; w0 *= dz; w1 *= dz; w2 *= dz; w3 *= dz;
#   define ACCUM_HYDRO(wn)           \
    h->jx  += wn*vx; h->jy  += wn*vy; h->jz  += wn*vz; h->rho += wn; \
    wn *= mc_q; dx = wn*ux; dy = wn*uy; dz = wn*uz;                  \
    h->px  += dx; h->py  += dy; h->pz  += dz; h->ke  += wn*ke_mc;    \
    h->txx += dx*vx; h->tyy += dy*vy; h->tzz += dz*vz;               \
    h->tyz += dy*vz; h->tzx += dz*vx; h->txy += dx*vy
    h = h0 + ii;    ACCUM_HYDRO(w0);
    h += stride_10; ACCUM_HYDRO(w1);
    h += stride_21; ACCUM_HYDRO(w2);
    h += stride_10; ACCUM_HYDRO(w3);
    h += stride_43; ACCUM_HYDRO(w4);
    h += stride_10; ACCUM_HYDRO(w5);
    h += stride_21; ACCUM_HYDRO(w6);
    h += stride_10; ACCUM_HYDRO(w7);
#   undef ACCUM_HYDRO
  }
}

/* sf_interface/hydro.c:165-200: every moment doubled on the node plane of a local face */
void orc_local_adjust_hydro(orc_hydro_t *h, const orc_grid_t *g) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  for (int face = 0; face < 6; face++) {
    int axis = face % 3, hi = face >= 3;
    if (!is_local_bc(g->fbc[face])) continue;
    box_t b = node_box(g, axis, hi ? n[axis] + 1 : 1);
    BOX_LOOP(b) { float *m = (float *)&h[VOXEL(x, y, z)]; for (int k = 0; k < 14; k++) m[k] *= 2; }
  }
}
int orc_hydro_count(const orc_grid_t *g, int dir) { return 7 * orc_rho_count(g, dir); }   /* 14 per node, hydro.c:40 */
int orc_pack_hydro(float *buf, const orc_hydro_t *h, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, k = 0;
  box_t b = node_box(g, axis, dir < 3 ? 1 : n[axis] + 1);
  BOX_LOOP(b) { const float *m = (const float *)&h[VOXEL(x, y, z)]; for (int c = 0; c < 14; c++) buf[k++] = m[c]; }
  return k;
}
int orc_unpack_hydro(orc_hydro_t *h, const float *buf, const orc_grid_t *g, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, k = 0;
  const float d = axis == 0 ? g->dx : axis == 1 ? g->dy : g->dz;
  float rw = d, lw = rw + d;               /* hydro.c:69-74 with the remote cell size == ours */
  rw /= lw; lw = d / lw; lw += lw; rw += rw;
  box_t b = node_box(g, axis, dir < 3 ? n[axis] + 1 : 1);
  BOX_LOOP(b) { float *m = (float *)&h[VOXEL(x, y, z)]; for (int c = 0; c < 14; c++) { m[c] = lw * m[c] + rw * buf[k]; k++; } }
  return k;
}
/* hydro.c:28-163 restricted to faces this domain shares with itself */
void orc_synchronize_hydro_local(orc_hydro_t *h, const orc_grid_t *g) {
  orc_local_adjust_hydro(h, g);
  for (int axis = 0; axis < 3; axis++) {
    if (g->fbc[axis] != g->rank || g->fbc[axis + 3] != g->rank) continue;
    int cnt = orc_hydro_count(g, axis);
    float *lo = (float *)malloc(sizeof(float) * (size_t)cnt), *hi = (float *)malloc(sizeof(float) * (size_t)cnt);
    orc_pack_hydro(lo, h, g, axis); orc_pack_hydro(hi, h, g, axis + 3);
    orc_unpack_hydro(h, lo, g, axis); orc_unpack_hydro(h, hi, g, axis + 3);
    free(lo); free(hi);
  }
}

/* ==========================================================================================
 * Face messages of the divergence-cleaning family for faces shared with ANOTHER domain
 * (uniform meshes, without the leading cell-size float).  kind:
 *   0 normal E       remote.c:136-207: e_X over the node plane 1 (travelling -X) or n (+X); lands
 *                    on the ghost plane n+1 / 0
 *   1 div_b_err      remote.c:209-281: over the face plane 1 / n; lands on the ghost plane n+1 / 0
 *   2 tang E, norm B remote.c:298-414: plane 1 / n+1: cB_X over the face box, then (e_Y, tca_Y)
 *                    over the Y-edge box, then (e_Z, tca_Z) over the Z-edge box; the receiver (plane
 *                    n+1 / 1) replaces its values by the average and returns the sum of squared
 *                    differences of cB and e                                                   */
int orc_msg_count(const orc_grid_t *g, int kind, int dir) {
  const int n[3] = {g->nx, g->ny, g->nz};
  int a = dir % 3, nY = n[(a + 1) % 3], nZ = n[(a + 2) % 3];
  if (kind == 0) return (nY + 1) * (nZ + 1);
  if (kind == 1) return nY * nZ;
  return nY * nZ + 2 * nY * (nZ + 1) + 2 * nZ * (nY + 1);
}
int orc_pack_msg(float *buf, const orc_field_t *f, const orc_grid_t *g, int kind, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, k = 0;
  if (kind == 0) {
    box_t b = node_box(g, axis, dir < 3 ? 1 : n[axis]);
    BOX_LOOP(b) buf[k++] = FC(f, VOXEL(x, y, z), F_EX, axis);
  } else if (kind == 1) {
    box_t b = face_box(g, axis, dir < 3 ? 1 : n[axis]);
    BOX_LOOP(b) buf[k++] = f[VOXEL(x, y, z)].div_b_err;
  } else {
    int plane = dir < 3 ? 1 : n[axis] + 1;
    box_t b = face_box(g, axis, plane);
    BOX_LOOP(b) buf[k++] = FC(f, VOXEL(x, y, z), F_CBX, axis);
    for (int t = 1; t <= 2; t++) {
      int ca = (axis + t) % 3;
      box_t e = plane_box(g, axis, plane, ca, 1);
      BOX_LOOP(e) { int v = VOXEL(x, y, z); buf[k++] = FC(f, v, F_EX, ca); buf[k++] = FC(f, v, F_TCAX, ca); }
    }
  }
  return k;
}
double orc_unpack_msg(orc_field_t *f, const float *buf, const orc_grid_t *g, int kind, int dir) {
  const int nx = g->nx, ny = g->ny, n[3] = {g->nx, g->ny, g->nz};
  int axis = dir % 3, k = 0;
  double err = 0, w1, w2;
  if (kind == 0) {
    box_t b = node_box(g, axis, dir < 3 ? n[axis] + 1 : 0);
    BOX_LOOP(b) FC(f, VOXEL(x, y, z), F_EX, axis) = buf[k++];        /* rw = 1, lw = 0 */
  } else if (kind == 1) {
    box_t b = face_box(g, axis, dir < 3 ? n[axis] + 1 : 0);
    BOX_LOOP(b) f[VOXEL(x, y, z)].div_b_err = buf[k++];
  } else {
    int plane = dir < 3 ? n[axis] + 1 : 1;
    box_t b = face_box(g, axis, plane);
    BOX_LOOP(b) {
      int v = VOXEL(x, y, z);
      w1 = buf[k++]; w2 = FC(f, v, F_CBX, axis); FC(f, v, F_CBX, axis) = 0.5 * (w1 + w2); err += (w1 - w2) * (w1 - w2);
    }
    for (int t = 1; t <= 2; t++) {
      int ca = (axis + t) % 3;
      box_t e = plane_box(g, axis, plane, ca, 1);
      BOX_LOOP(e) {
        int v = VOXEL(x, y, z);
        w1 = buf[k++]; w2 = FC(f, v, F_EX, ca); FC(f, v, F_EX, ca) = 0.5 * (w1 + w2); err += (w1 - w2) * (w1 - w2);
        w1 = buf[k++]; w2 = FC(f, v, F_TCAX, ca); FC(f, v, F_TCAX, ca) = 0.5 * (w1 + w2);
      }
    }
  }
  return err;
}
/* the two local adjustments synchronize_tang_e_norm_b starts with (remote.c:309-310), and one axis
 * of it for a domain that shares both faces of the axis with itself */
void orc_local_adjust_tang_e_norm_b(orc_field_t *f, const orc_grid_t *g) { orc_local_adjust_tang_e(f, g); orc_local_adjust_norm_b(f, g); }
double orc_synchronize_tang_e_norm_b_self(orc_field_t *f, const orc_grid_t *g, int axis) {
  if (g->fbc[axis] != g->rank || g->fbc[axis + 3] != g->rank) return 0;
  int cnt = orc_msg_count(g, 2, axis);
  float *lo = (float *)malloc(sizeof(float) * (size_t)cnt), *hi = (float *)malloc(sizeof(float) * (size_t)cnt);
  orc_pack_msg(lo, f, g, 2, axis); orc_pack_msg(hi, f, g, 2, axis + 3);
  double err = orc_unpack_msg(f, lo, g, 2, axis);
  err += orc_unpack_msg(f, hi, g, 2, axis + 3);
  free(lo); free(hi);
  return err;
}

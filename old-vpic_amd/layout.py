"""Host-visible array-of-struct layouts of the hot path, as numpy dtypes.

These are the reference's own struct layouts (the drop-in boundary keeps them byte for byte):
  particle_t 48 B            src/species_advance/species_advance.h:28-34
  particle_mover_t 16 B      src/species_advance/species_advance.h:39-42
  particle_injector_t 48 B   src/species_advance/species_advance.h:48-55
  interpolator_t 80 B        src/sf_interface/sf_interface.h:45-58
  accumulator_t 48 B         src/sf_interface/sf_interface.h:68-77
  field_t 80 B               src/field_advance/field_advance.h:159-171
  material_coefficient 64 B  src/field_advance/standard/sfa_private.h:24-32
The C side pins the same sizes/offsets with static_assert (include/vpic_hip.h).
"""
import numpy as np

particle_t = np.dtype([("dx", "f4"), ("dy", "f4"), ("dz", "f4"), ("i", "i4"),
                       ("ux", "f4"), ("uy", "f4"), ("uz", "f4"), ("q", "f4"),
                       ("tag", "i8"), ("tag2", "i8")], align=True)
particle_mover_t = np.dtype([("dispx", "f4"), ("dispy", "f4"), ("dispz", "f4"), ("i", "i4")], align=True)
particle_injector_t = np.dtype([("dx", "f4"), ("dy", "f4"), ("dz", "f4"), ("i", "i4"),
                                ("ux", "f4"), ("uy", "f4"), ("uz", "f4"), ("q", "f4"),
                                ("dispx", "f4"), ("dispy", "f4"), ("dispz", "f4"), ("sp_id", "i4")], align=True)
interpolator_t = np.dtype([("ex", "f4"), ("dexdy", "f4"), ("dexdz", "f4"), ("d2exdydz", "f4"),
                           ("ey", "f4"), ("deydz", "f4"), ("deydx", "f4"), ("d2eydzdx", "f4"),
                           ("ez", "f4"), ("dezdx", "f4"), ("dezdy", "f4"), ("d2ezdxdy", "f4"),
                           ("cbx", "f4"), ("dcbxdx", "f4"), ("cby", "f4"), ("dcbydy", "f4"),
                           ("cbz", "f4"), ("dcbzdz", "f4"), ("_pad", "f4", (2,))], align=True)
accumulator_t = np.dtype([("jx", "f4", (4,)), ("jy", "f4", (4,)), ("jz", "f4", (4,))], align=True)
hydro_t = np.dtype([("jx", "f4"), ("jy", "f4"), ("jz", "f4"), ("rho", "f4"), ("px", "f4"), ("py", "f4"), ("pz", "f4"), ("ke", "f4"),
                    ("txx", "f4"), ("tyy", "f4"), ("tzz", "f4"), ("tyz", "f4"), ("tzx", "f4"), ("txy", "f4"), ("_pad", "f4", (2,))], align=True)
field_t = np.dtype([("ex", "f4"), ("ey", "f4"), ("ez", "f4"), ("div_e_err", "f4"),
                    ("cbx", "f4"), ("cby", "f4"), ("cbz", "f4"), ("div_b_err", "f4"),
                    ("tcax", "f4"), ("tcay", "f4"), ("tcaz", "f4"), ("rhob", "f4"),
                    ("jfx", "f4"), ("jfy", "f4"), ("jfz", "f4"), ("rhof", "f4"),
                    ("ematx", "u2"), ("ematy", "u2"), ("ematz", "u2"), ("nmat", "u2"),
                    ("fmatx", "u2"), ("fmaty", "u2"), ("fmatz", "u2"), ("cmat", "u2")], align=True)
material_coefficient_t = np.dtype([("decayx", "f4"), ("drivex", "f4"), ("decayy", "f4"), ("drivey", "f4"),
                                   ("decayz", "f4"), ("drivez", "f4"), ("rmux", "f4"), ("rmuy", "f4"),
                                   ("rmuz", "f4"), ("nonconductive", "f4"), ("epsx", "f4"), ("epsy", "f4"),
                                   ("epsz", "f4"), ("pad", "f4", (3,))], align=True)

assert particle_t.itemsize == 48 and particle_mover_t.itemsize == 16 and particle_injector_t.itemsize == 48
assert interpolator_t.itemsize == 80 and accumulator_t.itemsize == 48 and field_t.itemsize == 80
assert material_coefficient_t.itemsize == 64

# Field boundary codes of a face not shared with a domain (src/grid/grid.h:56-66) and particle
# boundary codes (src/grid/grid.h:68-69).
PEC_FIELDS, SYMMETRIC_FIELDS, PMC_FIELDS, ABSORB_FIELDS = -1, -2, -3, -4
REFLECT_PARTICLES, ABSORB_PARTICLES = -1, -2


def nv(nx, ny, nz):
    """Voxels of a local domain including the one-cell ghost layer."""
    return (nx + 2) * (ny + 2) * (nz + 2)


def voxel(x, y, z, nx, ny, nz):
    """FORTRAN voxel index (src/util/util_base.h:158-159)."""
    return x + (nx + 2) * (y + (ny + 2) * z)

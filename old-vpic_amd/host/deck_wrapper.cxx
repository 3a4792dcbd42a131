// deck_wrapper.cxx -- compiles an input deck of the reference INTO the vpic_simulation of
// vpic_hip_host.hxx, the way the reference's src/deck_wrapper.cxx:16-53,541 does: the deck's
// begin_* blocks become bodies of member functions, so deck code sees the members as free names.
//   -DINPUT_DECK=/abs/path/deck.cxx
#include "vpic_hip_host.hxx"

#define begin_globals struct user_global_t
#define global ((struct user_global_t *)user_global)

#define begin_initialization \
  void vpic_simulation::user_initialization(int num_cmdline_arguments, char **cmdline_argument)
#define begin_diagnostics void vpic_simulation::user_diagnostics(void)
#define begin_particle_injection void vpic_simulation::user_particle_injection(void)
#define begin_current_injection void vpic_simulation::user_current_injection(void)
#define begin_field_injection void vpic_simulation::user_field_injection(void)
#define begin_particle_collisions void vpic_simulation::user_particle_collisions(void)

#define repeat(count) for (int64_t _remain = (int64_t)(count); _remain; _remain--)

#define LOCAL_CELL_ID(x, y, z) INDEX_FORTRAN_3(x, y, z, 0, grid->nx + 1, 0, grid->ny + 1, 0, grid->nz + 1)
#define field(x, y, z) field[LOCAL_CELL_ID(x, y, z)]

#define sim_log_local(x) std::cerr << __FILE__ << "(" << __LINE__ << ")[" << rank() << "]: " << x << std::endl
#define sim_log(x) do { if (rank() == 0) { sim_log_local(x); std::cerr.flush(); } } while (0)

// region helpers (src/deck_wrapper.cxx:119-503): field initialisation over a region of space.
// rgn is a logical expression in x,y,z (cell-centred coordinates of the quantity being set).
#define set_region_field(rgn, EX, EY, EZ, BX, BY, BZ) do {                                          \
    const double _x0 = grid->x0, _y0 = grid->y0, _z0 = grid->z0;                                    \
    const double _dx = grid->dx, _dy = grid->dy, _dz = grid->dz, _c = grid->cvac;                   \
    for (int _k = 1; _k <= grid->nz + 1; _k++) for (int _j = 1; _j <= grid->ny + 1; _j++)           \
      for (int _i = 1; _i <= grid->nx + 1; _i++) {                                                  \
        double x, y, z;                                                                             \
        const double _xl = _x0 + _dx * (_i - 1), _yl = _y0 + _dy * (_j - 1), _zl = _z0 + _dz * (_k - 1); \
        const double _xc = _xl + 0.5 * _dx, _yc = _yl + 0.5 * _dy, _zc = _zl + 0.5 * _dz;           \
        x = _xc; y = _yl; z = _zl; if ((rgn) && _i <= grid->nx) field(_i, _j, _k).ex = (EX);        \
        x = _xl; y = _yc; z = _zl; if ((rgn) && _j <= grid->ny) field(_i, _j, _k).ey = (EY);        \
        x = _xl; y = _yl; z = _zc; if ((rgn) && _k <= grid->nz) field(_i, _j, _k).ez = (EZ);        \
        x = _xl; y = _yc; z = _zc; if ((rgn) && _j <= grid->ny && _k <= grid->nz) field(_i, _j, _k).cbx = _c * (BX); \
        x = _xc; y = _yl; z = _zc; if ((rgn) && _k <= grid->nz && _i <= grid->nx) field(_i, _j, _k).cby = _c * (BY); \
        x = _xc; y = _yc; z = _zl; if ((rgn) && _i <= grid->nx && _j <= grid->ny) field(_i, _j, _k).cbz = _c * (BZ); \
      }                                                                                             \
  } while (0)
#define everywhere 1

#define VPIC_HOST_STR2(x) #x
#define VPIC_HOST_STR(x) VPIC_HOST_STR2(x)
#include VPIC_HOST_STR(INPUT_DECK)

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

// region helpers (src/deck_wrapper.cxx:465-503): field initialisation over a region of space, with the
// reference's conventions exactly -- EVERY voxel including the ghost layers (0..n+1) is visited; for
// voxel (i,j,k) the "centre" coordinate is x0+dx*(i-0.5), the "edge" coordinate x0+dx*i and the "lower"
// one x0+dx*(i-1.5); a component is set when the region holds at the centre of this voxel or of the
// lower neighbours that share it; ex is evaluated at (centre, edge, edge), cbx at (edge, centre,
// centre), cyclically.  rgn is a logical expression in x, y, z.
#define set_region_field(rgn, EX, EY, EZ, BX, BY, BZ) do {                                          \
    const double _x0 = grid->x0, _y0 = grid->y0, _z0 = grid->z0;                                    \
    const double _dx = grid->dx, _dy = grid->dy, _dz = grid->dz, _c = grid->cvac;                   \
    for (int _k = 0; _k <= grid->nz + 1; _k++) for (int _j = 0; _j <= grid->ny + 1; _j++)           \
      for (int _i = 0; _i <= grid->nx + 1; _i++) {                                                  \
        const double _cx = _x0 + _dx * (_i - 0.5), _cy = _y0 + _dy * (_j - 0.5), _cz = _z0 + _dz * (_k - 0.5); \
        const double _lx = _x0 + _dx * (_i - 1.5), _ly = _y0 + _dy * (_j - 1.5), _lz = _z0 + _dz * (_k - 1.5); \
        const double _ex = _x0 + _dx * _i, _ey = _y0 + _dy * _j, _ez = _z0 + _dz * _k;              \
        double x, y, z;                                                                             \
        bool _in[8];                                      /* bit 0: lower x, bit 1: lower y, bit 2: lower z */ \
        for (int _b = 0; _b < 8; _b++) {                                                            \
          x = (_b & 1) ? _lx : _cx; y = (_b & 2) ? _ly : _cy; z = (_b & 4) ? _lz : _cz;             \
          _in[_b] = (rgn);                                                                          \
        }                                                                                           \
        vpic_field_t &_f = field(_i, _j, _k);                                                       \
        x = _cx; y = _ey; z = _ez; if (_in[0] || _in[2] || _in[4] || _in[6]) _f.ex = (EX);          \
        x = _ex; y = _cy; z = _ez; if (_in[0] || _in[4] || _in[1] || _in[5]) _f.ey = (EY);          \
        x = _ex; y = _ey; z = _cz; if (_in[0] || _in[1] || _in[2] || _in[3]) _f.ez = (EZ);          \
        x = _ex; y = _cy; z = _cz; if (_in[0] || _in[1]) _f.cbx = _c * (BX);                        \
        x = _cx; y = _ey; z = _cz; if (_in[0] || _in[2]) _f.cby = _c * (BY);                        \
        x = _cx; y = _cy; z = _ez; if (_in[0] || _in[4]) _f.cbz = _c * (BZ);                        \
      }                                                                                             \
  } while (0)
// set_region_material (src/deck_wrapper.cxx:228-278): the same 8 region samples per voxel; a SURFACE material
// is given to every component that touches the region, then a VOLUME material to the components that lie
// entirely inside it.  Either name may be unknown (NULL or ""): that part is skipped.
#define set_region_material(rgn, vname, sname) do {                                                 \
    const material_id _vmat = lookup_material((const char *)(vname)), _smat = lookup_material((const char *)(sname)); \
    const double _x0 = grid->x0, _y0 = grid->y0, _z0 = grid->z0, _dx = grid->dx, _dy = grid->dy, _dz = grid->dz; \
    for (int _k = 0; _k <= grid->nz + 1; _k++) for (int _j = 0; _j <= grid->ny + 1; _j++)           \
      for (int _i = 0; _i <= grid->nx + 1; _i++) {                                                  \
        double x, y, z;                                                                             \
        bool _in[8], _all = true, _any = false;           /* bit 0: lower x, bit 1: lower y, bit 2: lower z */ \
        for (int _b = 0; _b < 8; _b++) {                                                            \
          x = _x0 + _dx * (_i - ((_b & 1) ? 1.5 : 0.5)); y = _y0 + _dy * (_j - ((_b & 2) ? 1.5 : 0.5)); \
          z = _z0 + _dz * (_k - ((_b & 4) ? 1.5 : 0.5));                                            \
          _in[_b] = (rgn); _all = _all && _in[_b]; _any = _any || _in[_b];                          \
        }                                                                                           \
        vpic_field_t &_f = field(_i, _j, _k);                                                       \
        if (_smat != invalid_material_id) {                                                         \
          if (_in[0] || _in[2] || _in[4] || _in[6]) _f.ematx = _smat;                               \
          if (_in[0] || _in[4] || _in[1] || _in[5]) _f.ematy = _smat;                               \
          if (_in[0] || _in[1] || _in[2] || _in[3]) _f.ematz = _smat;                               \
          if (_in[0] || _in[1]) _f.fmatx = _smat;                                                   \
          if (_in[0] || _in[2]) _f.fmaty = _smat;                                                   \
          if (_in[0] || _in[4]) _f.fmatz = _smat;                                                   \
          if (_any) _f.nmat = _smat;                                                                \
        }                                                                                           \
        if (_vmat != invalid_material_id) {                                                         \
          if (_in[0] && _in[2] && _in[4] && _in[6]) _f.ematx = _vmat;                               \
          if (_in[0] && _in[4] && _in[1] && _in[5]) _f.ematy = _vmat;                               \
          if (_in[0] && _in[1] && _in[2] && _in[3]) _f.ematz = _vmat;                               \
          if (_in[0] && _in[1]) _f.fmatx = _vmat;                                                   \
          if (_in[0] && _in[2]) _f.fmaty = _vmat;                                                   \
          if (_in[0] && _in[4]) _f.fmatz = _vmat;                                                   \
          if (_all) _f.nmat = _vmat;                                                                \
          if (_in[0]) _f.cmat = _vmat;                                                              \
        }                                                                                           \
      }                                                                                             \
  } while (0)
// define_surface_emitter / define_volume_emitter (src/deck_wrapper.cxx:346-463): the faces through which one
// steps from a cell whose centre is outside the region into a neighbour whose centre is inside it / the
// cells whose centre is inside it.  A surface component names the cell OUTSIDE and the face towards the region.
#define VPIC_HOST_EMITTER_SCAN(rgn, BODY) do {                                                       \
    for (int _k = 1; _k <= grid->nz; _k++) for (int _j = 1; _j <= grid->ny; _j++) for (int _i = 1; _i <= grid->nx; _i++) { \
      double x, y, z;                                                                               \
      const double _cx = grid->x0 + grid->dx * (_i - 0.5), _cy = grid->y0 + grid->dy * (_j - 0.5), _cz = grid->z0 + grid->dz * (_k - 0.5); \
      bool _in[7];                                     /* centre, then the -x -y -z +x +y +z neighbours' centres */ \
      static const int _off[7][3] = {{0, 0, 0}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; \
      for (int _b = 0; _b < 7; _b++) {                                                              \
        x = _cx + grid->dx * _off[_b][0]; y = _cy + grid->dy * _off[_b][1]; z = _cz + grid->dz * _off[_b][2]; \
        _in[_b] = (rgn);                                                                            \
      }                                                                                             \
      BODY                                                                                          \
    }                                                                                               \
  } while (0)
#define define_surface_emitter(name, sp, emission_model, rgn) do {                                   \
    static const int _code[7] = {0, BOUNDARY(-1, 0, 0), BOUNDARY(0, -1, 0), BOUNDARY(0, 0, -1), BOUNDARY(1, 0, 0), BOUNDARY(0, 1, 0), BOUNDARY(0, 0, 1)}; \
    int _nf = 0;                                                                                    \
    VPIC_HOST_EMITTER_SCAN(rgn, for (int _b = 1; _b < 7; _b++) if (!_in[0] && _in[_b]) _nf++;);     \
    emitter_t *_emit = new_emitter((name), (sp), (emission_model_t)(emission_model), _nf, &emitter_list); \
    if (!_emit) break;                                                                              \
    _emit->n_component = _nf;                                                                       \
    _nf = 0;                                                                                        \
    VPIC_HOST_EMITTER_SCAN(rgn, for (int _b = 1; _b < 7; _b++) if (!_in[0] && _in[_b])              \
                                  _emit->component[_nf++] = COMPONENT_ID(LOCAL_CELL_ID(_i, _j, _k), _code[_b]);); \
  } while (0)
#define define_volume_emitter(name, sp, emission_model, rgn) do {                                    \
    int _nc = 0;                                                                                    \
    VPIC_HOST_EMITTER_SCAN(rgn, if (_in[0]) _nc++;);                                                \
    emitter_t *_emit = new_emitter((name), (sp), (emission_model_t)(emission_model), _nc, &emitter_list); \
    if (!_emit) break;                                                                              \
    _emit->n_component = _nc;                                                                       \
    _nc = 0;                                                                                        \
    VPIC_HOST_EMITTER_SCAN(rgn, if (_in[0]) _emit->component[_nc++] = COMPONENT_ID(LOCAL_CELL_ID(_i, _j, _k), BOUNDARY(0, 0, 0));); \
  } while (0)
#define everywhere 1

// at most NUM_TURNSTILES ranks inside the bracket at a time (src/deck_wrapper.cxx:505-533): ranks are cut
// into runs of `stride`; inside a run each rank waits for its predecessor's token
#define begin_turnstile(NUM_TURNSTILES) do {                                                         \
    int _stride = (int)(nproc() / (double)(NUM_TURNSTILES));                                         \
    if ((int)nproc() % (int)(NUM_TURNSTILES) > 0) _stride++;                                         \
    if (_stride > nproc()) _stride = (int)nproc();                                                   \
    int _token = 0, _me = (int)rank();                                                               \
    if (nproc() != 1 && _me % _stride != 0) mp_recv_i(&_token, 1, _me - 1, grid->mp);
#define end_turnstile                                                                                \
    if (nproc() != 1 && _me % _stride != _stride - 1 && _me != nproc() - 1) mp_send_i(&_token, 1, _me + 1, grid->mp); \
  } while (0)

// deck code that hands a mirror array straight to fwrite (demand-mode mirrors, vpic_hip_host.cxx)
#define fwrite vpic_host_fwrite

#define VPIC_HOST_STR2(x) #x
#define VPIC_HOST_STR(x) VPIC_HOST_STR2(x)
#include VPIC_HOST_STR(INPUT_DECK)

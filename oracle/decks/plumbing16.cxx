// plumbing16.cxx -- an input deck for THE REFERENCE (compiled by oracle/Makefile `deck` target into
// oracle/_ref/plumbing16.exe; test infrastructure, authored for this repo -- the reference ships no
// decks of this kind).  BASELINE.json configs[0]: 16x16x16 periodic box, 1 electron species, 8 ppc,
// 50 steps, cvac = eps0 = 1, unit cells, dt = 0.95 Courant, vacuum, standard field advance.
//
// Particles are loaded WITHOUT the random number generator: positions and momenta are plain
// double-precision arithmetic on (global cell, k), mirrored operation by operation in
// oracle/deck16.py, so that the engine under test starts from bit-identical particles.
// Energies are written every step at full precision; the final fields and particles are dumped raw.

// -DCLEAN_INTERVAL=k (DECK_DEFS of the two deck Makefiles) turns on divergence cleaning of E and B
// and the shared-face synchronisation every k steps (advance.cxx:151-208); default off.
#ifndef CLEAN_INTERVAL
#define CLEAN_INTERVAL 0
#endif
// -DABSORBING: open box -- every outer face absorbs fields (Higdon) and particles (their charge goes to rhob).
// -DEMITTER: a child-langmuir surface emitter (define_surface_emitter) in a uniform E_z.
// -DREFLUX: conducting z walls with the maxwellian_reflux particle boundary handler.
// -DANTENNA: begin_field_injection adds a driven E_y on the x = 0 face every step (a deck hook that WRITES fields).
// -DINJECT: 24 more particles every step from begin_particle_injection (inject_particle while the run is under way).
// -DMATERIALS: a dielectric slab and a block of anisotropic conductor (define_material, set_region_material).
// -DRESTART_AT=k: write restart files at step k.
// -DWRITE_DUMPS: also write the binary V0 dumps (dump_fields, dump_hydro, dump_particles) at step 10,
// the text / grid dumps at start-up, and the strided field_dump / hydro_dump files (banded and
// interleaved) with their .vpc global header.

begin_globals {
  int unused;
#ifdef WRITE_DUMPS
  DumpParameters fd_band, fd_inter, fd_full, hd_band, hd_inter, hd_full;
#endif
};

static inline double frac( double t ) { return t - floor(t); }

begin_initialization {
  const int n = 16, ppc = 8;
  const double len = 16;

  num_step             = 50;
  status_interval      = 0;
  clean_div_e_interval = CLEAN_INTERVAL;
  clean_div_b_interval = CLEAN_INTERVAL;
  sync_shared_interval = CLEAN_INTERVAL;

  grid->cvac = 1;
  grid->eps0 = 1;
  grid->damp = 0;
  grid->dt   = 0.95*courant_length( len, len, len, n, n, n );
  // -DTOPO_Y=a -DTOPO_Z=b: bricks instead of x-slabs (nproc must be a multiple of a*b)
#ifndef TOPO_Y
#define TOPO_Y 1
#endif
#ifndef TOPO_Z
#define TOPO_Z 1
#endif
#ifdef ABSORBING
  define_absorbing_grid( 0, 0, 0, len, len, len, n, n, n, nproc()/( TOPO_Y*TOPO_Z ), TOPO_Y, TOPO_Z, absorb_particles );
#else
  define_periodic_grid( 0, 0, 0, len, len, len, n, n, n, nproc()/( TOPO_Y*TOPO_Z ), TOPO_Y, TOPO_Z );
#endif
#ifdef REFLUX
  // conducting z walls that re-emit every particle they catch with a wall Maxwellian (boundary/maxwellian_reflux.c)
  {
    maxwellian_reflux_t mr;
    memset( &mr, 0, sizeof(mr) );
    mr.ut_para[0] = 0.12; mr.ut_perp[0] = 0.08;          // species id 0
    const int reflux = add_boundary( grid, maxwellian_reflux, &mr );
    set_domain_field_bc( BOUNDARY(0,0,-1), pec_fields ); set_domain_field_bc( BOUNDARY(0,0,1), pec_fields );
    set_domain_particle_bc( BOUNDARY(0,0,-1), reflux );  set_domain_particle_bc( BOUNDARY(0,0,1), reflux );
  }
#endif
  define_material( "vacuum", 1 );
#ifdef MATERIALS
  define_material( "glass", 2.5, 1.2, 0 );                              // dielectric / magnetic
  define_material( "lossy", 1.0, 1.3, 0.9, 1, 1, 1.1, 0.7, 0.3, 1.1 );  // anisotropic conductor
#endif
  finalize_field_advance( standard_field_advance );
#ifdef EMITTER
  // a cathode: the slab z < 2 is "inside"; the faces of the cells just above it emit electrons as long as E_z
  // pulls them out (child-langmuir law, src/emitter/child-langmuir.c), two per face and step
  set_region_field( everywhere, 0, 0, -0.3, 0, 0, 0 );
#endif
#ifdef MATERIALS
  set_region_material( x>4 && x<8, "glass", "glass" );
  set_region_material( x>10 && x<13 && y>3 && y<9 && z>2 && z<14, "lossy", "lossy" );
#endif

  species_t * electron = define_species( "electron", -1, 2*n*n*n*ppc/nproc(), -1, 20, 1 );

#ifdef EMITTER
  define_surface_emitter( "cathode", electron, child_langmuir, z<2 );
  if( find_emitter_name( "cathode", emitter_list ) ) {
    child_langmuir_t * cl = (child_langmuir_t *)find_emitter_name( "cathode", emitter_list )->model_parameters;
    cl->n_emit_per_face = 2; cl->ut_perp = 0.02; cl->ut_para = 0.05;
  }
#endif
  const double q = -0.01;
  for( int iz=0; iz<n; iz++ ) for( int iy=0; iy<n; iy++ ) for( int ix=0; ix<n; ix++ ) {
    const double c = (double)( ix + n*( iy + n*iz ) );
    for( int k=0; k<ppc; k++ ) {
      const double kk = (double)k;
      const double r1 = frac( c*0.7548776662466927 + kk*0.1234567891234567 + 0.03 );
      const double r2 = frac( c*0.5698402909980532 + kk*0.3456789123456789 + 0.41 );
      const double r3 = frac( c*0.3819660112501051 + kk*0.5678912345678912 + 0.77 );
      const double r4 = frac( c*0.6180339887498949 + kk*0.7891234567891234 + 0.19 );
      const double r5 = frac( c*0.2360679774997897 + kk*0.9123456789123456 + 0.63 );
      const double r6 = frac( c*0.4142135623730951 + kk*0.2345678912345678 + 0.87 );
      const double x  = ( (double)ix + r1 )*( len/(double)n );
      const double y  = ( (double)iy + r2 )*( len/(double)n );
      const double z  = ( (double)iz + r3 )*( len/(double)n );
      // two interpenetrating beams along x with a flat spread: crossings every step
      const double ux = ( (k&1) ? 0.3 : -0.3 ) + 0.2*( r4 - 0.5 );
      const double uy = 0.2*( r5 - 0.5 );
      const double uz = 0.2*( r6 - 0.5 );
      inject_particle( electron, x, y, z, ux, uy, uz, q, (int64_t)( c*ppc + kk ), 0, 0 );   // tag = global particle number
    }
  }
#ifdef WRITE_DUMPS
  dump_species( "species16.txt" );
  dump_materials( "materials16.txt" );
  dump_grid( "grid16" );
  struct setup { static void go( DumpParameters & p, const char * base, DumpFormat fmt, size_t sx, size_t sy, size_t sz ) {
    p.format = fmt; p.stride_x = sx; p.stride_y = sy; p.stride_z = sz;
    sprintf( p.baseDir, "%s", "." ); sprintf( p.baseFileName, "%s", base ); } };
  setup::go( global->fd_band,  "fband",  band,            2, 4, 1 );
  setup::go( global->fd_inter, "finter", band_interleave, 4, 2, 8 );
  setup::go( global->fd_full,  "ffull",  band,            1, 1, 1 );
  setup::go( global->hd_band,  "hband",  band,            2, 4, 1 );
  setup::go( global->hd_inter, "hinter", band_interleave, 4, 2, 8 );
  setup::go( global->hd_full,  "hfull",  band_interleave, 1, 1, 1 );
  // (the globals block is zero bytes, never constructed: masks start empty)
  global->fd_band.output_variables( electric | magnetic | rhof | emat | cmat );
  global->fd_full.output_variables( all );
  global->hd_band.output_variables( current_density | ke_density | stress_tensor );
  std::vector<DumpParameters *> dp;
  dp.push_back( &global->fd_band );
  dp.push_back( &global->hd_band );
  global_header( "global16", dp );
#endif
}

begin_diagnostics {
  species_t * sp = species_list;
  double en_f[6], en_p;
  field_advance->method->energy_f( en_f, field_advance->f, field_advance->m, field_advance->g );
  en_p = energy_p( sp->p, sp->np, sp->q_m, interpolator, grid );
  if( rank()==0 ) {
    FILE * f = fopen( "energies16.txt", step==0 ? "w" : "a" );
    fprintf( f, "%i %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (int)step,
             en_f[0], en_f[1], en_f[2], en_f[3], en_f[4], en_f[5], en_p );
    fclose( f );
  }
#ifdef RESTART_AT
  if( step==RESTART_AT ) dump_restart( "restart16", 0 );   // then `deck.exe restart restart16` goes on from here
#endif
#ifdef WRITE_DUMPS
  if( step==10 ) {
    dump_fields( "fields16" );
    dump_hydro( "electron", "hydro16" );
    dump_particles( "electron", "particles16" );
    field_dump( global->fd_band );
    field_dump( global->fd_inter );
    field_dump( global->fd_full );
    hydro_dump( "electron", global->hd_band );
    hydro_dump( "electron", global->hd_inter );
    hydro_dump( "electron", global->hd_full );
  }
#endif
  if( step==0 || step==num_step ) {
    char name[64];
    sprintf( name, "state16_step%i_rank%i.bin", (int)step, (int)rank() );
    FILE * f = fopen( name, "wb" );
    int hdr[4] = { grid->nx, grid->ny, grid->nz, sp->np };
    fwrite( hdr, sizeof(int), 4, f );
    fwrite( field, sizeof(field_t), (grid->nx+2)*(grid->ny+2)*(grid->nz+2), f );
    fwrite( sp->p, sizeof(particle_t), sp->np, f );
    fclose( f );
  }
}

begin_particle_injection {
#ifdef INJECT
  // a beam fed in while the run is under way: 24 particles per step at positions every rank computes alike
  // (inject_particle keeps only those inside the local domain, misc.cxx:42-48)
  species_t * sp = species_list;
  for( int k=0; k<24; k++ ) {
    const double s = (double)step, kk = (double)k, len = 16;
    const double x = len*frac( s*0.3819660112501051 + kk*0.0411 + 0.013 );
    const double y = len*frac( s*0.2360679774997897 + kk*0.1733 + 0.291 );
    const double z = len*frac( s*0.4142135623730951 + kk*0.3571 + 0.577 );
    inject_particle( sp, x, y, z, 0.4, 0.05*( frac( kk*0.37 ) - 0.5 ), 0, -0.02, (int64_t)( 1000000 + 24*step + k ), ( k%3==2 ) ? 0.37 : 0, k&1 );   // every other one leaves its charge, negated, in rhob; every third has lived 0.37 of the step already
  }
#endif
}
begin_current_injection {}
begin_field_injection {
#ifdef ANTENNA
  // a sheet antenna on the global x = 0 face: the deck edits E in place, every step (advance.cxx:141)
  if( rank()==0 ) {
    const float drive = 0.05*sin( 0.35*step );
    for( int z=1; z<=grid->nz+1; z++ ) for( int y=1; y<=grid->ny; y++ ) field( 1, y, z ).ey += drive;
  }
#endif
}
begin_particle_collisions {}

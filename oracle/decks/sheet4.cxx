// sheet4.cxx -- an input deck for THE REFERENCE and for the HIP host (test infrastructure, authored for
// this repo).  A small force-free current sheet set up the way production reconnection decks of the
// reference are (decks/trecon-part is the model for WHICH interfaces a deck of that family uses; no
// text of it is reused): four plasma species split by sheet side, conducting walls and reflecting
// particles in z, periodic x and y, x-slab topology, initial B from set_region_field, divergence
// cleaning and shared-face synchronisation, energies / strided field_dump / hydro_dump output through
// DumpParameters, FileIO and a turnstile -- and two TRACER species that are taken off species_list
// and advanced from begin_particle_injection with direct calls of advance_p / boundary_p / sort_p.
//
// Normals are drawn by Box-Muller from uniform_rand() inside the deck, so the reference executable and
// the HIP host load bit-identical particles (maxwellian_rand differs between them).

// Sizes can be overridden from the build line (DECK_DEFS="-DSHEET_NX=128 ...") for timing runs; the
// golden fixtures are for the defaults.
#ifndef SHEET_NX
#define SHEET_NX 32
#endif
#ifndef SHEET_NY
#define SHEET_NY 8
#endif
#ifndef SHEET_NZ
#define SHEET_NZ 16
#endif
#ifndef SHEET_PPC
#define SHEET_PPC 16
#endif
#ifndef SHEET_STEPS
#define SHEET_STEPS 40
#endif
#define SHEET_DUMP_STEP 20

begin_globals {
  DumpParameters fd, hd;          // zero bytes, never constructed (as in the reference): masks start empty
  species_t * tracers;            // the species taken off species_list
  int dump_step;
};

#define DRAW_NORMAL( out, dev ) BEGIN_PRIMITIVE {                       \
    const double _u1 = uniform_rand( 0, 1 ), _u2 = uniform_rand( 0, 1 ); \
    (out) = (dev)*sqrt( -2*log( _u1 ) )*cos( 6.283185307179586*_u2 );   \
  } END_PRIMITIVE

begin_initialization {
  const double wpe_wce = 0.5, vthe = 0.25, vthi = 0.12, mi_me = 4, bg = 0.2;
  const double b0 = 1/wpe_wce;                       // me = c = e = eps0 = wpe = 1
  const double hx = 0.5, Lx = hx*SHEET_NX, Ly = hx*SHEET_NY, Lz = hx*SHEET_NZ, L = 1.5;
  const double nppc = SHEET_PPC;

  num_step             = SHEET_STEPS;
  status_interval      = 0;
  clean_div_e_interval = 10;
  clean_div_b_interval = 10;
  sync_shared_interval = 10;
  global->dump_step    = SHEET_DUMP_STEP;

  grid->cvac = 1;
  grid->eps0 = 1;
  grid->damp = 0.001;
  grid->dt   = 0.9*courant_length( Lx, Ly, Lz, SHEET_NX, SHEET_NY, SHEET_NZ );
  define_periodic_grid( 0, -0.5*Ly, -0.5*Lz, Lx, 0.5*Ly, 0.5*Lz, SHEET_NX, SHEET_NY, SHEET_NZ, nproc(), 1, 1 );
  set_domain_field_bc( BOUNDARY(0,0,-1), pec_fields );
  set_domain_field_bc( BOUNDARY(0,0, 1), pec_fields );
  set_domain_particle_bc( BOUNDARY(0,0,-1), reflect_particles );
  set_domain_particle_bc( BOUNDARY(0,0, 1), reflect_particles );

  double Ne = nppc*SHEET_NX*SHEET_NY*SHEET_NZ;
  Ne = trunc_granular( Ne, nproc() );
  const double qe = -Lx*Ly*Lz/Ne, qi = -qe;          // n0 = 1

  species_t * eT = define_species( "eT", -1,       2*Ne/nproc(), -1, 5, 1 );
  species_t * eB = define_species( "eB", -1,       2*Ne/nproc(), -1, 5, 1 );
  species_t * iT = define_species( "iT", 1/mi_me,  2*Ne/nproc(), -1, 8, 1 );
  species_t * iB = define_species( "iB", 1/mi_me,  2*Ne/nproc(), -1, 8, 1 );
  species_t * eR = define_species( "eR", -1,       2*Ne/nproc(), -1, 5, 1 );
  species_t * iR = define_species( "iR", 1/mi_me,  2*Ne/nproc(), -1, 8, 1 );
  // the two tracer species were defined last, so they lead species_list: cut them off
  global->tracers    = species_list;
  species_list       = species_list->next->next;
  global->tracers->next->next = NULL;

  define_material( "vacuum", 1 );
  finalize_field_advance( standard_field_advance );

#define SHEET_BX ( b0*tanh( z/L ) )
#define SHEET_BY ( sqrt( b0*b0*( 1 + bg*bg ) - SHEET_BX*SHEET_BX ) )
  set_region_field( everywhere, 0, 0, 0, SHEET_BX, SHEET_BY, 0 );

  seed_rand( 7*nproc() + rank() );
  const double xmin = grid->x0, xmax = grid->x0 + grid->dx*grid->nx;
  const double ymin = grid->y0, ymax = grid->y0 + grid->dy*grid->ny;
  const double zmin = grid->z0, zmax = grid->z0 + grid->dz*grid->nz;
  int64_t count = 0;
  repeat( Ne/nproc() ) {
    const double x = uniform_rand( xmin, xmax ), y = uniform_rand( ymin, ymax ), z = uniform_rand( zmin, zmax );
    // the current that supports the rotating field is carried half by each sign of charge
    const double sech = 1/cosh( z/L ), jy = -0.5*( b0/L )*sech*sech, jx = jy*SHEET_BX/SHEET_BY;
    double ux, uy, uz;
    const int64_t tag = ( ( (int64_t)rank() )<<40 ) | ( ++count );
    DRAW_NORMAL( ux, vthe ); DRAW_NORMAL( uy, vthe ); DRAW_NORMAL( uz, vthe );
    ux += jx; uy += jy;
    species_t * se = z>0 ? eT : eB;
    inject_particle( se, x, y, z, ux, uy, uz, qe, tag, 0, 0 );
    if( count%4==0 ) {                               // every fourth electron is also a tracer (charge 0: no back-reaction)
      particle_t * t = eR->p + ( eR->np++ );
      *t = se->p[ se->np-1 ];
      t->q = 0;
    }
    DRAW_NORMAL( ux, vthi ); DRAW_NORMAL( uy, vthi ); DRAW_NORMAL( uz, vthi );
    ux -= jx; uy -= jy;
    species_t * si = z>0 ? iT : iB;
    inject_particle( si, x, y, z, ux, uy, uz, qi, tag, 0, 0 );
    if( count%8==0 ) {
      particle_t * t = iR->p + ( iR->np++ );
      *t = si->p[ si->np-1 ];
      t->q = 0;
    }
  }

  global->fd.format = band;
  global->fd.stride_x = 2; global->fd.stride_y = 1; global->fd.stride_z = 2;
  sprintf( global->fd.baseDir, "fields" ); sprintf( global->fd.baseFileName, "fields" );
  global->fd.output_variables( electric | magnetic | current );
  global->hd.format = band;
  global->hd.stride_x = 1; global->hd.stride_y = 1; global->hd.stride_z = 1;
  sprintf( global->hd.baseDir, "hydro" ); sprintf( global->hd.baseFileName, "eThydro" );
  global->hd.output_variables( current_density | charge_density );
}

begin_diagnostics {
  if( step==0 ) {
    dump_mkdir( "fields" );
    dump_mkdir( "hydro" );
    dump_mkdir( "rundata" );
    dump_grid( "rundata/grid" );
    dump_materials( "rundata/materials" );
    dump_species( "rundata/species" );
    std::vector<DumpParameters *> params;
    params.push_back( &global->fd );
    params.push_back( &global->hd );
    global_header( "global", params );
  }

  double en_f[6], en_p[4];
  int n = 0;
  species_t * sp;
  field_advance->method->energy_f( en_f, field_advance->f, field_advance->m, field_advance->g );
  LIST_FOR_EACH( sp, species_list ) en_p[n++] = energy_p( sp->p, sp->np, sp->q_m, interpolator, grid );
  if( rank()==0 ) {
    FileIO out;
    if( out.open( "energies4.txt", step==0 ? io_write : io_append )!=ok ) ERROR(( "Cannot open file." ));
    out.print( "%i %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (int)step,
               en_f[0], en_f[1], en_f[2], en_f[3], en_f[4], en_f[5], en_p[0], en_p[1], en_p[2], en_p[3] );
    out.close();
  }
  if( step%10==0 ) dump_energies( "rundata/energies", step==0 ? 0 : 1 );

  if( step==global->dump_step ) {
    begin_turnstile( 1 );          // one writer at a time; nothing collective may sit inside the bracket
    field_dump( global->fd );
    end_turnstile;
    hydro_dump( "eT", global->hd );   // sums moments across ranks: outside
  }

  if( step==num_step ) {
    // the tracers (tag + state) and the fields at the end, raw
    char name[64];
    sprintf( name, "tracers4_rank%i.bin", (int)rank() );
    FileIO out;
    if( out.open( name, io_write )!=ok ) ERROR(( "Cannot open file." ));
    for( sp=global->tracers; sp; sp=sp->next ) {
      out.write( &sp->np, 1 );
      out.write( sp->p, sp->np );
    }
    out.close();
    sprintf( name, "fields4_rank%i.bin", (int)rank() );
    if( out.open( name, io_write )!=ok ) ERROR(( "Cannot open file." ));
    out.write( field, (size_t)( grid->nx+2 )*( grid->ny+2 )*( grid->nz+2 ) );
    LIST_FOR_EACH( sp, species_list ) out.write( &sp->np, 1 );
    out.close();
  }
}

begin_particle_injection {
  // tracers: not on species_list, so vpic_simulation::advance does not push them; the deck does, with
  // the reference's own L3 calls (advance_p finishes in-domain moves, boundary_p the rest)
  static accumulator_t * scratch = NULL;
  if( !scratch ) scratch = new_accumulators( grid );
  for( species_t * s=global->tracers; s; s=s->next ) {
    s->nm += advance_p( s->p, s->np, s->q_m, s->pm, s->max_nm, scratch, interpolator, grid );
    for( int pass=0; pass<3; pass++ ) boundary_p( s, field, scratch, grid, rng );
    if( step%s->sort_interval==0 ) sort_p( s, grid );
  }
}

begin_current_injection {}
begin_field_injection {}
begin_particle_collisions {}

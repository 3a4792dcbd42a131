// twostream.cxx -- an input deck for THE REFERENCE (test infrastructure, authored for this repo): the
// workload of bench.py (BASELINE.json configs[1]) cut down to one 24^3 block of it per rank, for timing
// the reference's own scalar code on the host cores of the GPU box -- `cpu_baseline` with kind
// "reference".  Periodic box of (24 * nproc) x 24 x 24 unit cells, x-slab topology, two electron beams
// (u_x = +-0.2, thermal spread 0.02 c per component), PPC particles per cell and species, dt = 0.95
// Courant, sort every 10 steps, no cleaning, no diagnostics.  The SAME FILE compiled against the HIP host with
// -DTS_N=128 is bench.py's `deck_host` figure (configs[1] through the deck API).
//   mpiexec -n C twostream.exe -tpp=1 <steps>        (main.cxx reports "simulation time" on rank 0)
#ifndef TS_PPC
#define TS_PPC 32
#endif
#ifndef TS_N
#define TS_N 24    // cells per side of a rank's block (-DTS_N=128 -DTS_PPC=32: BASELINE configs[1] whole, for the HIP host)
#endif

begin_globals { int unused; };

begin_initialization {
  const int n = TS_N, ppc = TS_PPC;
  const double wp_dt = 0.2;
  num_step        = num_cmdline_arguments>1 ? atoi( cmdline_argument[1] ) : 40;
  status_interval = 0;
  grid->cvac = 1; grid->eps0 = 1; grid->damp = 0;
  grid->dt   = 0.95*courant_length( n, n, n, n, n, n );
  define_periodic_grid( 0, 0, 0, n*nproc(), n, n, n*nproc(), n, n, nproc(), 1, 1 );
  define_material( "vacuum", 1 );
  finalize_field_advance( standard_field_advance );
  const double np_local = (double)n*n*n*ppc;
  const double q = -( wp_dt/grid->dt )*( wp_dt/grid->dt )/( 2*ppc );      // wp^2 = n |q|, both beams together
  species_t * beam[2];
  beam[0] = define_species( "right", -1, 1.2*np_local, -1, 10, 1 );
  beam[1] = define_species( "left",  -1, 1.2*np_local, -1, 10, 1 );
  seed_rand( 1 + rank() );
  const double x0 = grid->x0, x1 = grid->x0 + grid->dx*grid->nx;
  for( int s=0; s<2; s++ ) repeat( np_local )
    inject_particle( beam[s], uniform_rand( x0, x1 ), uniform_rand( 0, n ), uniform_rand( 0, n ),
                     ( s ? -0.2 : 0.2 ) + maxwellian_rand( 0.02 ), maxwellian_rand( 0.02 ), maxwellian_rand( 0.02 ), q, 0, 0, 0 );
}

begin_diagnostics {}
begin_particle_injection {}
begin_current_injection {}
begin_field_injection {}
begin_particle_collisions {}

/* config.h for decks/trecon-part/turbulence.cxx (TEST INFRASTRUCTURE, authored for this repo): the deck reads all
 * its sizes from a file of this name next to it.  oracle/Makefile target `trecon` puts this file and symbolic
 * links to the reference's turbulence.cxx / tracer.cxx / energy.cxx into one directory and builds the deck,
 * UNCHANGED, twice: as the reference executable and against the HIP host.  Small box, few steps: a smoke-and-
 * parity run of the production deck by default; every size can be overridden from the build line
 * (make -C oracle trecon TOPO=1 NAME=big EXTRA="-DVPIC_PARTICLE_X=64 ...") for timing runs. */
#define QUIET_RUN
#define VPIC_FILE_PER_PARTICLE 0
#ifndef VPIC_TIMESTEPS
#define VPIC_TIMESTEPS 40
#endif
#ifndef VPIC_DUMPS
#define VPIC_DUMPS 2
#endif
#define VPIC_DUMP_INTERVAL (VPIC_TIMESTEPS / VPIC_DUMPS)
#ifndef VPIC_TOPOLOGY_X
#define VPIC_TOPOLOGY_X 1
#endif
#define VPIC_TOPOLOGY_Y 1
#define VPIC_TOPOLOGY_Z 1
#ifndef VPIC_PARTICLE_X
#define VPIC_PARTICLE_X 32
#endif
#ifndef VPIC_PARTICLE_Y
#define VPIC_PARTICLE_Y 8
#endif
#ifndef VPIC_PARTICLE_Z
#define VPIC_PARTICLE_Z 16
#endif

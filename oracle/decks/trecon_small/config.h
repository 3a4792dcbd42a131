/* config.h for decks/trecon-part/turbulence.cxx (TEST INFRASTRUCTURE, authored for this repo): the deck reads all
 * its sizes from a file of this name next to it.  oracle/Makefile target `trecon` puts this file and symbolic
 * links to the reference's turbulence.cxx / tracer.cxx / energy.cxx into one directory and builds the deck,
 * UNCHANGED, twice: as the reference executable and against the HIP host.  Small box, few steps: a smoke-and-
 * parity run of the production deck, not a benchmark. */
#define QUIET_RUN
#define VPIC_FILE_PER_PARTICLE 0
#define VPIC_TIMESTEPS 40
#define VPIC_DUMPS     2
#define VPIC_DUMP_INTERVAL (VPIC_TIMESTEPS / VPIC_DUMPS)
#ifndef VPIC_TOPOLOGY_X
#define VPIC_TOPOLOGY_X 1
#endif
#define VPIC_TOPOLOGY_Y 1
#define VPIC_TOPOLOGY_Z 1
#define VPIC_PARTICLE_X 32
#define VPIC_PARTICLE_Y 8
#define VPIC_PARTICLE_Z 16

// main.cxx -- driver of a deck executable on the HIP host (the reference's src/main.cxx:24-122
// without pipeline dispatchers and MPI: one process, one GPU, one domain).
#include "vpic_hip_host.hxx"
#include <chrono>

int main(int argc, char **argv) {
  int m = 0;
  for (int n = 0; n < argc; n++)                      // -tpp=N is accepted and ignored (no host pipelines)
    if (strncmp(argv[n], "-tpp=", 5) != 0) argv[m++] = argv[n];
  argv[m] = NULL; argc = m;
  vpic_simulation simulation;
  if (argc >= 3 && strcmp(argv[1], "restart") == 0) ERROR(("restart is not supported by this host yet"));
  simulation.initialize(argc, argv);
  MESSAGE(("**** Beginning simulation advance on the HIP engine ****"));
  const auto t0 = std::chrono::steady_clock::now();
  while (simulation.advance());
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  MESSAGE(("simulation time: %lf\n", dt));
  simulation.finalize();
  MESSAGE(("Maximum number of time steps reached.  Job has completed."));
  return 0;
}

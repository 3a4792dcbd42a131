// main.cxx -- driver of a deck executable on the HIP host (the reference's src/main.cxx:24-122
// without pipeline dispatchers; one process = one domain = one GPU, several of them under mpiexec
// when built with MPI=1).
#include "vpic_hip_host.hxx"
#include <chrono>

int main(int argc, char **argv) {
  vpic_host_mp_init(&argc, &argv);                     // mp_init, src/main.cxx:63
  int m = 0;
  for (int n = 0; n < argc; n++)                      // -tpp=N is accepted and ignored (no host pipelines)
    if (strncmp(argv[n], "-tpp=", 5) != 0) argv[m++] = argv[n];
  argv[m] = NULL; argc = m;
  vpic_simulation simulation;
  if (argc >= 3 && strcmp(argv[1], "restart") == 0) simulation.restart(argv[2]);   // src/main.cxx:83-86
  else simulation.initialize(argc, argv);
  const bool talk = vpic_host_mp_rank() == 0;
  if (talk) MESSAGE(("**** Beginning simulation advance on the HIP engine ****"));
  const auto t0 = std::chrono::steady_clock::now();
  while (simulation.advance());
  simulation.sync();                                  // (the time step is asynchronous: what was enqueued has to have happened)
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (talk) MESSAGE(("simulation time: %lf\n", dt));
  simulation.finalize();
  if (talk) MESSAGE(("Maximum number of time steps reached.  Job has completed."));
  vpic_host_mp_finalize();
  return 0;
}

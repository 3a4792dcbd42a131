// What does HBM deliver for the push kernel's traffic shape?  8 SoA arrays read, 6 written,
// np elements, wave-contiguous 512-element spans as in advance_p_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Arr { float *a[8]; };
template <int VEC, int NW>   // VEC floats per lane per access; NW arrays written (0..6)
__global__ __launch_bounds__(256) void k(Arr p, int np, float s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long first = ((long)blockIdx.x * 4 + wave) * 512;
  for (int it = 0; it < 512 / (64 * VEC); it++) {
    const long i = first + (long)it * 64 * VEC + lane * VEC;
    if (i >= np) break;
    float v[8][VEC];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (VEC == 4) { float4 t = *reinterpret_cast<const float4 *>(p.a[c] + i); v[c][0] = t.x; v[c][1 % VEC] = t.y; v[c][2 % VEC] = t.z; v[c][3 % VEC] = t.w; }
      else v[c][0] = p.a[c][i];
    }
#pragma unroll
    for (int c = 0; c < NW; c++) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) o[j] = v[c][j] * s + v[7][j] + v[6][j];
      if (VEC == 4) *reinterpret_cast<float4 *>(p.a[c] + i) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
      else p.a[c][i] = o[0];
    }
    if (NW == 0) { float t = 0; for (int c = 0; c < 8; c++) for (int j = 0; j < VEC; j++) t += v[c][j]; if (t == 1.2345f) p.a[0][i] = t; }
  }
}
template <int VEC, int NW> void run(Arr p, int np) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int nb = (np + 2047) / 2048;
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<VEC, NW>), dim3(nb), dim3(256), 0, 0, p, np, 1.0001f);
  hipEventRecord(a, 0);
  const int reps = 10;
  for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k<VEC, NW>), dim3(nb), dim3(256), 0, 0, p, np, 1.0001f);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double bytes = (double)np * 4 * (8 + NW);
  printf("VEC=%d read 8 write %d: %.3f ms  %.2f TB/s\n", VEC, NW, ms, bytes / ms * 1e-9);
}
int main() {
  const int np = 67108864;
  Arr p;
  for (int c = 0; c < 8; c++) { hipMalloc(&p.a[c], (size_t)np * 4 + 8192); hipMemset(p.a[c], 0, (size_t)np * 4); }
  run<1, 0>(p, np); run<1, 3>(p, np); run<1, 6>(p, np);
  run<4, 0>(p, np); run<4, 3>(p, np); run<4, 6>(p, np);
  return 0;
}

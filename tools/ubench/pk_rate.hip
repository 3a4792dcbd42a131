// Issue-rate probe: scalar f32 VALU vs packed (v_pk_*) f32 on gfx950, no MFMA in sight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = {a1, a0}, b5 = {a3, a2}, b6 = {a5, a4}, b7 = {a7, a6};
  const float c = 1.0000001f; const f2 c2 = {c, c};
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {   // 16 scalar ops (8 mul + 8 add)
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 1) {  // 16 packed ops (8 pk_mul + 8 pk_add) = 2x the flops
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c2));
    } else if (MODE == 2) {  // 16 v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else {                 // 16 v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c2));
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b0.y + b1.x + b1.y + b2.x + b2.y + b3.x + b3.y + b4.x + b4.y + b5.x + b5.y + b6.x + b6.y + b7.x + b7.y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> static void run(const char *name, int wgs, float *d) {
  const int iters = 300000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, 100);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr = (double)iters * 16 * wgs * 4;          // wave-instructions
  printf("%-14s wgs=%5d  %.3f ms  %.1f G wave-instr/s  (%.2f clk/instr/SIMD at 2.4 GHz, %d waves/SIMD)\n", name, wgs, ms,
         instr / ms * 1e-6, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (wgs / 256.0)), wgs / 256);
}
int main() {
  float *d; hipMalloc(&d, 256 * 4096 * 4);
  for (int rep = 0; rep < 2; rep++) for (int w : {256, 1024, 2048}) {
    run<0>("mul/add f32", w, d); run<1>("pk_mul/pk_add", w, d); run<2>("fma f32", w, d); run<3>("pk_fma f32", w, d);
  }
  return 0;
}

// What do the cell-crossers' late, scattered stores cost the memory system?  The push kernel's traffic shape -- per wavefront
// pass 64 consecutive elements of 8 arrays read and 6 written -- plus, every 8 passes, 64 "late" stores into elements of
// those 8 passes (12.5 % of the elements, what the two-stream deck's crossers are): as four 4-byte stores into four arrays
// (the struct-of-arrays layout), as one 16-byte store into a float4 array (position + cell as one record), or not at all.
// No arithmetic: what differs between the variants is what HBM sees.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Arr { float *a[8]; float4 *r; };
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// LAYOUT 0: eight float arrays (read 8, write 6); 1: float4 record + ux,uy,uz,q (read r + 4, write r + 3)
// LATE 0: none; 1: the layout's late stores (4 x 4 B, or 1 x 16 B); 2: SoA but ONE 4-byte store only
template <int LAYOUT, int LATE>
__global__ __launch_bounds__(256) void k(Arr p, long np, float s, int passes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long first = ((long)blockIdx.x * 4 + wave) * 64 * passes;
  for (int it = 0; it < passes; it++) {
    const long i = first + (long)it * 64 + lane;
    if (i >= np) break;
    if (LAYOUT == 0) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = p.a[c][i];
#pragma unroll
      for (int c = 0; c < 6; c++) p.a[c][i] = v[c] * s + v[7] + v[6];
    } else {
      const float4 r = p.r[i];
      const float u0 = p.a[4][i], u1 = p.a[5][i], u2 = p.a[6][i], q = p.a[7][i];
      p.r[i] = make_float4(r.x * s + q, r.y * s + q, r.z * s + q, r.w);
      p.a[4][i] = u0 * s + q; p.a[5][i] = u1 * s + q; p.a[6][i] = u2 * s + q;
    }
    if (LATE && (it & 7) == 7) {
      // 64 elements of the last 8 passes, one per lane, pseudo-random
      const long j = first + (long)(it - 7) * 64 + (mix((unsigned)(i * 2654435761u)) & 511u);
      if (j < np) {
        if (LAYOUT == 0) { p.a[0][j] = s; if (LATE == 1) { p.a[1][j] = s; p.a[2][j] = s; p.a[3][j] = s; } }
        else p.r[j] = make_float4(s, s, s, s);
      }
    }
  }
}
template <int LAYOUT, int LATE> void run(Arr p, long np, const char *what) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int passes = 16;
  const unsigned nb = (unsigned)((np + 256 * passes - 1) / (256 * passes));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<LAYOUT, LATE>), dim3(nb), dim3(256), 0, 0, p, np, 1.0001f, passes);
  hipEventRecord(a, 0);
  const int reps = 5;
  for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k<LAYOUT, LATE>), dim3(nb), dim3(256), 0, 0, p, np, 1.0001f, passes);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double bytes = (double)np * (LAYOUT == 0 ? 56 : 60);
  printf("%-58s %8.3f ms  %.2f TB/s streaming\n", what, ms, bytes / ms * 1e-9);
}
int main(int argc, char **argv) {
  const long np = argc > 1 ? atol(argv[1]) : 268435456l;
  Arr p;
  for (int c = 0; c < 8; c++) { hipMalloc(&p.a[c], (size_t)np * 4 + 8192); hipMemset(p.a[c], 0, (size_t)np * 4); }
  hipMalloc(&p.r, (size_t)np * 16 + 8192); hipMemset(p.r, 0, (size_t)np * 16);
  printf("%ld elements\n", np);
  run<0, 0>(p, np, "8 arrays: read 8, write 6, no late stores");
  run<0, 1>(p, np, "8 arrays: + 12.5 % late stores, 4 x 4 B each");
  run<0, 2>(p, np, "8 arrays: + 12.5 % late stores, 1 x 4 B each");
  run<1, 0>(p, np, "float4 + 4 arrays: read 32 B, write 28 B, no late stores");
  run<1, 1>(p, np, "float4 + 4 arrays: + 12.5 % late stores, 1 x 16 B each");
  return 0;
}

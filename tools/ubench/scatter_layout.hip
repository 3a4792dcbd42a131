// What does a sort's scatter cost in the particle layouts under consideration?  np records are
// moved to dst[i], a permutation of the kind a counting sort by cell produces after one step of a
// hot plasma (a fraction `hot` of the particles has moved to one of the 6 neighbouring cells).
//   soa8   : 8 arrays of 4 bytes          (the engine's layout today)
//   f4x2   : 2 arrays of 16 bytes         (position+cell, momentum+charge)
//   aos32  : 1 array of 32 bytes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
struct Soa { float *a[8]; };
__global__ __launch_bounds__(256) void k_soa8(Soa in, Soa out, const int *__restrict__ dst, int np) {
  const int i = blockIdx.x * 256 + threadIdx.x; if (i >= np) return;
  const int d = dst[i];
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; c++) v[c] = in.a[c][i];
#pragma unroll
  for (int c = 0; c < 8; c++) out.a[c][d] = v[c];
}
__global__ __launch_bounds__(256) void k_f4x2(const float4 *__restrict__ i0, const float4 *__restrict__ i1, float4 *o0, float4 *o1, const int *__restrict__ dst, int np) {
  const int i = blockIdx.x * 256 + threadIdx.x; if (i >= np) return;
  const int d = dst[i];
  const float4 a = i0[i], b = i1[i];
  o0[d] = a; o1[d] = b;
}
// the same move as a gather: first the inverse permutation (one scattered 4-byte store per record), then every output
// slot fetches its record from wherever it is (scattered 4-byte loads, coalesced stores)
__global__ __launch_bounds__(256) void k_inverse(const int *__restrict__ dst, int *__restrict__ src_of, int np) {
  const int i = blockIdx.x * 256 + threadIdx.x; if (i >= np) return;
  src_of[dst[i]] = i;
}
__global__ __launch_bounds__(256) void k_gather8(Soa in, Soa out, const int *__restrict__ src_of, int np) {
  const int j = blockIdx.x * 256 + threadIdx.x; if (j >= np) return;
  const int i = src_of[j];
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; c++) v[c] = in.a[c][i];
#pragma unroll
  for (int c = 0; c < 8; c++) out.a[c][j] = v[c];
}
struct Rec { float4 a, b; };
__global__ __launch_bounds__(256) void k_aos32(const Rec *__restrict__ in, Rec *out, const int *__restrict__ dst, int np) {
  const int i = blockIdx.x * 256 + threadIdx.x; if (i >= np) return;
  const int d = dst[i];
  const Rec r = in[i];
  out[d] = r;
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; w++) f();
  hipEventRecord(a, 0);
  for (int w = 0; w < 5; w++) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main(int argc, char **argv) {
  const int n = 128, ppc = 32; const long ncell = (long)n * n * n; const int np = (int)(ncell * ppc);
  const double hot = argc > 1 ? atof(argv[1]) : 0.6;
  // keys: cell-sorted, then a fraction moves to a neighbour; dst = stable counting sort by key
  std::vector<int> key(np);
  srand(1);
  for (int i = 0; i < np; i++) {
    long c = i / ppc;
    if ((double)rand() / RAND_MAX < hot) {
      const int dir = rand() % 6; const long step[6] = {1, -1, n, -n, (long)n * n, -(long)n * n};
      c = (c + step[dir] + ncell) % ncell;
    }
    key[i] = (int)c;
  }
  std::vector<int> count(ncell + 1, 0), dst(np);
  for (int i = 0; i < np; i++) count[key[i] + 1]++;
  for (long c = 0; c < ncell; c++) count[c + 1] += count[c];
  for (int i = 0; i < np; i++) dst[i] = count[key[i]]++;
  int *d_dst; hipMalloc(&d_dst, sizeof(int) * np); hipMemcpy(d_dst, dst.data(), sizeof(int) * np, hipMemcpyHostToDevice);
  float *in, *out; hipMalloc(&in, 32ul * np); hipMalloc(&out, 32ul * np); hipMemset(in, 0, 32ul * np);
  Soa si, so; for (int c = 0; c < 8; c++) { si.a[c] = in + (size_t)c * np; so.a[c] = out + (size_t)c * np; }
  const int nb = (np + 255) / 256;
  const double gb = 1e-9 * np * (32 + 32 + 4);
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(k_soa8, dim3(nb), dim3(256), 0, 0, si, so, d_dst, np); });
  printf("hot=%.2f  soa8  %.3f ms  %.0f GB/s\n", hot, ms, gb / ms * 1e3);
  int *d_src; hipMalloc(&d_src, sizeof(int) * np);
  ms = timeit([&] { hipLaunchKernelGGL(k_inverse, dim3(nb), dim3(256), 0, 0, d_dst, d_src, np); hipLaunchKernelGGL(k_gather8, dim3(nb), dim3(256), 0, 0, si, so, d_src, np); });
  printf("hot=%.2f  soa8 as inverse + gather  %.3f ms  %.0f GB/s\n", hot, ms, gb / ms * 1e3);
  ms = timeit([&] { hipLaunchKernelGGL(k_gather8, dim3(nb), dim3(256), 0, 0, si, so, d_src, np); });
  printf("hot=%.2f  soa8 gather alone          %.3f ms\n", hot, ms);
  ms = timeit([&] { hipLaunchKernelGGL(k_f4x2, dim3(nb), dim3(256), 0, 0, (const float4 *)in, (const float4 *)(in + 4ul * np), (float4 *)out, (float4 *)(out + 4ul * np), d_dst, np); });
  printf("hot=%.2f  f4x2  %.3f ms  %.0f GB/s\n", hot, ms, gb / ms * 1e3);
  ms = timeit([&] { hipLaunchKernelGGL(k_aos32, dim3(nb), dim3(256), 0, 0, (const Rec *)in, (Rec *)out, d_dst, np); });
  printf("hot=%.2f  aos32 %.3f ms  %.0f GB/s\n", hot, ms, gb / ms * 1e3);
  return 0;
}

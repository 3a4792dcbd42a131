// LDS cost of the instruction kinds the push kernel's deposit and crosser queue are made of (gfx950):
// cycles per wave-instruction PER CU with 5 waves per SIMD all issuing the same LDS instruction back to back,
// for different numbers of live lanes and address patterns.  (tools/ubench, round 2)
#include <hip/hip_runtime.h>
#include <cstdio>
enum { ADD_F32, ADD_U32, BPERM, PERM, WR_B32, WR_B128, RD_B32, RD_B128, ADD_RTN_F32, ADD_U64, ADD_RTN_U32, ADD_F64 };
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long live, int stride_words, int same) {
  __shared__ float s[8192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
  __syncthreads();
  // byte address of this lane: wave-private 2048-word region, lane * stride (or one address for all lanes)
  unsigned addr = (unsigned)(size_t)(s) + (wave * 2048 + (same ? 0 : (lane * stride_words) % 2048)) * 4;
  if (OP == WR_B128 || OP == RD_B128) addr = (unsigned)(size_t)(s) + (wave * 2048 + ((lane * 4 * (same ? 0 : 1)) % 2048)) * 4;
  float v = 1.f, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 w = {1.f, 2.f, 3.f, 4.f}, rr = {0, 0, 0, 0};
  const unsigned addr8 = (unsigned)(size_t)(s) + (wave * 2048 + (same ? 0 : (lane * 2 * stride_words) % 2048)) * 4;
  unsigned long long v64 = 3; double d64 = 1.0;
  const bool on = (live >> lane) & 1ull;
  const unsigned bp = (unsigned)((lane * 17) & 63) << 2;
  if (on) {
    for (int i = 0; i < iters; i++) {
#define R8(X) X X X X X X X X
      if (OP == ADD_F32) { asm volatile(R8("ds_add_f32 %0, %1\n") :: "v"(addr), "v"(v) : "memory"); }
      else if (OP == ADD_U32) { asm volatile(R8("ds_add_u32 %0, %1\n") :: "v"(addr), "v"(v) : "memory"); }
      else if (OP == ADD_RTN_F32) { asm volatile(R8("ds_add_rtn_f32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(addr), "v"(v) : "memory"); }
      else if (OP == ADD_U64) { asm volatile(R8("ds_add_u64 %0, %1\n") :: "v"(addr8), "v"(v64) : "memory"); }
      else if (OP == ADD_F64) { asm volatile(R8("ds_add_f64 %0, %1\n") :: "v"(addr8), "v"(d64) : "memory"); }
      else if (OP == ADD_RTN_U32) { asm volatile(R8("ds_add_rtn_u32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(addr), "v"(v) : "memory"); }
      else if (OP == BPERM) { asm volatile(R8("ds_bpermute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(bp), "v"(v) : "memory"); }
      else if (OP == PERM) { asm volatile(R8("ds_permute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(bp), "v"(v) : "memory"); }
      else if (OP == WR_B32) { asm volatile(R8("ds_write_b32 %0, %1\n") :: "v"(addr), "v"(v) : "memory"); }
      else if (OP == WR_B128) { asm volatile(R8("ds_write_b128 %0, %1\n") :: "v"(addr), "v"(w) : "memory"); }
      else if (OP == RD_B32) { asm volatile(R8("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(addr) : "memory"); }
      else if (OP == RD_B128) { asm volatile(R8("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=&v"(rr) : "v"(addr) : "memory"); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + rr.x + s[threadIdx.x];
}
template <int OP> static void run(const char *name, float *d, unsigned long long live, int stride, int same, const char *what) {
  const int iters = 4000, wgs = 1280;   // 5 workgroups of 4 waves per CU
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d, 10, live, stride, same);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d, iters, live, stride, same);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 20 waves x iters x 8 instructions
  printf("%-16s %-34s %7.1f clk per wave-instruction per CU (2.4 GHz)\n", name, what, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 20));
}
int main() {
  float *d; hipMalloc(&d, 256 * 4096 * 4);
  const unsigned long long ALL = ~0ull, ONE = 1ull, THREE = 0x0000100000100001ull, T60 = (1ull << 60) - 1;
  run<ADD_F32>("ds_add_f32", d, ALL, 1, 0, "64 lanes, consecutive words");
  run<ADD_F32>("ds_add_f32", d, ALL, 1, 0, "64 lanes, consecutive words");
  run<ADD_F32>("ds_add_f32", d, T60, 17, 0, "60 lanes, stride 17 words");
  run<ADD_F32>("ds_add_f32", d, THREE, 1, 0, "3 lanes");
  run<ADD_F32>("ds_add_f32", d, ONE, 1, 0, "1 lane");
  run<ADD_F32>("ds_add_f32", d, ALL, 1, 1, "64 lanes, one address");
  run<ADD_U32>("ds_add_u32", d, ALL, 1, 0, "64 lanes, consecutive words");
  run<ADD_U32>("ds_add_u32", d, THREE, 1, 0, "3 lanes");
  run<ADD_U32>("ds_add_u32", d, ALL, 1, 1, "64 lanes, one address");
  run<ADD_U64>("ds_add_u64", d, ALL, 1, 0, "64 lanes, consecutive dwords x2");
  run<ADD_U64>("ds_add_u64", d, THREE, 1, 0, "3 lanes");
  run<ADD_U64>("ds_add_u64", d, 0xffffull, 1, 1, "16 lanes, one address");
  run<ADD_F64>("ds_add_f64", d, ALL, 1, 0, "64 lanes");
  run<ADD_F64>("ds_add_f64", d, THREE, 1, 0, "3 lanes");
  run<ADD_RTN_U32>("ds_add_rtn_u32", d, ALL, 1, 0, "64 lanes (waited per 8)");
  run<ADD_U32>("ds_add_u32", d, 0xffull, 1, 1, "8 lanes, one address");
  run<ADD_U32>("ds_add_u32", d, 0xffffull, 1, 1, "16 lanes, one address");
  run<ADD_F32>("ds_add_f32", d, 0xffffull, 17, 0, "16 lanes, stride 17");
  run<ADD_RTN_F32>("ds_add_rtn_f32", d, ALL, 1, 0, "64 lanes, consecutive (waited)");
  run<BPERM>("ds_bpermute_b32", d, ALL, 1, 0, "64 lanes (waited per 8)");
  run<PERM>("ds_permute_b32", d, ALL, 1, 0, "64 lanes (waited per 8)");
  run<WR_B32>("ds_write_b32", d, ALL, 1, 0, "64 lanes, consecutive words");
  run<WR_B32>("ds_write_b32", d, THREE, 1, 0, "3 lanes");
  run<WR_B128>("ds_write_b128", d, ALL, 1, 0, "64 lanes, consecutive");
  run<WR_B128>("ds_write_b128", d, 0xffull, 1, 0, "8 lanes");
  run<RD_B32>("ds_read_b32", d, ALL, 1, 0, "64 lanes (waited per 8)");
  run<RD_B128>("ds_read_b128", d, ALL, 1, 0, "64 lanes (waited per 8)");
  return 0;
}

// Issue cost of the instruction kinds the push kernel is made of (gfx950), 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float c = 1.0000001f, f = (threadIdx.x & 3) ? 1.f : 0.f;
  int sink = 0;
  for (int i = 0; i < iters; i++) {
#define OPS(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(f) : "vcc");
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_DPP(n) "v_fmac_f32_dpp %" #n ", %" #n ", %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPPB(n) "v_fmac_f32_dpp %" #n ", %" #n ", %9 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_DSC(n) "v_div_scale_f32 %" #n ", vcc, %" #n ", %8, %" #n "\n"
#define I_DFX(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
#define I_DFM(n) "v_div_fmas_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_RDL(n) "v_readlane_b32 s22, %" #n ", 3\n"
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %" #n " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
    if (MODE == 0) { OPS(I_MUL) } else if (MODE == 1) { OPS(I_FMA) } else if (MODE == 2) { OPS(I_FMAC) }
    else if (MODE == 3) { OPS(I_DPP) } else if (MODE == 4) { OPS(I_DPPB) } else if (MODE == 5) { OPS(I_CND) }
    else if (MODE == 6) { asm volatile("s_mov_b64 s[20:21], vcc" ::: "s20", "s21"); OPS(I_CND64) } else if (MODE == 7) { OPS(I_CMP) } else if (MODE == 8) { OPS(I_MOV) }
    else if (MODE == 9) { OPS(I_ADDU) } else if (MODE == 10) { OPS(I_RCP) } else if (MODE == 11) { OPS(I_SQRT) }
    else if (MODE == 12) { OPS(I_DSC) } else if (MODE == 13) { OPS(I_DFX) } else if (MODE == 14) { OPS(I_DFM) }
    else if (MODE == 15) { OPS(I_MULLO) } else if (MODE == 16) { asm volatile("" ::: "s22"); OPS(I_RDL) } else if (MODE == 17) { OPS(I_MOVDPP) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sink;
}
template <int MODE> static void run(const char *name, float *d) {
  const int iters = 100000, wgs = 1280;   // 5 waves per SIMD, like the push kernel
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, 100);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-22s %.2f clk/instr/SIMD (at 2.4 GHz, 5 waves/SIMD)\n", name, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * 5));
}
int main() {
  float *d; hipMalloc(&d, 256 * 4096 * 4);
  run<0>("v_mul_f32", d); run<0>("v_mul_f32", d); run<1>("v_fma_f32", d); run<2>("v_fmac_f32", d); run<3>("v_fmac_f32_dpp row_shr", d); run<4>("v_fmac_f32_dpp bcast", d);
  run<5>("v_cndmask vcc", d); run<6>("v_cndmask_e64", d); run<7>("v_cmp_lt_f32", d); run<8>("v_mov_b32", d); run<9>("v_add_u32", d);
  run<10>("v_rcp_f32", d); run<11>("v_sqrt_f32", d); run<12>("v_div_scale_f32", d); run<13>("v_div_fixup_f32", d); run<14>("v_div_fmas_f32", d);
  run<15>("v_mul_lo_u32", d); run<16>("v_readlane_b32", d); run<17>("v_mov_b32_dpp", d);
  return 0;
}

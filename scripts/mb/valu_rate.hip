// micro-benchmark: issue cost (cycles per wave-instruction) of the integer / transcendental VALU ops the dropout hash and the
// softmax are made of, one wave on one SIMD, 8 independent chains so that latency does not show
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(s) s s s s s s s s
#define OP(name, text)                                                                                                  \
  __global__ __launch_bounds__(64) void k_##name(unsigned* out, long long* cyc, int iters) {                            \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned c = 0x9E3779B1u + threadIdx.x;                                                                            \
    long long t0 = __builtin_readcyclecounter();                                                                       \
    for (int i = 0; i < iters; ++i) {                                                                                  \
      asm volatile(REP8(text) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); \
    }                                                                                                                  \
    long long t1 = __builtin_readcyclecounter();                                                                       \
    out[threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                          \
    if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                                            \
  }

OP(mul_lo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
OP(mul_hi, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n")
OP(mul_u24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n")
OP(mad_u24, "v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7\n")
OP(xor_, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
OP(lshl_add, "v_lshl_add_u32 %0, %0, 5, %8\n v_lshl_add_u32 %1, %1, 5, %8\n v_lshl_add_u32 %2, %2, 5, %8\n v_lshl_add_u32 %3, %3, 5, %8\n v_lshl_add_u32 %4, %4, 5, %8\n v_lshl_add_u32 %5, %5, 5, %8\n v_lshl_add_u32 %6, %6, 5, %8\n v_lshl_add_u32 %7, %7, 5, %8\n")
OP(xad, "v_xad_u32 %0, %0, %8, %0\n v_xad_u32 %1, %1, %8, %1\n v_xad_u32 %2, %2, %8, %2\n v_xad_u32 %3, %3, %8, %3\n v_xad_u32 %4, %4, %8, %4\n v_xad_u32 %5, %5, %8, %5\n v_xad_u32 %6, %6, %8, %6\n v_xad_u32 %7, %7, %8, %7\n")
OP(alignbit, "v_alignbit_b32 %0, %0, %0, 13\n v_alignbit_b32 %1, %1, %1, 13\n v_alignbit_b32 %2, %2, %2, 13\n v_alignbit_b32 %3, %3, %3, 13\n v_alignbit_b32 %4, %4, %4, 13\n v_alignbit_b32 %5, %5, %5, 13\n v_alignbit_b32 %6, %6, %6, 13\n v_alignbit_b32 %7, %7, %7, 13\n")
OP(perm, "v_perm_b32 %0, %0, %8, %0\n v_perm_b32 %1, %1, %8, %1\n v_perm_b32 %2, %2, %8, %2\n v_perm_b32 %3, %3, %8, %3\n v_perm_b32 %4, %4, %8, %4\n v_perm_b32 %5, %5, %8, %5\n v_perm_b32 %6, %6, %8, %6\n v_perm_b32 %7, %7, %8, %7\n")
OP(exp_, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
OP(fma_, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n")
OP(dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
OP(cvt_pk, "v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8\n")
OP(cndmask, "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n")

template <typename K> void run(const char* name, K kern, int per_iter) {
  unsigned* out; long long* cyc; hipMalloc(&out, 256); hipMalloc(&cyc, 16);
  const int iters = 2000;
  kern<<<1, 64>>>(out, cyc, iters); hipDeviceSynchronize();
  kern<<<1, 64>>>(out, cyc, iters); hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s %6.2f cycles per instruction\n", name, (double)h / ((double)iters * per_iter));
  hipFree(out); hipFree(cyc);
}
int main() {
  run("mul_lo_u32", k_mul_lo, 64); run("mul_hi_u32", k_mul_hi, 64); run("mul_u24", k_mul_u24, 64); run("mad_u24", k_mad_u24, 64);
  run("xor", k_xor_, 64); run("lshl_add", k_lshl_add, 64); run("xad", k_xad, 64); run("alignbit", k_alignbit, 64); run("perm", k_perm, 64);
  run("exp", k_exp_, 64); run("fma", k_fma_, 64); run("mov_dpp", k_dpp, 64); run("cvt_pk_bf16", k_cvt_pk, 64); run("cmp+cndmask", k_cndmask, 64);
  return 0;
}

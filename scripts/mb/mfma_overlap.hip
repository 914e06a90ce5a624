// micro-benchmark (inline asm, nothing for the compiler to move): how many cycles does one wave need per
// v_mfma_f32_32x32x2_f32 when K independent instructions of one kind are issued between consecutive MFMAs?
#include <hip/hip_runtime.h>
#include <cstdio>
#define STR2(x) #x
#define STR(x) STR2(x)
#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP12(x) REP8(x) REP4(x)
#define MF0 "v_mfma_f32_32x32x2_f32 a[0:15], v20, v21, a[0:15]\n"
#define MF1 "v_mfma_f32_32x32x2_f32 a[16:31], v20, v22, a[16:31]\n"
#define F_VALU "v_mul_f32 v30, v31, v32\n"
#define F_LDS "ds_read_b32 v33, v40\n"
#define F_ACC "v_accvgpr_read_b32 v34, a40\n"
#define F_NOP "s_nop 0\n"
#define F_SALU "s_add_u32 s20, s20, 1\n"
#define F_WB "v_mul_f32 v21, v31, v32\n" /* rewrites the B operand of MF0 */
#define BODY(F) MF0 F MF1 F MF0 F MF1 F MF0 F MF1 F MF0 F MF1 F
#define KERNEL(NAME, F)                                                                                              \
  __global__ __launch_bounds__(64) void NAME(float* out, long long* cyc, int iters) {                                \
    __shared__ float lds[1024];                                                                                      \
    lds[threadIdx.x] = threadIdx.x;                                                                                  \
    __syncthreads();                                                                                                 \
    unsigned addr = (unsigned)(size_t)(lds + threadIdx.x);                                                           \
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 0.5\n v_mov_b32 v31, 1.0\n v_mov_b32 v32, 1.0\n" ::"v"(addr) : "v40", "v20", "v21", "v22", "v31", "v32"); \
    long long t0 = __builtin_readcyclecounter();                                                                     \
    for (int i = 0; i < iters; ++i)                                                                                  \
      asm volatile(BODY(F) "s_waitcnt lgkmcnt(0)\n" ::: "v30", "v33", "v34", "v21", "s20", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a40", "memory"); \
    long long t1 = __builtin_readcyclecounter();                                                                     \
    out[threadIdx.x] = 0.f;                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                       \
  }
KERNEL(k_none, "")
KERNEL(k_valu1, REP1(F_VALU)) KERNEL(k_valu2, REP2(F_VALU)) KERNEL(k_valu4, REP4(F_VALU)) KERNEL(k_valu8, REP8(F_VALU)) KERNEL(k_valu12, REP12(F_VALU))
KERNEL(k_lds1, REP1(F_LDS)) KERNEL(k_lds2, REP2(F_LDS)) KERNEL(k_lds4, REP4(F_LDS))
KERNEL(k_acc1, REP1(F_ACC)) KERNEL(k_acc2, REP2(F_ACC)) KERNEL(k_acc4, REP4(F_ACC))
KERNEL(k_nop4, REP4(F_NOP)) KERNEL(k_salu4, REP4(F_SALU)) KERNEL(k_salu12, REP12(F_SALU))
KERNEL(k_wb, F_WB)
KERNEL(k_mix, F_VALU F_LDS F_LDS)
#define RUN(NAME) do { NAME<<<1024, 64>>>(out, cyc, 500); (void)hipDeviceSynchronize(); NAME<<<1024, 64>>>(out, cyc, 500); (void)hipDeviceSynchronize(); \
  long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-10s %.1f cycles per mfma\n", #NAME, h / (500.0 * 8)); } while (0)
int main() {
  float* out; long long* cyc; (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 16);
  RUN(k_none); RUN(k_valu1); RUN(k_valu2); RUN(k_valu4); RUN(k_valu8); RUN(k_valu12);
  RUN(k_lds1); RUN(k_lds2); RUN(k_lds4); RUN(k_acc1); RUN(k_acc2); RUN(k_acc4); RUN(k_nop4); RUN(k_salu4); RUN(k_salu12); RUN(k_wb); RUN(k_mix);
  return 0;
}

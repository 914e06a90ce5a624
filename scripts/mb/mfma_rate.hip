// micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for 1, 2, 4 independent chains in one wave, 1 wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  long long w1 = wall_clock64();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int CH> void run(int blocks) {
  float* out; long long* cyc; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, 16);
  int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<blocks, 64>>>(out, cyc, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<CH><<<blocks, 64>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  double n = (double)iters * 8 * CH;
  printf("chains=%d blocks=%d: %.1f memtime ticks/mfma, %.2f wallclock(100MHz) ticks/mfma, %.3f ms total -> %.1f ns/mfma\n", CH, blocks, h[0] / n, h[1] / n, ms, ms * 1e6 / n);
}
int main() {
  run<1>(1); run<2>(1); run<4>(1);
  run<1>(1024); run<2>(1024); run<4>(1024);
  run<2>(2048);
  return 0;
}

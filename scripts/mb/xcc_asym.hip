// Is memory slower from some XCCs?  256 workgroups of 256 threads (one per CU), each streams the SAME amount of data three ways and stamps
// s_memtime around each phase and its XCC id:
//   hbm   a private 4 MB region of a 1 GB buffer, read once (16-byte loads, 8 in flight per lane)
//   l2    a private 64 KB region, read 64 times (L2-resident after the first pass)
//   shared a 2 MB region that EVERY workgroup reads (an operand all tiles share), 2 passes
// Output: per XCC the median cycles of each phase.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const uint4* big, const uint4* shared, long long* out, float* sink) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float acc = 0.f;
  auto sum = [&](const uint4 v) { acc += __uint_as_float(v.x & 0x3F800000u) + __uint_as_float(v.w & 0x3F800000u); };
  long long t0 = __builtin_readcyclecounter();
  {  // hbm: 4 MB = 262144 uint4 per workgroup, 1024 per thread
    const uint4* p = big + (size_t)wg * 262144 + tid;
    for (int i = 0; i < 1024; i += 8) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(i + j) * 256];
#pragma unroll
      for (int j = 0; j < 8; ++j) sum(v[j]);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  {  // l2: 64 KB = 4096 uint4 per workgroup, 16 per thread, 64 passes
    const uint4* p = big + (size_t)(256 + wg) * 262144 + tid;
    for (int pass = 0; pass < 64; ++pass) {
      uint4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = p[(size_t)j * 256];
#pragma unroll
      for (int j = 0; j < 16; ++j) sum(v[j]);
      asm volatile("" ::: "memory");
    }
  }
  long long t2 = __builtin_readcyclecounter();
  {  // shared: 2 MB = 131072 uint4, 512 per thread, 2 passes
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < 512; i += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = shared[(size_t)(i + j) * 256 + tid];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum(v[j]);
      }
  }
  long long t3 = __builtin_readcyclecounter();
  if (tid == 0) {
    out[wg * 4 + 0] = xcc;
    out[wg * 4 + 1] = t1 - t0;
    out[wg * 4 + 2] = t2 - t1;
    out[wg * 4 + 3] = t3 - t2;
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  uint4 *big, *shared;
  long long* out;
  float* sink;
  const size_t nbig = (size_t)512 * 262144;  // 2 GB
  hipMalloc(&big, nbig * 16);
  hipMalloc(&shared, 131072 * 16);
  hipMalloc(&out, 256 * 4 * 8);
  hipMalloc(&sink, 4);
  hipMemset(big, 0x11, nbig * 16);
  hipMemset(shared, 0x22, 131072 * 16);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, big, shared, out, sink);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    printf("rep %d  (median cycles per workgroup: hbm 4 MB | l2 64 KB x 64 | shared 2 MB x 2)\n", rep);
    for (int x = 0; x < 8; ++x) {
      std::vector<long long> a, b, c;
      for (int w = 0; w < 256; ++w)
        if (h[w * 4] == x) { a.push_back(h[w * 4 + 1]); b.push_back(h[w * 4 + 2]); c.push_back(h[w * 4 + 3]); }
      if (a.empty()) continue;
      std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
      printf("  XCC %d (%zu workgroups): %8lld | %8lld | %8lld\n", x, a.size(), a[a.size() / 2], b[b.size() / 2], c[c.size() / 2]);
    }
  }
  return 0;
}

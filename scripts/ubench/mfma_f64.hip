// micro-benchmark: cost of v_mfma_f64_16x16x4_f64 on one wavefront - a dependent chain through the accumulator versus 2 and 4
// independent chains (cycles per instruction, s_memtime ticks), with 1, 2 and 4 waves per SIMD competing.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long tick(d4& dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
  return (long long)t;
}
__global__ void k(double* out, long long* cyc, double x0) {
  const double a = x0 + threadIdx.x * 1e-9, b = 1.0000001 + threadIdx.x * 1e-9;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = tick(c0);
#pragma unroll
  for (int i = 0; i < 64; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  long long t1 = tick(c0);
#pragma unroll
  for (int i = 0; i < 32; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0); }
  c0 += c1;
  long long t2 = tick(c0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
  }
  c0 += c1 + c2 + c3;
  long long t3 = tick(c0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 65536); (void)hipMalloc(&cyc, 64);
  for (int threads : {64, 256, 512, 1024}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0);
    long long h[3];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%4d threads (%d waves/SIMD): 64 MFMA  dependent %lld ticks (%.1f each), 2 chains %lld (%.1f), 4 chains %lld (%.1f)\n", threads, (threads + 255) / 256,
           h[0], h[0] / 64.0, h[1], h[1] / 64.0, h[2], h[2] / 64.0);
  }
  return 0;
}

// micro-benchmarks behind HISTORY.md's (rounds 1-3) latency model of the diagonal-tile factorisation:
// dependent-chain and issue cost of fp64 FMA / rcp / rsq on one wavefront, accuracy of the raw seeds.
#include <hip/hip_runtime.h>
#define HIPIGN(x) (void)(x)
__device__ __forceinline__ long long tick(double& dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
  return (long long)t;
}
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k_lat(double* out, long long* cyc, double x0) {
  double x = x0 + threadIdx.x * 1e-9, y = 1.0000001;
  long long t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = fma(x, y, 1e-9);
  long long t1 = tick(x);
  double a = x, b = x + 1, c = x + 2, d = x + 3;
#pragma unroll
  for (int i = 0; i < 64; ++i) { a = fma(a, y, 1e-9); b = fma(b, y, 1e-9); c = fma(c, y, 1e-9); d = fma(d, y, 1e-9); }
  double r = a + b + c + d;
  long long t2 = tick(r);
#pragma unroll
  for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rsq(r + 1.5);
  long long t3 = tick(r);
#pragma unroll
  for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rcp(r + 1.5);
  long long t4 = tick(r);
  double e = r, f = r + 1, g = r + 2, h = r + 3;
#pragma unroll
  for (int i = 0; i < 16; ++i) { e = __builtin_amdgcn_rsq(e + 1.5); f = __builtin_amdgcn_rsq(f + 1.5); g = __builtin_amdgcn_rsq(g + 1.5); h = __builtin_amdgcn_rsq(h + 1.5); }
  double sum = e + f + g + h;
  long long t5 = tick(sum);
  out[threadIdx.x] = sum;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
__global__ void k_acc(const double* in, double* rs, double* rc, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { rs[i] = __builtin_amdgcn_rsq(in[i]); rc[i] = __builtin_amdgcn_rcp(in[i]); }
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
  long long h[5]; (void)hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
  printf("dependent fma f64: %.1f cyc/op; 4 independent chains: %.1f cyc/op; dependent rsq(+add): %.1f; dependent rcp(+add): %.1f; 4 indep rsq(+add): %.1f per pair\n",
         h[0] / 256.0, h[1] / 256.0, h[2] / 64.0, h[3] / 64.0, h[4] / 64.0);
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n);
  for (int i = 0; i < n; ++i) x[i] = std::exp(-30.0 + 60.0 * (i + 0.5) / n) * (1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0);
  double *dx, *da, *db; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_acc, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
  (void)hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) { e1 = std::fmax(e1, std::fabs(a[i] * std::sqrt(x[i]) - 1.0)); e2 = std::fmax(e2, std::fabs(b[i] * x[i] - 1.0)); }
  printf("max rel err raw v_rsq_f64: %.3e (2^%.1f)   raw v_rcp_f64: %.3e (2^%.1f)\n", e1, std::log2(e1), e2, std::log2(e2));
  return 0;
}

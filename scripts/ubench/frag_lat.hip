// micro-benchmark: latency of fetching one 32x32 fp64 tile operand per workgroup, dependent (one after the other), in the two
// forms the factorisation uses: ct_gld (2 x 16 B per lane, fully coalesced, whole tile per workgroup) and ct_gfrag (8 x 8 B
// per lane in the MFMA operand layout, half a tile per wave). Cold (tiles last written by another kernel) and warm (L2).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long tick(double& dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
  return (long long)t;
}
__global__ void k_fill(double* A, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) A[i] = 1e-3 * (i & 1023);
}
__global__ __launch_bounds__(256) void k_lat(const double* __restrict__ A, int ntile, int mode, int iters, double* out, long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, blk = w & 1;
  double s = 0.0;
  int tile = (blockIdx.x * 7919) % ntile;
  long long t0 = tick(s);
  for (int it = 0; it < iters; ++it) {
    const double* G = A + (size_t)tile * 1024;
    double v = 0.0;
    if (mode == 0) {
      const double2* g2 = reinterpret_cast<const double2*>(G);
      const double2 a = g2[tid], b = g2[tid + 256];
      v = a.x + a.y + b.x + b.y;
    } else {
      const double* p = G + 16 * blk + (lane & 15) + 32 * (lane >> 4);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) v += p[32 * 4 * kk];
    }
    s += v;
    tile = (tile + 1 + ((int)(v * 1e-9) & 1)) % ntile;   // dependent chain
  }
  long long t1 = tick(s);
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int ntile = 8192; const size_t n = (size_t)ntile * 1024;
  double *A, *out; long long* cyc;
  (void)hipMalloc(&A, n * 8); (void)hipMalloc(&out, 8 * 256 * 2048); (void)hipMalloc(&cyc, 8 * 2048);
  for (int grid : {1, 256, 1024}) for (int mode : {0, 1}) {
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, A, n);
    const int iters = 16;
    long long h[2048];
    double res[2];
    for (int pass = 0; pass < 2; ++pass) {
      hipLaunchKernelGGL(k_lat, dim3(grid), dim3(256), 0, 0, A, ntile, mode, iters, out, cyc);
      (void)hipMemcpy(h, cyc, 8 * grid, hipMemcpyDeviceToHost);
      double m = 0; for (int i = 0; i < grid; ++i) m += h[i];
      res[pass] = m / grid / iters;
    }
    printf("grid %4d  %s: cold %.0f ticks per dependent fetch, second pass %.0f\n", grid, mode ? "ct_gfrag (8 x 8 B, operand layout)" : "ct_gld   (2 x 16 B, coalesced)    ", res[0], res[1]);
  }
  return 0;
}

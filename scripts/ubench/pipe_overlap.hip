// What one wavefront can overlap on gfx950 (s_memtime ticks, one wave on one SIMD):
//   A  48 dependent v_fma_f64                                   B  8 independent v_mfma_f64_16x16x4
//   C  the 48 fmas with the 8 MFMAs spread between them          (A + B if the fp64 MFMA and the fp64 VALU share a pipe, max(A, B) if not)
//   D  ds_write_b64 -> s_waitcnt -> 6 ds_read (b128 / b64) -> s_waitcnt: the LDS broadcast of a pivot block
//   E  v_rcp_f64 + third-order step, four in a dependent chain
//   F  C with v_fma_f32 instead of fp64 (does ANY VALU work overlap an fp64 MFMA?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define TICK() ((long long)__builtin_readcyclecounter())
__global__ void k(double* out, long long* cyc, double x0) {
  __shared__ __attribute__((aligned(16))) double pan[128];
  const int lane = threadIdx.x;
  double a = x0 + lane * 1e-9, b = 1.0000001 + lane * 1e-9, v = a;
  const d4 zero = {0, 0, 0, 0};
  d4 c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = zero;
  long long t[8];
  asm volatile("" : "+v"(v));
  t[0] = TICK();
#pragma unroll
  for (int i = 0; i < 48; ++i) v = fma(v, b, a);
  asm volatile("" : "+v"(v));
  t[1] = TICK();
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(c[i]));
  t[2] = TICK();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) v = fma(v, b, a);
  }
  asm volatile("" : "+v"(v));
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(c[i]));
  t[3] = TICK();
  double s = 0.0;
#pragma unroll
  for (int rep = 0; rep < 8; ++rep) {
    double* pb = pan + 64 * (rep & 1);
    pb[lane] = v + s;
    const double2 r0 = *reinterpret_cast<const double2*>(pb + 4), r1 = *reinterpret_cast<const double2*>(pb + 6);
    const double q = pb[21];
    const double2 r2 = *reinterpret_cast<const double2*>(pb + 22), r3 = *reinterpret_cast<const double2*>(pb + 38);
    const double q2 = pb[55];
    s = r0.x + r0.y + r1.x + r1.y + q + r2.x + r2.y + r3.x + r3.y + q2;
  }
  asm volatile("" : "+v"(s));
  t[4] = TICK();
  double r = a;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double y = __builtin_amdgcn_rcp(r);
    const double e = fma(-r, y, 1.0);
    r = fma(y, fma(e, e, e), y) + 1.5;
  }
  asm volatile("" : "+v"(r));
  t[5] = TICK();
  float vf = (float)a, bf = (float)b, af = 0.25f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) vf = fmaf(vf, bf, af);
  }
  asm volatile("" : "+v"(vf));
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(c[i]));
  t[6] = TICK();
  double acc = v + s + r + vf;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += c[i][0] + c[i][3];
  out[lane] = acc;
  if (lane == 0)
    for (int i = 0; i < 6; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&cyc, 64);
  long long h[6];
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("A 48 dependent v_fma_f64: %lld ticks (%.1f each)\n", h[0], h[0] / 48.0);
  printf("B 8 independent v_mfma_f64_16x16x4: %lld ticks (%.1f each)\n", h[1], h[1] / 8.0);
  printf("C 48 v_fma_f64 with the 8 MFMAs between them: %lld ticks (A + B = %lld, max = %lld)\n", h[2], h[0] + h[1], h[0] > h[1] ? h[0] : h[1]);
  printf("D LDS broadcast (write b64, 6 reads, sum): %.1f ticks per round\n", h[3] / 8.0);
  printf("E v_rcp_f64 + third-order step + add, dependent: %.1f ticks each\n", h[4] / 4.0);
  printf("F 48 v_fma_f32 with the 8 fp64 MFMAs between them: %lld ticks\n", h[5]);
  return 0;
}

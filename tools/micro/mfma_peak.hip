// What the f32 matrix pipe sustains with NO memory work: W wavefronts per SIMD issuing independent
// v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 back to back on every CU for ~1 ms.  The figure bench.py prices
// against (157.3 TF/s = 256 CUs x 256 flop/clk x 2.4 GHz) assumes the boost clock; this prints what the part holds.
// build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BIG>
__global__ void __launch_bounds__(256) k_mfma(int iters, float* out, unsigned long long* clk) {
  const float a = (float)threadIdx.x * 1e-9f, b = 1e-9f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (BIG) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    r = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0}, c6 = {0}, c7 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
    }
    r = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (r == 12345.f) out[0] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int BIG>
static void run(int blocks, int iters, float* out, unsigned long long* clk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<BIG>, dim3(blocks), dim3(256), 0, 0, iters, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mfma<BIG>, dim3(blocks), dim3(256), 0, 0, iters, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double per = BIG ? 4.0 * 32 * 32 * 2 * 2 : 8.0 * 16 * 16 * 4 * 2;  // flop per wavefront per iteration
  const double fl = (double)blocks * 4 * iters * per;
  printf("%s blocks=%d (%.1f waves/SIMD) iters=%d: %.3f ms  %.1f TF/s  (%.3f of 157.3)  s_memtime ticks %llu (100 MHz: %.3f ms)\n",
         BIG ? "32x32x2 " : "16x16x4 ", blocks, blocks / 256.0, iters, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, c, c / 1e5);
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 64);
  hipMalloc(&clk, 64);
  for (int blocks : {256, 512, 768}) {
    run<1>(blocks, 10000 * 256 / blocks, out, clk);
    run<0>(blocks, 10000 * 256 / blocks, out, clk);
  }
  run<1>(512, 40000, out, clk);  // ~8 ms: has the clock settled?
  return 0;
}

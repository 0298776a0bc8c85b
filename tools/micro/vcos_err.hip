// max abs error of v_cos_f32 (input in revolutions) behind a three-term Cody-Waite reduction modulo 2*pi, against
// float64 cos of the float32 argument; compared with the polynomial path of tg_common.h (cos_cw).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../www2023tiger_amd/csrc -I../../include vcos_err.hip -o vcos_err
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "tg_common.h"
using namespace tg;
__device__ __forceinline__ float cos_hw(float x) {
  const float n = rintf(__fmul_rn(x, 0.15915494309189535f));
  float r = fmaf(-n, 6.2831854820251465f, x);          // 2*pi split in three floats (Cody-Waite)
  r = fmaf(-n, -1.7484555314695172e-07f, r);
  r = fmaf(-n, -7.1054273576010019e-15f, r);
  return __builtin_amdgcn_cosf(__fmul_rn(r, 0.15915494309189535f));
}
__global__ void k(const float* x, float* a, float* b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = cos_hw(x[i]); b[i] = cos_cw(x[i]); }
}
int main() {
  const int n = 1 << 24;
  std::vector<float> x(n), a(n), b(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    const double mag = i % 4 == 0 ? 10.0 : i % 4 == 1 ? 1.0e3 : i % 4 == 2 ? 1.0e5 : 3.0e6;
    x[i] = (float)((2.0 * u - 1.0) * mag);
  }
  float *dx, *da, *db;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
  hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  double ea[4] = {0, 0, 0, 0}, eb[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double ref = cos((double)x[i]);
    ea[i % 4] = fmax(ea[i % 4], fabs(a[i] - ref));
    eb[i % 4] = fmax(eb[i % 4], fabs(b[i] - ref));
  }
  for (int j = 0; j < 4; ++j) printf("range %d: v_cos path max abs err %.3e   polynomial path %.3e\n", j, ea[j], eb[j]);
  return 0;
}

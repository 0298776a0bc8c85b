// Per-CU load bandwidth from L2 for the access patterns of the GEMM staging loads.
// build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/micro/ldbw.hip -o /tmp/ldbw && /tmp/ldbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// pattern 0: a wave reads 1 KB contiguous per instruction (thread t: 16 B at t*16)
// pattern 1: 8 threads per row, 128 B per row, rows `ld` floats apart (GEMM A/W tile staging, BK = 32)
// pattern 2: 16 threads per row, 256 B per row (BK = 64)
// pattern 3: 32 threads per row, 512 B per row (BK = 128)
template <int PAT>
__global__ void __launch_bounds__(256) k_ld(const float4* __restrict__ p, size_t n4, int ld4, int iters, float* out,
                                            int rows_region) {
  const int t = threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  const int tpr = PAT == 0 ? 256 : (8 << (PAT - 1));  // threads per row
  const int r = t / tpr, c = t % tpr;
  const int rows_per_pass = 256 / tpr;
  // each block walks its own row band (like a 64-row M tile), K advancing by tpr*4 floats per iteration
  size_t row0 = ((size_t)blockIdx.x * 64) % rows_region;
#pragma unroll 4
  for (int it = 0; it < iters; ++it) {
    size_t idx;
    if (PAT == 0) {
      idx = ((size_t)blockIdx.x * 4096 + (size_t)it * 256 + t) % n4;
    } else {
      const int kt = it % (ld4 / tpr);                 // k tile
      const int pass = (it / (ld4 / tpr)) % (64 / rows_per_pass);
      idx = (row0 + pass * rows_per_pass + r) * (size_t)ld4 + (size_t)kt * tpr + c;
    }
    const float4 v = p[idx];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int PAT>
void run(const float4* d, size_t n4, int ld4, float* out, int blocks, int rows_region) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_ld<PAT>, dim3(blocks), dim3(256), 0, 0, d, n4, ld4, iters, out, rows_region);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_ld<PAT>, dim3(blocks), dim3(256), 0, 0, d, n4, ld4, iters, out, rows_region);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 256 * 16 * iters;
  printf("pattern %d blocks %4d: %.3f ms  %.2f TB/s  %.1f B/clk/CU (2.4 GHz, %d CUs busy)\n", PAT, blocks, ms,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 2.4e9 / (blocks < 256 ? blocks : 256), blocks < 256 ? blocks : 256);
}

int main() {
  const int ld = 1216, ld4 = ld / 4;       // row length in floats (multiple of 128 floats for pattern 3: 1216 = 9.5 * 128 -> use 1280)
  const int LD = 1280, LD4 = LD / 4;
  const int rows = 3072;
  const size_t n4 = (size_t)rows * LD4;    // 15.7 MB: L2 + MALL resident
  float4* d; float* out;
  hipMalloc(&d, n4 * 16); hipMalloc(&out, 4);
  hipMemset(d, 0, n4 * 16);
  (void)ld; (void)ld4;
  for (int blocks : {144, 256, 512, 1024}) {
    run<0>(d, n4, LD4, out, blocks, rows);
    run<1>(d, n4, LD4, out, blocks, rows);
    run<2>(d, n4, LD4, out, blocks, rows);
    run<3>(d, n4, LD4, out, blocks, rows);
  }
  return 0;
}

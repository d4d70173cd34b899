// Microbenchmark: effective VALU issue rate (clock) of the chip under an all-SIMD FMA load, for short (30 us) and long kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f, g = a + 4.f, h = a + 5.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
      g = __builtin_fmaf(g, b, c); h = __builtin_fmaf(h, b, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd : {1, 2, 4}) for (int iters : {100, 400, 4000, 40000}) {
    const int grid = 256 * waves_per_simd;  // 256 threads = 4 waves per WG -> one per SIMD of a CU
    k_fma<<<grid, 256>>>(out, iters); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
      hipEventRecord(e0); k_fma<<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double insts = (double) iters * 96 * waves_per_simd;  // VALU instructions per SIMD
    printf("waves/simd %d iters %6d: %8.2f us -> %.2f GHz equivalent (4 clk per wave64 VALU op)\n", waves_per_simd, iters, best * 1e3, insts * 4 / (best * 1e-3) / 1e9);
  }
  return 0;
}

// Can a PCIe copy kernel run NEXT TO a kernel that fills every SIMD's register file (k_back: 123 VGPRs, 4 waves per SIMD)?
// Without help its workgroups only find slots when the big kernel drains.  Tried here: stream priority, and CU masks
// (hipExtStreamCreateWithCUMask) that keep a few CUs free of the big kernel and confine the copy kernel to them.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/cu_mask_overlap.hip -o build/cu_mask_overlap && build/cu_mask_overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}
// ~25 us of dependent FMAs per wave generation, register file full (the inline asm pins the VGPR count above 120)
__global__ __launch_bounds__(256) void k_busy(float* out, int iters) {
  float a = threadIdx.x, b = 1.0001f;
  asm volatile("v_mov_b32 v" TOPV ", 0" ::: "v" TOPV);
  for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
  if (a == 123.456f) out[0] = a + b;
}
int main() {
  const size_t nbytes = 640 * 480 * 7 / 16 * 16;
  char *h, *d;
  hipHostMalloc((void**) &h, nbytes, hipHostMallocDefault); memset(h, 1, nbytes);
  hipMalloc((void**) &d, nbytes);
  float* out; hipMalloc((void**) &out, 4);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("%d CUs\n", ncu);
  auto make_masked = [&](int first_free, int n_free, bool copy_side) {
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int cu = 0; cu < ncu; cu++) {
      const bool reserved = cu >= first_free && cu < first_free + n_free;
      if (reserved == copy_side) mask[cu / 32] |= 1u << (cu % 32);
    }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t) mask.size(), mask.data());
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); s = nullptr; }
    return s;
  };
  hipStream_t plain_a, plain_b, hi;
  hipStreamCreateWithFlags(&plain_a, hipStreamNonBlocking); hipStreamCreateWithFlags(&plain_b, hipStreamNonBlocking);
  int lo_p, hi_p; hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
  hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, hi_p);
  const int busy_grid = 4096, busy_iters = 6000, reps = 50;
  auto measure = [&](const char* name, hipStream_t sb, hipStream_t sc, bool busy, bool copy) {
    for (int w = 0; w < 3; w++) {
      if (busy) hipLaunchKernelGGL(k_busy, dim3(busy_grid), dim3(256), 0, sb, out, busy_iters);
      if (copy) hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, sc, (const uint4*) h, (uint4*) d, nbytes / 16);
    }
    hipDeviceSynchronize();
    const double t0 = now();
    for (int r = 0; r < reps; r++) {
      if (busy) hipLaunchKernelGGL(k_busy, dim3(busy_grid), dim3(256), 0, sb, out, busy_iters);
      if (copy) hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, sc, (const uint4*) h, (uint4*) d, nbytes / 16);
    }
    hipDeviceSynchronize();
    printf("%-72s %7.1f us per round\n", name, (now() - t0) / reps);
  };
  measure("busy kernel alone", plain_a, plain_b, true, false);
  measure("copy kernel alone (2.15 MB over PCIe)", plain_a, plain_b, false, true);
  measure("both, two plain streams", plain_a, plain_b, true, true);
  measure("both, copy on a highest-priority stream", plain_a, hi, true, true);
  for (int n_free : {4}) {
    hipStream_t sb = make_masked(0, n_free, false), sc = make_masked(0, n_free, true);
    if (!sb || !sc) continue;
    char name[128];
    snprintf(name, sizeof name, "busy kernel alone on a stream without CUs [0, %d)", n_free);
    measure(name, sb, sc, true, false);
    snprintf(name, sizeof name, "copy kernel alone confined to CUs [0, %d)", n_free);
    measure(name, sb, sc, false, true);
    snprintf(name, sizeof name, "both: busy without CUs [0, %d), copy confined to them", n_free);
    measure(name, sb, sc, true, true);
    snprintf(name, sizeof name, "both: busy without CUs [0, %d), copy on a plain stream", n_free);
    measure(name, sb, plain_b, true, true);
  }
  return 0;
}

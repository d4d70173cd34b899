// How should a one-shot 128 MB result be brought to the host?  (extractMesh: soup + V / C / F)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 128u << 20;
  void* d; hipMalloc(&d, n); hipMemset(d, 1, n); hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); void* h; hipHostMalloc(&h, n, hipHostMallocDefault); double t1 = now();
    hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t2 = now();
    hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t3 = now();
    hipHostFree(h); double t4 = now();
    printf("pinned:   alloc %.2f ms, first copy %.2f, second copy %.2f, free %.2f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    t0 = now(); char* m = (char*) malloc(n); t1 = now();
    hipMemcpy(m, d, n, hipMemcpyDeviceToHost); t2 = now();
    hipMemcpy(m, d, n, hipMemcpyDeviceToHost); t3 = now();
    free(m); t4 = now();
    printf("pageable: alloc %.2f ms, first copy (untouched pages) %.2f, second copy %.2f, free %.2f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    t0 = now(); m = (char*) malloc(n); memset(m, 0, n); t1 = now();
    hipMemcpy(m, d, n, hipMemcpyDeviceToHost); t2 = now();
    printf("pageable zero-filled first (std::vector::resize): fill %.2f ms, copy %.2f\n", t1 - t0, t2 - t1);
    t0 = now(); hipHostRegister(m, n, hipHostRegisterDefault); t1 = now();
    hipMemcpy(m, d, n, hipMemcpyDeviceToHost); t2 = now();
    hipHostUnregister(m); t3 = now();
    printf("register: %.2f ms, copy %.2f, unregister %.2f\n", t1 - t0, t2 - t1, t3 - t2);
    free(m);
    void* d2; t0 = now(); hipMalloc(&d2, n); t1 = now(); hipFree(d2); t2 = now();
    printf("device:   hipMalloc %.2f ms, hipFree %.2f\n", t1 - t0, t2 - t1);
  }
  return 0;
}

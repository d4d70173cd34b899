// What does it cost to get 34 MB of pinned host memory for a first extraction's V / C / F?  (mrh_capi.hip HostVec)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pac tools/micro/pinned_alloc_cost.hip && /tmp/pac
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t sizes[3] = {14u << 20, 14u << 20, 8u << 20};
  (void) hipFree(nullptr);
  void* warm; (void) hipHostMalloc(&warm, 4096, hipHostMallocDefault);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    void* p[3];
    for (int i = 0; i < 3; i++) (void) hipHostMalloc(&p[i], sizes[i], hipHostMallocDefault);
    printf("three hipHostMalloc (14 + 14 + 8 MB): %.2f ms\n", now() - t0);
    for (int i = 0; i < 3; i++) (void) hipHostFree(p[i]);
    t0 = now();
    void* q; (void) hipHostMalloc(&q, 36u << 20, hipHostMallocDefault);
    printf("one hipHostMalloc (36 MB): %.2f ms\n", now() - t0);
    (void) hipHostFree(q);
    t0 = now();
    (void) hipHostMalloc(&q, 36u << 20, hipHostMallocNonCoherent);
    printf("one hipHostMalloc non-coherent (36 MB): %.2f ms\n", now() - t0);
    (void) hipHostFree(q);
    t0 = now();
    const size_t span = (36u << 20) + (2u << 20);
    char* m = (char*) mmap(nullptr, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    char* a = (char*) (((uintptr_t) m + (2u << 20) - 1) & ~(uintptr_t) ((2u << 20) - 1));
    madvise(a, 36u << 20, MADV_HUGEPAGE);
    const double t1 = now();
    for (size_t o = 0; o < (36u << 20); o += 4096) a[o] = 1;
    const double t2 = now();
    hipError_t e = hipHostRegister(a, 36u << 20, hipHostRegisterDefault);
    const double t3 = now();
    printf("mmap + MADV_HUGEPAGE %.2f ms, first touch %.2f ms, hipHostRegister %.2f ms (%s): %.2f ms in all\n", t1 - t0, t2 - t1, t3 - t2, hipGetErrorString(e), t3 - t0);
    (void) hipHostUnregister(a);
    munmap(m, span);
  }
  return 0;
}

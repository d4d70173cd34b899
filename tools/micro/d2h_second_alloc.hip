// Does a device-to-host copy into the SECOND host buffer of a process run as fast as into the first?  (mrh_capi.hip HostVec)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h2 tools/micro/d2h_second_alloc.hip && /tmp/d2h2
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
enum Kind { PINNED, PINNED_NC, MMAP_REG, MMAP_THP, MALLOC };
static const char* names[] = {"hipHostMalloc default", "hipHostMalloc non-coherent", "mmap+THP+hipHostRegister", "mmap+THP pageable", "malloc pageable"};
struct Buf { void* p; void* base; size_t span; Kind k; };
static Buf get(Kind k, size_t bytes) {
  Buf b{nullptr, nullptr, 0, k};
  if (k == PINNED) hipHostMalloc(&b.p, bytes, hipHostMallocDefault);
  else if (k == PINNED_NC) hipHostMalloc(&b.p, bytes, hipHostMallocNonCoherent);
  else if (k == MALLOC) { b.p = malloc(bytes); memset(b.p, 1, bytes); }
  else {
    b.span = bytes + (2u << 20);
    b.base = mmap(nullptr, b.span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    b.p = (void*) (((uintptr_t) b.base + (2u << 20) - 1) & ~(uintptr_t) ((2u << 20) - 1));
    madvise(b.p, bytes, MADV_HUGEPAGE);
    memset(b.p, 1, bytes);
    if (k == MMAP_REG) hipHostRegister(b.p, bytes, hipHostRegisterDefault);
  }
  return b;
}
static void put(Buf& b, size_t bytes) {
  if (b.k == PINNED || b.k == PINNED_NC) hipHostFree(b.p);
  else if (b.k == MALLOC) free(b.p);
  else { if (b.k == MMAP_REG) hipHostUnregister(b.p); munmap(b.base, b.span); }
}
static double copy_ms(void* h, void* d, size_t bytes, hipStream_t s) {
  double best = 1e9;
  for (int i = 0; i < 6; i++) {
    const double t0 = now();
    hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    const double dt = now() - t0;
    if (i && dt < best) best = dt;
  }
  return best;
}
int main(int argc, char** argv) {
  const size_t bytes = 12u << 20;  // one of V / C
  void* d;
  hipMalloc(&d, bytes);
  hipMemset(d, 3, bytes);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  for (int k = 0; k < 5; k++) {
    if (only >= 0 && k != only) continue;
    Buf a = get((Kind) k, bytes);
    const double ta = copy_ms(a.p, d, bytes, s);
    put(a, bytes);
    Buf b = get((Kind) k, bytes);
    const double tb = copy_ms(b.p, d, bytes, s);
    Buf c = get((Kind) k, bytes + (1u << 20));
    const double tc = copy_ms(c.p, d, bytes, s);
    const double tb2 = copy_ms(b.p, d, bytes, s);
    put(b, bytes); put(c, bytes);
    printf("%-28s first %.3f ms (%.1f GB/s) | after free+alloc %.3f (%.1f) | third, larger, beside it %.3f (%.1f) | second again %.3f\n", names[k], ta, bytes / ta / 1e6, tb,
           bytes / tb / 1e6, tc, bytes / tc / 1e6, tb2);
  }
  return 0;
}

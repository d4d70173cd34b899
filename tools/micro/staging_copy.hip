// How fast can a 1.2 MB image be copied from pageable memory into pinned staging?  (mrh_upload_depth: the host-side
// cost that bounds GeoWrapper.setDepthImage)  memcpy vs non-temporal stores vs two threads, sources cold (220 distinct frames).
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void copy_nt(void* dst, const void* src, size_t n) {
  const __m256i* s = (const __m256i*) src; __m256i* d = (__m256i*) dst;
  for (size_t i = 0; i < n / 32; i++) _mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
  _mm_sfence();
}
int main() {
  const size_t n = 640 * 480 * 4, frames = 220;
  std::vector<char*> src(frames);
  for (auto& p : src) { p = (char*) malloc(n); memset(p, 1, n); }
  void* pin; hipHostMalloc(&pin, n, hipHostMallocDefault);
  for (int mode = 0; mode < 3; mode++) {
    double t0 = now();
    for (size_t f = 0; f < frames; f++) {
      if (mode == 0) memcpy(pin, src[f], n);
      else if (mode == 1) copy_nt(pin, src[f], n);
      else { std::thread t([&] { memcpy((char*) pin + n / 2, src[f] + n / 2, n / 2); }); memcpy(pin, src[f], n / 2); t.join(); }
    }
    double dt = (now() - t0) / frames;
    printf("%s: %.1f us per 1.2 MB image (%.1f GB/s)\n", mode == 0 ? "memcpy" : mode == 1 ? "avx2 non-temporal stores" : "memcpy on 2 threads (spawned per call)", dt, n / dt / 1e3);
  }
  return 0;
}

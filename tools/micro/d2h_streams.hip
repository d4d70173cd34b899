// Does a device-to-host copy run as fast on the process's LATER streams as on its first?  (looking for the 24 vs 54 GB/s of
// V / C / F in the second context of a process)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2hs tools/micro/d2h_streams.hip && /tmp/d2hs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(unsigned* p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = (unsigned) i; }
static double copy_ms(void* h, void* d, size_t bytes, hipStream_t s, bool with_kernel) {
  double best = 1e9;
  for (int i = 0; i < 6; i++) {
    if (with_kernel) k_touch<<<1024, 256, 0, s>>>((unsigned*) d, bytes / 4);
    hipStreamSynchronize(s);
    const double t0 = now();
    hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    const double dt = now() - t0;
    if (i && dt < best) best = dt;
  }
  return best;
}
int main(int argc, char** argv) {
  const size_t bytes = 12u << 20;
  const int nstreams = argc > 1 ? atoi(argv[1]) : 12;
  void *d, *h;
  hipMalloc(&d, bytes);
  hipMemset(d, 3, bytes);
  hipHostMalloc(&h, bytes, hipHostMallocDefault);
  printf("streams created one after the other, each destroyed before the next (a context's lifetime):\n");
  for (int i = 0; i < nstreams; i++) {
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);  // a context's copy stream
    void* h2d;
    hipHostMalloc(&h2d, bytes, hipHostMallocDefault);
    hipMemcpyAsync(d, h2d, bytes, hipMemcpyHostToDevice, s2);  // an upload on the second stream
    hipStreamSynchronize(s2);
    const double t = copy_ms(h, d, bytes, s, true);
    printf("  stream pair %2d: D2H %.3f ms (%.1f GB/s)\n", i, t, bytes / t / 1e6);
    hipHostFree(h2d);
    hipStreamDestroy(s);
    hipStreamDestroy(s2);
  }
  printf("streams alive together:\n");
  std::vector<hipStream_t> ss(nstreams);
  for (auto& s : ss) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int i = 0; i < nstreams; i++) {
    const double t = copy_ms(h, d, bytes, ss[i], true);
    printf("  stream %2d: D2H %.3f ms (%.1f GB/s)\n", i, t, bytes / t / 1e6);
  }
  return 0;
}

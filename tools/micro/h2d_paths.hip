// How fast can one 640x480 frame (1.2 MB depth + 0.9 MB colour, both in pinned host memory) reach HBM per frame of a loop?
// The host-input path of the library is bound by exactly this (tools/host_path_breakdown.py: 77 us per frame whatever the
// staging pool does).  Paths: hipMemcpyAsync on one stream (one SDMA engine), the two images on two streams, each image
// split over several streams, a copy kernel reading the pinned buffer over PCIe, and kernel + SDMA side by side.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/h2d_paths.hip -o /tmp/h2d_paths && /tmp/h2d_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t nd = 640 * 480 * 4, nc = 640 * 480 * 3;
  const int iters = 300, ring = 3;
  char *hd[ring], *hc[ring], *dd[ring], *dc[ring];
  for (int r = 0; r < ring; r++) {
    hipHostMalloc((void**) &hd[r], nd, hipHostMallocDefault); hipHostMalloc((void**) &hc[r], nc, hipHostMallocDefault);
    memset(hd[r], 1, nd); memset(hc[r], 2, nc);
    hipMalloc((void**) &dd[r], nd); hipMalloc((void**) &dc[r], nc);
  }
  hipStream_t s[8];
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 20; i++) body(i % ring);
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < iters; i++) body(i % ring);
    const double t_enq = now() - t0;
    hipDeviceSynchronize();
    const double dt = (now() - t0) / iters;
    printf("%-62s %6.1f us per frame (%5.1f GB/s), host enqueue %5.1f us\n", name, dt, (nd + nc) / dt / 1e3, t_enq / iters);
  };
  run("hipMemcpyAsync, both images on one stream", [&](int r) {
    hipMemcpyAsync(dd[r], hd[r], nd, hipMemcpyHostToDevice, s[0]); hipMemcpyAsync(dc[r], hc[r], nc, hipMemcpyHostToDevice, s[0]); });
  run("hipMemcpyAsync, depth on stream 0, colour on stream 1", [&](int r) {
    hipMemcpyAsync(dd[r], hd[r], nd, hipMemcpyHostToDevice, s[0]); hipMemcpyAsync(dc[r], hc[r], nc, hipMemcpyHostToDevice, s[1]); });
  run("hipMemcpyAsync, each image in halves on 4 streams", [&](int r) {
    hipMemcpyAsync(dd[r], hd[r], nd / 2, hipMemcpyHostToDevice, s[0]); hipMemcpyAsync(dd[r] + nd / 2, hd[r] + nd / 2, nd / 2, hipMemcpyHostToDevice, s[1]);
    hipMemcpyAsync(dc[r], hc[r], nc / 2, hipMemcpyHostToDevice, s[2]); hipMemcpyAsync(dc[r] + nc / 2, hc[r] + nc / 2, nc / 2, hipMemcpyHostToDevice, s[3]); });
  for (int grid : {64, 256, 1024}) {
    char name[96];
    snprintf(name, sizeof name, "copy kernel over PCIe (%d x 256 threads, 16 B per lane), one stream", grid);
    run(name, [&](int r) {
      hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s[0], (const uint4*) hd[r], (uint4*) dd[r], nd / 16);
      hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s[0], (const uint4*) hc[r], (uint4*) dc[r], nc / 16); });
  }
  run("copy kernel for depth (stream 0) + hipMemcpyAsync for colour (stream 1)", [&](int r) {
    hipLaunchKernelGGL(k_copy16, dim3(256), dim3(256), 0, s[0], (const uint4*) hd[r], (uint4*) dd[r], nd / 16);
    hipMemcpyAsync(dc[r], hc[r], nc, hipMemcpyHostToDevice, s[1]); });
  run("copy kernels, depth on stream 0, colour on stream 1", [&](int r) {
    hipLaunchKernelGGL(k_copy16, dim3(256), dim3(256), 0, s[0], (const uint4*) hd[r], (uint4*) dd[r], nd / 16);
    hipLaunchKernelGGL(k_copy16, dim3(256), dim3(256), 0, s[1], (const uint4*) hc[r], (uint4*) dc[r], nc / 16); });
  return 0;
}

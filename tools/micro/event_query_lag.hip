// How soon after a kernel has finished does hipEventQuery say so?  A ~17 us kernel on stream A, hipEventRecord, then the host
// spins T us without touching the runtime and asks once.  (Round 4: the pipelined frames skip the main stream's cross-stream
// wait when the front half's event is already complete; with the front half enqueued 70 us earlier half of the queries still
// said "not ready".)
//   hipcc -O2 --offload-arch=gfx950 tools/micro/event_query_lag.hip -o /tmp/eql && /tmp/eql
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_busy(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  for (int i = 0; i < iters; i++) x = fmaf(x, y, 1e-7f);
  if (x == 12345.f) out[0] = x;
}
int main() {
  float* d; (void) hipMalloc(&d, 64);
  hipStream_t A, B; (void) hipStreamCreateWithFlags(&A, hipStreamNonBlocking); (void) hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  for (unsigned flags : {(unsigned) hipEventDisableTiming, 0u}) {
    hipEvent_t ev; (void) hipEventCreateWithFlags(&ev, flags);
    for (int busyB = 0; busyB < 2; busyB++)
      for (double T : {20.0, 40.0, 80.0, 160.0, 400.0}) {
        int ready = 0, ready2 = 0;
        const int N = 200;
        for (int i = 0; i < N; i++) {
          if (busyB) k_busy<<<2048, 256, 0, B>>>(d, 745 * 8);  // a long kernel on the other stream, as k_back would be
          k_busy<<<1328, 256, 0, A>>>(d, 500);
          (void) hipEventRecord(ev, A);
          const double t0 = now();
          while (now() - t0 < T) {}
          ready += hipEventQuery(ev) == hipSuccess;
          ready2 += hipEventQuery(ev) == hipSuccess;  // a second look right away
          (void) hipDeviceSynchronize();
        }
        printf("flags 0x%x, other stream %s, query %5.0f us after the record: ready %3d / %d, second query %3d\n", flags, busyB ? "busy" : "idle", T, ready, N, ready2);
      }
  }
  return 0;
}

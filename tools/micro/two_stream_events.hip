// What do cross-stream dependencies cost in a software-pipelined frame loop?  Stream A runs a ~16 us latency-ish kernel (R, "rays"),
// stream B a short one (I, "insert") and a ~25 us busy one (K, "back").  Per frame g:  A: [wait back(g-2)] R(g) record ;
// B: wait rays(g), I(g), K(g), record.  Compared with everything on one stream, and with two streams and no events at all.
// Prints host enqueue time and wall time per frame.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/two_stream_events.hip -o /tmp/tse && /tmp/tse
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// busy: `iters` dependent FMAs per lane
__global__ void k_busy(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  for (int i = 0; i < iters; i++) x = fmaf(x, y, 1e-7f);
  if (x == 12345.f) out[0] = x;
}
// latency: a chain of dependent global loads (pointer chase) per lane, few waves
__global__ void k_chase(const int* __restrict__ next, int* out, int hops) {
  int p = (blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF;
  for (int i = 0; i < hops; i++) p = next[p];
  if (p == -1) out[0] = p;
}
int main() {
  float* d_out; hipMalloc(&d_out, 64);
  int* d_next; const int N = 1 << 20; hipMalloc(&d_next, N * 4);
  { int* h = new int[N]; for (int i = 0; i < N; i++) h[i] = (int) ((i * 2654435761u + 12345u) & (N - 1)); hipMemcpy(d_next, h, N * 4, hipMemcpyHostToDevice); delete[] h; }
  hipStream_t A, B; hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  hipEvent_t evR[4], evK[4];
  unsigned flags = hipEventDisableTiming;
  if (const char* e = getenv("EVFLAGS")) flags = (unsigned) strtoul(e, nullptr, 0);
  printf("event flags 0x%x\n", flags);
  for (auto& e : evR) if (hipEventCreateWithFlags(&e, flags) != hipSuccess) { printf("event flags refused\n"); return 1; }
  for (auto& e : evK) hipEventCreateWithFlags(&e, flags);
  // calibrate: R ~ 16 us on a full machine of 8-wave/SIMD (1328 WG x 256), K ~ 25 us (2048 WG x 256), I ~ 4 us latency-bound
  auto R = [&](hipStream_t s) { k_busy<<<1328, 256, 0, s>>>(d_out, 500); };
  auto I = [&](hipStream_t s) { k_chase<<<428, 256, 0, s>>>(d_next, (int*) d_out, 4); };
  auto K = [&](hipStream_t s) { k_busy<<<2048, 256, 0, s>>>(d_out, 745); };
  auto time1 = [&](const char* name, auto&& f) {
    for (int i = 0; i < 5; i++) f(A);
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < 50; i++) f(A);
    hipDeviceSynchronize();
    printf("  %-8s alone: %6.2f us\n", name, (now() - t0) / 50);
  };
  time1("R", R); time1("I", I); time1("K", K);
  const int iters = 400;
  auto run = [&](const char* name, auto&& body) {
    for (int g = 0; g < 20; g++) body(g);
    hipDeviceSynchronize();
    const double t0 = now();
    for (int g = 20; g < 20 + iters; g++) body(g);
    const double t_enq = now() - t0;
    hipDeviceSynchronize();
    printf("%-78s %6.2f us per frame, host enqueue %5.2f us\n", name, (now() - t0) / iters, t_enq / iters);
  };
  run("one stream: R I K", [&](int) { R(B); I(B); K(B); });
  run("two streams, NO events (not a valid schedule; upper bound of the overlap)", [&](int) { R(A); I(B); K(B); });
  run("two streams, B waits rays(g); A waits back(g-2)", [&](int g) {
    if (g >= 22) hipStreamWaitEvent(A, evK[(g - 2) & 3], 0);
    R(A); hipEventRecord(evR[g & 3], A);
    hipStreamWaitEvent(B, evR[g & 3], 0);
    I(B); K(B); hipEventRecord(evK[g & 3], B);
  });
  run("two streams, B waits rays(g) only", [&](int g) {
    R(A); hipEventRecord(evR[g & 3], A);
    hipStreamWaitEvent(B, evR[g & 3], 0);
    I(B); K(B);
  });
  run("lazy shape: A: [wait back(g-2)] F(g)=R+I ; B: wait F(g-1)... K(g-1)", [&](int g) {
    if (g >= 22) hipStreamWaitEvent(A, evK[(g - 2) & 3], 0);
    R(A); I(A); hipEventRecord(evR[g & 3], A);
    if (g >= 21) hipStreamWaitEvent(B, evR[(g - 1) & 3], 0);
    K(B); hipEventRecord(evK[(g - 1) & 3], B);
  });
  run("ONE stream, R launched with hipExtAnyOrderLaunch behind K (lazy shape: K(g-1) || R(g)+I(g))", [&](int) {
    K(B);
    hipExtLaunchKernelGGL(k_busy, dim3(1328), dim3(256), 0, B, nullptr, nullptr, hipExtAnyOrderLaunch, d_out, 500);
    hipExtLaunchKernelGGL(k_chase, dim3(428), dim3(256), 0, B, nullptr, nullptr, hipExtAnyOrderLaunch, (const int*) d_next, (int*) d_out, 4);
  });
  run("ONE stream, all three any-order (no ordering at all: bound)", [&](int) {
    hipExtLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, B, nullptr, nullptr, hipExtAnyOrderLaunch, d_out, 745);
    hipExtLaunchKernelGGL(k_busy, dim3(1328), dim3(256), 0, B, nullptr, nullptr, hipExtAnyOrderLaunch, d_out, 500);
    hipExtLaunchKernelGGL(k_chase, dim3(428), dim3(256), 0, B, nullptr, nullptr, hipExtAnyOrderLaunch, (const int*) d_next, (int*) d_out, 4);
  });
  run("lazy, A never waits: A: R(g) I(g) record ; B: wait front(g), K(g)", [&](int g) {
    R(A); I(A); hipEventRecord(evR[g & 3], A);
    hipStreamWaitEvent(B, evR[g & 3], 0);
    K(B);
  });
  run("the same, R and I on A as ONE kernel-ish (R only), K on B", [&](int g) {
    R(A); hipEventRecord(evR[g & 3], A);
    hipStreamWaitEvent(B, evR[g & 3], 0);
    K(B);
  });
  // host cost of the API calls alone
  {
    const double t0 = now();
    for (int i = 0; i < 2000; i++) hipEventRecord(evR[i & 3], A);
    const double t1 = now();
    for (int i = 0; i < 2000; i++) hipStreamWaitEvent(B, evR[i & 3], 0);
    const double t2 = now();
    hipDeviceSynchronize();
    printf("host cost: hipEventRecord %.2f us, hipStreamWaitEvent %.2f us (2000 calls each, idle streams)\n", (t1 - t0) / 2000, (t2 - t1) / 2000);
  }
  return 0;
}

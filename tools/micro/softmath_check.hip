// device-vs-host check of include/mrh_softmath.h: both sides must produce the same bits
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/micro/softmath_check.hip -o /tmp/smc && /tmp/smc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/mrh_softmath.h"
__global__ void k(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c;
  mrh_sincosf(a[i], &s, &c);
  out[4 * i] = s; out[4 * i + 1] = c; out[4 * i + 2] = mrh_atan2f(a[i], b[i]); out[4 * i + 3] = mrh_asinf(b[i] * 0.3f);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> a(n), b(n), o(4 * n);
  unsigned st = 12345;
  for (int i = 0; i < n; i++) { st = st * 1664525u + 1013904223u; a[i] = ((st >> 8) / 16777216.0f - 0.5f) * 6.6f; st = st * 1664525u + 1013904223u; b[i] = ((st >> 8) / 16777216.0f - 0.5f) * 6.6f; }
  float *da, *db, *dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, 16 * (size_t) n);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dout, n);
  hipMemcpy(o.data(), dout, 16 * (size_t) n, hipMemcpyDeviceToHost);
  long bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    float s, c;
    mrh_sincosf(a[i], &s, &c);
    const float h[4] = {s, c, mrh_atan2f(a[i], b[i]), mrh_asinf(b[i] * 0.3f)};
    for (int j = 0; j < 4; j++) if (memcmp(&h[j], &o[4 * i + j], 4) != 0) { if (bad[j]++ < 3) printf("fn %d a %.9g b %.9g host %.9g dev %.9g\n", j, a[i], b[i], h[j], o[4 * i + j]); }
  }
  printf("mismatches sin %ld cos %ld atan2 %ld asin %ld of %d\n", bad[0], bad[1], bad[2], bad[3], n);
  return (bad[0] | bad[1] | bad[2] | bad[3]) ? 1 : 0;
}

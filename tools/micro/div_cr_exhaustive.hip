// Is   q = a*r;  e = fma(-b, q, a);  q' = fma(e, r, q)   the correctly rounded a / b for EVERY finite fp32 a when
// r = RN(1 / b) is the correctly rounded reciprocal (computed in fp64 on the host)?  Exhaustive over all 2^32 dividends
// for b = 1 .. 510 (the weight sums of the running mean) and for a few half-voxel sizes.  Counted separately: dividends
// with 2^-100 <= |a| <= 2^100 or a == 0 (no intermediate can underflow or overflow: the range the kernels use) and the rest.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o /tmp/divcr tools/micro/div_cr_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_check(const float b, const float r, unsigned long long* out) {
  // out[0] mismatches inside the working range, out[1] outside it (underflow / overflow of an intermediate, -0), out[2] first bad bits
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad = 0, bad_sub = 0;
  for (uint32_t k = 0; k < 4096; k++) {
    const uint32_t bits = tid * 4096u + k;
    const float a = __uint_as_float(bits);
    if (!(fabsf(a) < INFINITY)) continue;  // NaN / Inf
    const float ref = a / b;
    float q = a * r;
    const float e = fmaf(-b, q, a);
    q = fmaf(e, r, q);
    if (__float_as_uint(q) != __float_as_uint(ref)) {
      const float m = fabsf(a);
      if ((m >= 7.888609e-31f && m <= 1.2676506e30f) || bits == 0u) { bad++; atomicCAS(&out[2], 0ull, (unsigned long long) bits | 1ull << 40); }
      else bad_sub++;
    }
  }
  if (bad) atomicAdd(&out[0], bad);
  if (bad_sub) atomicAdd(&out[1], bad_sub);
}

static float rn_reciprocal(float b) {  // correctly rounded 1 / b: fp64 quotient, then the nearest of the three neighbouring floats
  const double x = 1.0 / (double) b;
  float c = (float) x, best = c;
  double err = fabs(1.0 - (double) c * (double) b);  // c * b is exact in fp64 (24 x 24 bits)
  for (float t : {nextafterf(c, 0.f), nextafterf(c, INFINITY)}) {
    const double e = fabs(1.0 - (double) t * (double) b);
    if (e < err) { err = e; best = t; }
  }
  return best;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 24);
  std::vector<float> divisors;
  for (int w = 1; w <= 510; w++) divisors.push_back((float) w);
  for (float vs : {0.01f, 0.02f, 0.005f, 0.2f, 0.015f, 0.03f, 0.25f, 0.1f, 0.35f, 0.004f}) divisors.push_back(vs / 2);
  unsigned long long total_bad = 0, total_sub = 0;
  for (float b : divisors) {
    hipMemset(d, 0, 24);
    k_check<<<4096, 256>>>(b, rn_reciprocal(b), d);
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    if (h[0]) printf("b = %.9g: %llu mismatches in range, first bad dividend bits 0x%08llx\n", b, h[0], h[2] & 0xFFFFFFFFull);
    total_bad += h[0]; total_sub += h[1];
  }
  printf("%zu divisors x 2^32 dividends: %llu mismatches for 2^-100 <= |a| <= 2^100, %llu outside that range\n", divisors.size(), total_bad, total_sub);
  return 0;
}

// Microbenchmark: cycles per wave64 instruction for the VALU op classes the fusion kernels use (gfx950), measured with
// 4 waves per SIMD on every SIMD, 6 independent chains per lane (and one dependent chain for latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLK_GHZ 2.28
template <int OP>
__global__ void k_op(float* out, int iters) {
  float a = threadIdx.x * 1e-3f + 1.f, d = a + 1.f, e = a + 2.f, f = a + 3.f, g = a + 4.f, h = a + 5.f;
  const float b = 1.0001f, c = 0.5f;
  unsigned ia = threadIdx.x + 1, id = ia + 1, ie = ia + 2, iff = ia + 3, ig = ia + 4, ih = ia + 5;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 pa = {a, d}, pd = {e, f}, pe = {g, h}, pf = {a + 6, a + 7}, pg = {a + 8, a + 9}, ph = {a + 10, a + 11};
  const v2 pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (OP == 0) { a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c); g = __builtin_fmaf(g, b, c); h = __builtin_fmaf(h, b, c); }
      if (OP == 1) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia) : "v"(id)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(id) : "v"(ie)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ie) : "v"(iff)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(iff) : "v"(ig)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ig) : "v"(ih)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ih) : "v"(ia)); }
      if (OP == 2) { pa = __builtin_elementwise_fma(pa, pb, pc); pd = __builtin_elementwise_fma(pd, pb, pc); pe = __builtin_elementwise_fma(pe, pb, pc); pf = __builtin_elementwise_fma(pf, pb, pc); pg = __builtin_elementwise_fma(pg, pb, pc); ph = __builtin_elementwise_fma(ph, pb, pc); }
      if (OP == 3) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); asm volatile("v_rcp_f32 %0, %0" : "+v"(d)); asm volatile("v_rcp_f32 %0, %0" : "+v"(e)); asm volatile("v_rcp_f32 %0, %0" : "+v"(f)); asm volatile("v_rcp_f32 %0, %0" : "+v"(g)); asm volatile("v_rcp_f32 %0, %0" : "+v"(h)); }
      if (OP == 4) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia) : "v"(id)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(id) : "v"(ie)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ie) : "v"(iff)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(iff) : "v"(ig)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ig) : "v"(ih)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ih) : "v"(ia)); }
      if (OP == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(id)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(id) : "v"(ie)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ie) : "v"(iff)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iff) : "v"(ig)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ig) : "v"(ih)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ih) : "v"(ia)); }
      if (OP == 6) { asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ia) : "v"(a)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a) : "v"(ia)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(id) : "v"(d)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(d) : "v"(id)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ie) : "v"(e)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(e) : "v"(ie)); }
      if (OP == 7) { a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); a = __builtin_fmaf(a, b, c); }  // dependent chain
      if (OP == 8) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(d) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(d), "v"(e) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(e), "v"(f) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f), "v"(g) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(g), "v"(h) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(h), "v"(a) : "vcc"); }
      if (OP == 9) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia) : "v"(id)); asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(id)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(ie) : "v"(iff)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iff) : "v"(ig)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(ig) : "v"(ih)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(d)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h + ia + id + ie + iff + ig + ih + pa.x + pd.y + pe.x + pf.y + pg.x + ph.y;
}
template <int OP>
void run(const char* name, float* out, int wps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, grid = 256 * wps;
  k_op<OP><<<grid, 256>>>(out, iters); hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 3; r++) { hipEventRecord(e0); k_op<OP><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  const double insts = (double) iters * 96 * wps;
  printf("%-28s waves/simd %d: %.2f clk per wave64 instruction (at %.2f GHz)\n", name, wps, best * 1e-3 * CLK_GHZ * 1e9 / insts, CLK_GHZ);
}
int main() {
  float* out; (void) hipMalloc(&out, 4096 * 256 * 4);
  for (int wps : {1, 4}) {
    run<0>("v_fma_f32 (6 chains)", out, wps); run<7>("v_fma_f32 (1 dependent chain)", out, wps); run<2>("v_pk_fma_f32", out, wps);
    run<1>("v_add_u32", out, wps); run<9>("and/shift/or/xor/max/min mix", out, wps); run<5>("v_cndmask_b32", out, wps); run<8>("v_cmp_lt_f32", out, wps);
    run<6>("v_cvt i32<->f32", out, wps); run<4>("v_mul_lo_u32", out, wps); run<3>("v_rcp_f32", out, wps);
  }
  return 0;
}

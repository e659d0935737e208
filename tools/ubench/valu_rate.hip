// Microbenchmark: issue cost of the integer VALU ops the flood uses vs v_fma_f32, per wave64, on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 512
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed, int iters) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
      if (OP == 1) { a0 = (a0 & a1) | (~a0 & a2); a1 = (a1 & a2) | (~a1 & a3); a2 = (a2 & a3) | (~a2 & a4); a3 = (a3 & a4) | (~a3 & a5);
                     a4 = (a4 & a5) | (~a4 & a6); a5 = (a5 & a6) | (~a5 & a7); a6 = (a6 & a7) | (~a6 & a0); a7 = (a7 & a0) | (~a7 & a1); }
      if (OP == 2) { a0 = __brev(a0) ^ a1; a1 = __brev(a1) ^ a2; a2 = __brev(a2) ^ a3; a3 = __brev(a3) ^ a4; a4 = __brev(a4) ^ a5; a5 = __brev(a5) ^ a6; a6 = __brev(a6) ^ a7; a7 = __brev(a7) ^ a0; }
      if (OP == 3) { f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f4); f3 = fmaf(f3, f4, f5); f4 = fmaf(f4, f5, f6); f5 = fmaf(f5, f6, f7); f6 = fmaf(f6, f7, f0); f7 = fmaf(f7, f0, f1); }
      if (OP == 4) { a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a4; a4 = a4 * a5; a5 = a5 * a6; a6 = a6 * a7; a7 = a7 * a0; }
      if (OP == 5) { a0 = (a0 & a1) | a2; a1 = (a1 & a2) | a3; a2 = (a2 & a3) | a4; a3 = (a3 & a4) | a5; a4 = (a4 & a5) | a6; a5 = (a5 & a6) | a7; a6 = (a6 & a7) | a0; a7 = (a7 & a0) | a1; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}
template <int OP>
void run(const char *name, int waves_per_simd) {
  int cus = 256, iters = 200;
  int blocks = cus * 4 * waves_per_simd;
  uint32_t *out; hipMalloc(&out, blocks * 64 * 4);
  k<OP><<<blocks, 64>>>(out, 1, 2);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<OP><<<blocks, 64>>>(out, 1, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double inst_per_simd = (double)waves_per_simd * iters * REP * (OP == 2 ? 2 : 1);
  double ns_per_inst = ms * 1e6 / inst_per_simd;
  printf("%-12s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD  (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("v_add_u32", w); run<1>("v_bfi/bitop3", w); run<2>("bfrev+xor", w); run<3>("v_fma_f32", w); run<4>("v_mul_lo", w); run<5>("v_and_or", w);
  }
  return 0;
}

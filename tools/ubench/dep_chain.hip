// dep_chain.hip - what a DEPENDENT chain of VALU instructions costs per instruction with 1, 2, 3, 4, 8 waves on a SIMD
// (tools/ubench/valu_rate2.hip measures independent streams: the issue rate).  Chains: v_add (2-cycle op), v_bfrev (4-cycle op),
// the flood's visit (bitop3, add, bitop3, bfrev, add, bitop3: gg_common.h FLOOD_VISIT), and the visit as TWO interleaved chains.
//   hipcc -O3 --offload-arch=gfx950 -o dep_chain dep_chain.hip && ./dep_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define B3(a, b, c, t) __builtin_amdgcn_bitop3_b32((a), (b), (c), (t))
constexpr int kIters = 2048;

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed) {
  uint32_t x = seed + threadIdx.x, y = seed * 3u + 1u, m = ~seed, mr = seed ^ 0x55u, x2 = x ^ 77u;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); }
      if (MODE == 1) { asm volatile("v_bfrev_b32 %0, %0" : "+v"(x)); }
      if (MODE == 4) {   // the visit as two chains interleaved INSTRUCTION BY INSTRUCTION (inline asm: the compiler keeps whole visits apart)
        uint32_t ta, tb;
        asm volatile("v_bitop3_b32 %0, %4, %5, %0 bitop3:0xea\n\tv_bitop3_b32 %1, %4, %5, %1 bitop3:0xea\n\t"
                     "v_add_u32 %2, %5, %0\n\tv_add_u32 %3, %5, %1\n\t"
                     "v_bitop3_b32 %0, %2, %0, %5 bitop3:0xca\n\tv_bitop3_b32 %1, %3, %1, %5 bitop3:0xca\n\t"
                     "v_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\t"
                     "v_add_u32 %2, %6, %0\n\tv_add_u32 %3, %6, %1\n\t"
                     "v_bitop3_b32 %0, %2, %0, %6 bitop3:0xca\n\tv_bitop3_b32 %1, %3, %1, %6 bitop3:0xca"
                     : "+v"(x), "+v"(x2), "=&v"(ta), "=&v"(tb) : "v"(y), "v"(m), "v"(mr));
      }
      if (MODE == 5) {   // one chain, the same twelve instructions back to back (the asm form of MODE 2, two visits)
        uint32_t ta;
        asm volatile("v_bitop3_b32 %0, %2, %3, %0 bitop3:0xea\n\tv_add_u32 %1, %3, %0\n\tv_bitop3_b32 %0, %1, %0, %3 bitop3:0xca\n\t"
                     "v_bfrev_b32 %0, %0\n\tv_add_u32 %1, %4, %0\n\tv_bitop3_b32 %0, %1, %0, %4 bitop3:0xca\n\t"
                     "v_bitop3_b32 %0, %2, %3, %0 bitop3:0xea\n\tv_add_u32 %1, %3, %0\n\tv_bitop3_b32 %0, %1, %0, %3 bitop3:0xca\n\t"
                     "v_bfrev_b32 %0, %0\n\tv_add_u32 %1, %4, %0\n\tv_bitop3_b32 %0, %1, %0, %4 bitop3:0xca"
                     : "+v"(x), "=&v"(ta) : "v"(y), "v"(m), "v"(mr));
      }
      if (MODE == 6) {   // twelve INDEPENDENT two-cycle instructions (six v_add, six v_bitop3 on six registers): the issue rate of a wave
        uint32_t a0 = x, a1 = x2, a2 = y, a3 = m, a4 = mr, a5 = seed;
        asm volatile("v_add_u32 %0, %0, %6\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %6\n\tv_add_u32 %4, %4, %6\n\tv_add_u32 %5, %5, %6\n\t"
                     "v_bitop3_b32 %0, %0, %6, %7 bitop3:0xca\n\tv_bitop3_b32 %1, %1, %6, %7 bitop3:0xca\n\tv_bitop3_b32 %2, %2, %6, %7 bitop3:0xca\n\t"
                     "v_bitop3_b32 %3, %3, %6, %7 bitop3:0xca\n\tv_bitop3_b32 %4, %4, %6, %7 bitop3:0xca\n\tv_bitop3_b32 %5, %5, %6, %7 bitop3:0xca"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(y), "v"(m));
        x = a0 ^ a2 ^ a4; x2 = a1 ^ a3 ^ a5;
      }
      if (MODE == 2 || MODE == 3) {
        uint32_t s = B3(y, m, x, 0xEA), t = m + s, uu = B3(t, s, m, 0xCA), v = __brev(uu), t2 = mr + v;
        x = B3(t2, v, mr, 0xCA);
        asm volatile("" : "+v"(x));
        if (MODE == 3) {
          uint32_t s_ = B3(y, m, x2, 0xEA), t_ = m + s_, u_ = B3(t_, s_, m, 0xCA), v_ = __brev(u_), t2_ = mr + v_;
          x2 = B3(t2_, v_, mr, 0xCA);
          asm volatile("" : "+v"(x2));
        }
      }
    }
  }
  if (x == 0x12345u && x2 == 7u) out[0] = x;
}

template <int MODE>
void run(const char *name, int instr_per_iter) {
  uint32_t *out;
  (void)hipMalloc(&out, 4);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int simds = p.multiProcessorCount * 4;
  for (int w : {1, 2, 3, 4, 8}) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<simds * w, 64>>>(out, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 4; ++r) k<MODE><<<simds * w, 64>>>(out, 12345u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / 4.0 / ((double)kIters * 8 * instr_per_iter);
    printf("%-28s %d waves per SIMD: %.2f ns per instruction of a wave, %.2f ns per instruction of the SIMD\n", name, w, ns, ns / w);
  }
}

int main() {
  run<0>("v_add_u32 chain", 1);
  run<1>("v_bfrev_b32 chain", 1);
  run<2>("flood visit, one chain", 6);
  run<3>("flood visit, two chains", 12);
  run<5>("visit x 2, asm, one chain", 12);
  run<4>("visit x 2, asm, interleaved", 12);
  run<6>("12 independent 2-cycle ops", 12);
  return 0;
}

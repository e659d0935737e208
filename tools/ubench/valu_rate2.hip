// Per-op issue cost (wave64, gfx950) of the VALU instructions used by the flood / analysis code, via inline asm
// so the compiler cannot substitute. 8 independent chains, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define STR(x) #x
#define RUN8(INS) \
  asm volatile(INS(0, 1, 2) INS(1, 2, 3) INS(2, 3, 4) INS(3, 4, 5) INS(4, 5, 6) INS(5, 6, 7) INS(6, 7, 0) INS(7, 0, 1) \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#define I_XOR(d, x, y) "v_xor_b32 %" #d ", %" #x ", %" #y "\n"
#define I_AND(d, x, y) "v_and_b32 %" #d ", %" #x ", %" #y "\n"
#define I_ADD(d, x, y) "v_add_u32 %" #d ", %" #x ", %" #y "\n"
#define I_BFREV(d, x, y) "v_bfrev_b32 %" #d ", %" #x "\n"
#define I_ANDOR(d, x, y) "v_and_or_b32 %" #d ", %" #d ", %" #x ", %" #y "\n"
#define I_OR3(d, x, y) "v_or3_b32 %" #d ", %" #d ", %" #x ", %" #y "\n"
#define I_LSHLOR(d, x, y) "v_lshl_or_b32 %" #d ", %" #x ", 1, %" #y "\n"
#define I_LSHR(d, x, y) "v_lshrrev_b32 %" #d ", 1, %" #x "\n"
#define I_LSHL(d, x, y) "v_lshlrev_b32 %" #d ", 1, %" #x "\n"
#define I_BITOP3(d, x, y) "v_bitop3_b32 %" #d ", %" #d ", %" #x ", %" #y " bitop3:0xca\n"
#define I_BFI(d, x, y) "v_bfi_b32 %" #d ", %" #d ", %" #x ", %" #y "\n"
#define I_BCNT(d, x, y) "v_bcnt_u32_b32 %" #d ", %" #x ", %" #y "\n"
#define I_MOV(d, x, y) "v_mov_b32 %" #d ", %" #x "\n"
#define I_CNDMASK(d, x, y) "v_cndmask_b32 %" #d ", %" #x ", %" #y ", vcc\n"
#define I_ADD3(d, x, y) "v_add3_u32 %" #d ", %" #d ", %" #x ", %" #y "\n"
#define I_MUL24(d, x, y) "v_mul_u32_u24 %" #d ", %" #x ", %" #y "\n"
#define I_DOT4(d, x, y) "v_dot4_u32_u8 %" #d ", %" #x ", %" #y ", %" #d "\n"
#define I_ALIGNBIT(d, x, y) "v_alignbit_b32 %" #d ", %" #x ", %" #y ", 1\n"
#define I_XAD(d, x, y) "v_xad_u32 %" #d ", %" #d ", %" #x ", %" #y "\n"
#define I_LSHLADD(d, x, y) "v_lshl_add_u32 %" #d ", %" #x ", 1, %" #y "\n"
#define I_OR(d, x, y) "v_or_b32 %" #d ", %" #x ", %" #y "\n"
#define I_SUB(d, x, y) "v_sub_u32 %" #d ", %" #x ", %" #y "\n"
#define I_FFBL(d, x, y) "v_ffbl_b32 %" #d ", %" #x "\n"
#define I_BFE(d, x, y) "v_bfe_u32 %" #d ", %" #x ", 3, 5\n"
#define I_READLANE(d, x, y) "v_readlane_b32 s20, %" #x ", 3\n"
#define I_PERM(d, x, y) "v_perm_b32 %" #d ", %" #d ", %" #x ", %" #y "\n"

#define KERNEL(NAME, INS)                                                 \
  __global__ void NAME(uint32_t *out, uint32_t seed, int iters) {         \
    uint32_t a[8];                                                        \
    for (int i = 0; i < 8; ++i) a[i] = (threadIdx.x + seed) * (2 * i + 3); \
    for (int it = 0; it < iters; ++it) {                                  \
      _Pragma("unroll") for (int r = 0; r < 64; ++r) { RUN8(INS) }        \
    }                                                                     \
    uint32_t x = 0;                                                       \
    for (int i = 0; i < 8; ++i) x ^= a[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;                       \
  }
KERNEL(k_xor, I_XOR) KERNEL(k_and, I_AND) KERNEL(k_add, I_ADD) KERNEL(k_bfrev, I_BFREV) KERNEL(k_andor, I_ANDOR)
KERNEL(k_or3, I_OR3) KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_lshr, I_LSHR) KERNEL(k_lshl, I_LSHL) KERNEL(k_bitop3, I_BITOP3)
KERNEL(k_bfi, I_BFI) KERNEL(k_bcnt, I_BCNT) KERNEL(k_mov, I_MOV) KERNEL(k_cndmask, I_CNDMASK) KERNEL(k_add3, I_ADD3)
KERNEL(k_mul24, I_MUL24) KERNEL(k_dot4, I_DOT4) KERNEL(k_alignbit, I_ALIGNBIT) KERNEL(k_xad, I_XAD) KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_or, I_OR) KERNEL(k_sub, I_SUB) KERNEL(k_ffbl, I_FFBL) KERNEL(k_bfe, I_BFE) KERNEL(k_readlane, I_READLANE) KERNEL(k_perm, I_PERM)

typedef void (*kern_t)(uint32_t *, uint32_t, int);
double base_ns = 0;
void run(const char *name, kern_t k) {
  const int wps = 4, iters = 100, blocks = 256 * 4 * wps;
  uint32_t *out; (void)hipMalloc(&out, blocks * 64 * 4);
  k<<<blocks, 64>>>(out, 1, 2); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0); k<<<blocks, 64>>>(out, 1, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double ns = best * 1e6 / ((double)wps * iters * 64 * 8);
  if (base_ns == 0) base_ns = ns;
  printf("%-14s %.3f ns per wave-instruction per SIMD   x%.2f of v_xor\n", name, ns, ns / base_ns);
  (void)hipFree(out);
}
int main() {
  run("v_xor_b32", k_xor); run("v_and_b32", k_and); run("v_or_b32", k_or); run("v_add_u32", k_add); run("v_sub_u32", k_sub);
  run("v_bfrev_b32", k_bfrev); run("v_and_or_b32", k_andor); run("v_or3_b32", k_or3); run("v_lshl_or_b32", k_lshlor);
  run("v_lshrrev_b32", k_lshr); run("v_lshlrev_b32", k_lshl); run("v_bitop3_b32", k_bitop3); run("v_bfi_b32", k_bfi);
  run("v_bcnt_u32", k_bcnt); run("v_mov_b32", k_mov); run("v_cndmask_b32", k_cndmask); run("v_add3_u32", k_add3);
  run("v_mul_u32_u24", k_mul24); run("v_dot4_u32_u8", k_dot4); run("v_alignbit_b32", k_alignbit); run("v_xad_u32", k_xad);
  run("v_lshl_add_u32", k_lshladd); run("v_ffbl_b32", k_ffbl); run("v_bfe_u32", k_bfe); run("v_readlane_b32", k_readlane);
  run("v_perm_b32", k_perm);
  return 0;
}

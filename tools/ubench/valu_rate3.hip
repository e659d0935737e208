// Issue cost (wave64, gfx950) of selects, compares, 32-bit multiplies and DPP moves: the instructions of the lane-per-board
// phases of k_rollout3. Inline asm so the compiler cannot substitute; 8 independent chains, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define RUN8(INS) \
  asm volatile(INS(0, 1, 2) INS(1, 2, 3) INS(2, 3, 4) INS(3, 4, 5) INS(4, 5, 6) INS(5, 6, 7) INS(6, 7, 0) INS(7, 0, 1) \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])       \
               :: "vcc", "s20", "s21", "s22", "s23");
#define I_XOR(d, x, y) "v_xor_b32 %" #d ", %" #x ", %" #y "\n"
#define I_CND_VCC(d, x, y) "v_cndmask_b32 %" #d ", %" #x ", %" #y ", vcc\n"
#define I_CND_SGPR(d, x, y) "v_cndmask_b32_e64 %" #d ", %" #x ", %" #y ", s[20:21]\n"
#define I_CND_CONST(d, x, y) "v_cndmask_b32_e64 %" #d ", 0, %" #y ", s[20:21]\n"
#define I_CMP_VCC(d, x, y) "v_cmp_lt_u32 vcc, %" #x ", %" #y "\n"
#define I_CMP_SGPR(d, x, y) "v_cmp_lt_u32_e64 s[22:23], %" #x ", %" #y "\n"
#define I_CMP_CND(d, x, y) "v_cmp_lt_u32 vcc, %" #x ", %" #y "\n s_nop 1\n v_cndmask_b32 %" #d ", %" #x ", %" #y ", vcc\n"
#define I_CMP_CND_E64(d, x, y) "v_cmp_lt_u32_e64 s[22:23], %" #x ", %" #y "\n s_nop 1\n v_cndmask_b32_e64 %" #d ", %" #x ", %" #y ", s[22:23]\n"
#define I_MIN(d, x, y) "v_min_u32 %" #d ", %" #x ", %" #y "\n"
#define I_MULLO(d, x, y) "v_mul_lo_u32 %" #d ", %" #x ", %" #y "\n"
#define I_MULHI(d, x, y) "v_mul_hi_u32 %" #d ", %" #x ", %" #y "\n"
#define I_MAD24(d, x, y) "v_mad_u32_u24 %" #d ", %" #x ", %" #y ", %" #d "\n"
#define I_DPP(d, x, y) "v_mov_b32_dpp %" #d ", %" #x " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPP_WAVE(d, x, y) "v_mov_b32_dpp %" #d ", %" #x " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_XOR_DPP(d, x, y) "v_xor_b32_dpp %" #d ", %" #x ", %" #y " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_SUBB(d, x, y) "v_sub_co_u32 %" #d ", vcc, %" #x ", %" #y "\n"
#define I_ADDC(d, x, y) "v_addc_co_u32 %" #d ", vcc, %" #x ", %" #y ", vcc\n"
#define I_LSHL64(d, x, y) "v_lshlrev_b32 %" #d ", %" #x ", %" #y "\n"
#define I_LSHR_V(d, x, y) "v_lshrrev_b32 %" #d ", %" #x ", %" #y "\n"
#define I_BFE_V(d, x, y) "v_bfe_u32 %" #d ", %" #x ", %" #y ", 1\n"
#define CND3_VCC(d, x, y) "v_cndmask_b32 %" #d ", %" #x ", %" #y ", vcc\n v_cndmask_b32 %" #x ", %" #y ", %" #d ", vcc\n v_cndmask_b32 %" #y ", %" #d ", %" #x ", vcc\n"
#define CND3_SGPR(d, x, y) "v_cndmask_b32_e64 %" #d ", %" #x ", %" #y ", s[22:23]\n v_cndmask_b32_e64 %" #x ", %" #y ", %" #d ", s[22:23]\n v_cndmask_b32_e64 %" #y ", %" #d ", %" #x ", s[22:23]\n"
#define I_SMOV_CND3(d, x, y) "s_mov_b64 vcc, s[20:21]\n s_nop 1\n" CND3_VCC(d, x, y)
#define I_SAND_CND3(d, x, y) "s_and_b64 vcc, vcc, s[20:21]\n s_nop 1\n" CND3_VCC(d, x, y)
#define I_SAND_CND3_SGPR(d, x, y) "s_and_b64 s[22:23], s[20:21], s[20:21]\n s_nop 1\n" CND3_SGPR(d, x, y)
#define I_VCMP_CND3(d, x, y) "v_cmp_lt_u32 vcc, %" #x ", %" #y "\n s_nop 1\n" CND3_VCC(d, x, y)
#define I_VCMP_CND3_SGPR(d, x, y) "v_cmp_lt_u32_e64 s[22:23], %" #x ", %" #y "\n s_nop 1\n" CND3_SGPR(d, x, y)
#define I_VCMP_SAND_CND3(d, x, y) "v_cmp_lt_u32 vcc, %" #x ", %" #y "\n s_and_b64 vcc, vcc, s[20:21]\n s_nop 1\n" CND3_VCC(d, x, y)

#define KERNEL(NAME, INS)                                                 \
  __global__ void NAME(uint32_t *out, uint32_t seed, int iters) {         \
    uint32_t a[8];                                                        \
    for (int i = 0; i < 8; ++i) a[i] = (threadIdx.x + seed) * (2 * i + 3); \
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc\n" :: "v"(a[0]), "v"(a[1]) : "vcc", "s20", "s21"); \
    for (int it = 0; it < iters; ++it) {                                  \
      _Pragma("unroll") for (int r = 0; r < 64; ++r) { RUN8(INS) }        \
    }                                                                     \
    uint32_t x = 0;                                                       \
    for (int i = 0; i < 8; ++i) x ^= a[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;                       \
  }
KERNEL(k_xor, I_XOR) KERNEL(k_cnd_vcc, I_CND_VCC) KERNEL(k_cnd_sgpr, I_CND_SGPR) KERNEL(k_cnd_const, I_CND_CONST)
KERNEL(k_cmp_vcc, I_CMP_VCC) KERNEL(k_cmp_sgpr, I_CMP_SGPR) KERNEL(k_cmp_cnd, I_CMP_CND) KERNEL(k_cmp_cnd_e64, I_CMP_CND_E64)
KERNEL(k_min, I_MIN) KERNEL(k_mullo, I_MULLO) KERNEL(k_mulhi, I_MULHI) KERNEL(k_mad24, I_MAD24) KERNEL(k_dpp, I_DPP)
KERNEL(k_dpp_wave, I_DPP_WAVE) KERNEL(k_xor_dpp, I_XOR_DPP) KERNEL(k_subb, I_SUBB) KERNEL(k_addc, I_ADDC)
KERNEL(k_smov_cnd3, I_SMOV_CND3) KERNEL(k_sand_cnd3, I_SAND_CND3) KERNEL(k_sand_cnd3_sgpr, I_SAND_CND3_SGPR)
KERNEL(k_vcmp_cnd3, I_VCMP_CND3) KERNEL(k_vcmp_cnd3_sgpr, I_VCMP_CND3_SGPR) KERNEL(k_vcmp_sand_cnd3, I_VCMP_SAND_CND3)
KERNEL(k_lshl_v, I_LSHL64) KERNEL(k_lshr_v, I_LSHR_V) KERNEL(k_bfe_v, I_BFE_V)

typedef void (*kern_t)(uint32_t *, uint32_t, int);
double base_ns = 0;
void run(const char *name, kern_t k, int per) {
  const int wps = 4, iters = 100, blocks = 256 * 4 * wps;
  uint32_t *out; (void)hipMalloc(&out, blocks * 64 * 4);
  k<<<blocks, 64>>>(out, 1, 2); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0); k<<<blocks, 64>>>(out, 1, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); return; }
  double ns = best * 1e6 / ((double)wps * iters * 64 * 8);
  if (base_ns == 0) base_ns = ns;
  printf("%-34s %.3f ns per group of %d per SIMD   x%.2f of v_xor\n", name, ns, per, ns / base_ns);
  (void)hipFree(out);
}
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  run("v_xor_b32", k_xor, 1);
  run("v_cndmask_b32 (vcc)", k_cnd_vcc, 1); run("v_cndmask_b32_e64 (sgpr pair)", k_cnd_sgpr, 1);
  run("v_cndmask_b32_e64 0, v, sgpr", k_cnd_const, 1);
  run("v_cmp_lt_u32 -> vcc", k_cmp_vcc, 1); run("v_cmp_lt_u32_e64 -> sgpr", k_cmp_sgpr, 1);
  run("v_cmp + s_nop 1 + v_cndmask (vcc)", k_cmp_cnd, 3); run("v_cmp_e64 + s_nop 1 + v_cndmask_e64", k_cmp_cnd_e64, 3);
  run("v_min_u32", k_min, 1); run("v_mul_lo_u32", k_mullo, 1); run("v_mul_hi_u32", k_mulhi, 1); run("v_mad_u32_u24", k_mad24, 1);
  run("v_mov_b32_dpp row_shr:1", k_dpp, 1); run("v_mov_b32_dpp wave_shr:1", k_dpp_wave, 1); run("v_xor_b32_dpp row_shr:1", k_xor_dpp, 1);
  run("v_sub_co_u32 -> vcc", k_subb, 1); run("v_addc_co_u32 vcc -> vcc", k_addc, 1);
  run("v_lshlrev_b32 v, v", k_lshl_v, 1); run("v_lshrrev_b32 v, v", k_lshr_v, 1); run("v_bfe_u32 v, v, 1", k_bfe_v, 1);
  run("s_mov vcc + 3 v_cndmask (vcc)", k_smov_cnd3, 4); run("s_and vcc + 3 v_cndmask (vcc)", k_sand_cnd3, 4);
  run("s_and sgpr + 3 v_cndmask_e64", k_sand_cnd3_sgpr, 4); run("v_cmp vcc + 3 v_cndmask (vcc)", k_vcmp_cnd3, 4);
  run("v_cmp_e64 sgpr + 3 v_cndmask_e64", k_vcmp_cnd3_sgpr, 4); run("v_cmp vcc + s_and vcc + 3 v_cndmask", k_vcmp_sand_cnd3, 5);
  return 0;
}

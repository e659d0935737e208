// Where do the waves of a one-wave-per-workgroup launch land?  G workgroups of 64 threads with the resources of the
// multi-ply kernel (9 728 B LDS, 128 VGPRs: 4 waves per SIMD, 16 per CU) spin for a fixed time; every wave records its
// (XCC, SE, CU, SIMD).  Prints the histogram of waves per SIMD over the time all of them are resident.
//   hipcc -O3 --offload-arch=gfx950 placement.hip -o placement && ./placement [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(64, 4) void k(uint32_t *rec, long long spin) {
  __shared__ uint32_t lds[9728 / 4];
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hw));    // HW_REG_HW_ID
  asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xcc));  // HW_REG_XCC_ID
  lds[threadIdx.x] = hw;
  // hold 128 VGPRs
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) { lds[(threadIdx.x * 7 + 1) & 1023] += 1; }
  if (threadIdx.x == 0) { rec[2 * blockIdx.x] = hw; rec[2 * blockIdx.x + 1] = xcc + lds[5] * 0; }
}
int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 4096;
  uint32_t *d;
  hipMalloc(&d, 2 * G * sizeof(uint32_t));
  k<<<G, 64>>>(d, 20000);   // 200 us at 100 MHz
  hipDeviceSynchronize();
  std::vector<uint32_t> h(2 * G);
  hipMemcpy(h.data(), d, 2 * G * sizeof(uint32_t), hipMemcpyDeviceToHost);
  std::map<uint32_t, int> simd, cu;
  for (int i = 0; i < G; ++i) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
    const uint32_t simd_id = (hw >> 4) & 3, cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const uint32_t cukey = (xcc << 16) | (se << 8) | (sh << 7) | cu_id;
    cu[cukey]++;
    simd[(cukey << 2) | simd_id]++;
  }
  std::map<int, int> hs, hc;
  for (auto &kv : simd) hs[kv.second]++;
  for (auto &kv : cu) hc[kv.second]++;
  printf("G = %d workgroups: %zu distinct CUs, %zu distinct SIMDs seen\n", G, cu.size(), simd.size());
  printf("waves per CU   :");
  for (auto &kv : hc) printf("  %d x%d", kv.first, kv.second);
  printf("\nwaves per SIMD :");
  for (auto &kv : hs) printf("  %d x%d", kv.first, kv.second);
  printf("\n");
  return 0;
}

// Cost of 4-byte LDS accesses at unaligned byte addresses (lane stride 19 B, like a 19x19 board row) vs aligned.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct __attribute__((packed, aligned(1))) W32u { uint32_t v; };
template <int STRIDE, bool WRITE>
__global__ void k(uint32_t *out, int iters, int off) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint8_t)i;
  __syncthreads();
  uint32_t acc = 0;
  const int base = (threadIdx.x & 31) * STRIDE + off;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint8_t *p = lds + base + 4 * k + ((it & 1) ? 64 : 0);
      if (WRITE) reinterpret_cast<W32u *>(p)->v = acc + k;
      else acc += reinterpret_cast<const W32u *>(p)->v;
    }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc + lds[threadIdx.x];
}
template <int STRIDE, bool WRITE>
void run(const char *name, int off) {
  const int blocks = 256 * 16, iters = 2000;
  uint32_t *out; (void)hipMalloc(&out, blocks * 64 * 4);
  k<STRIDE, WRITE><<<blocks, 64>>>(out, 10, off); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<STRIDE, WRITE><<<blocks, 64>>>(out, iters, off); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double ns = ms * 1e6 / (16.0 /*blocks per CU*/ * iters * 16);
  printf("%-34s off=%d: %.2f ns per wave-instruction per CU\n", name, off, ns);
  (void)hipFree(out);
}
int main() {
  run<20, false>("read  stride 20 (aligned)", 0);
  run<19, false>("read  stride 19 (unaligned)", 0);
  run<19, false>("read  stride 19 (unaligned)", 6);
  run<16, false>("read  stride 16 (aligned)", 0);
  run<20, true>("write stride 20 (aligned)", 0);
  run<19, true>("write stride 19 (unaligned)", 0);
  run<16, true>("write stride 16 (aligned)", 0);
  return 0;
}

// ds_read_b32 / ds_write_b32 / ds_read_u8 at aligned vs unaligned byte addresses (inline asm: no merging).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void k(uint32_t *out, int iters, int stride, int off) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint8_t)i;
  __syncthreads();
  uint32_t acc = 0, t0, t1, t2, t3;
  uint32_t addr = (uint32_t)(uintptr_t)lds + (threadIdx.x & 31) * stride + off;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n s_waitcnt lgkmcnt(0)" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(addr));
    if (MODE == 1) asm volatile("ds_write_b32 %4, %0\n ds_write_b32 %4, %1 offset:4\n ds_write_b32 %4, %2 offset:8\n ds_write_b32 %4, %3 offset:12\n s_waitcnt lgkmcnt(0)" :: "v"(acc), "v"(acc), "v"(acc), "v"(acc), "v"(addr));
    if (MODE == 2) asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:1\n ds_read_u8 %2, %4 offset:2\n ds_read_u8 %3, %4 offset:3\n s_waitcnt lgkmcnt(0)" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(addr));
    if (MODE == 3) asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&t0), "=v"(*(uint64_t*)&t2) : "v"(addr));
    if (MODE != 1) acc += t0 + t1 + t2 + t3;
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE>
void run(const char *name, int stride, int off) {
  const int blocks = 256 * 16, iters = 4000;
  uint32_t *out; (void)hipMalloc(&out, blocks * 64 * 4);
  k<MODE><<<blocks, 64>>>(out, 10, stride, off); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, iters, stride, off); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  int per = MODE == 3 ? 2 : 4;
  printf("%-14s stride %2d off %d: %.2f ns per wave-instruction per CU (16 waves/CU)\n", name, stride, off, ms * 1e6 / (16.0 * iters * per));
  (void)hipFree(out);
}
int main() {
  run<0>("ds_read_b32", 20, 0); run<0>("ds_read_b32", 19, 0); run<0>("ds_read_b32", 19, 1); run<0>("ds_read_b32", 19, 2);
  run<1>("ds_write_b32", 20, 0); run<1>("ds_write_b32", 19, 0); run<1>("ds_write_b32", 19, 2);
  run<2>("ds_read_u8", 19, 0);
  run<3>("ds_read_b64", 24, 0); run<3>("ds_read_b64", 19, 0); run<3>("ds_read_b64", 20, 0);
  return 0;
}

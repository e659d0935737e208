// Microbenchmark: HBM write bandwidth of the store patterns a children-style kernel can use (MI355X).
//   hipcc -O3 --offload-arch=gfx950 write_patterns.hip -o write_patterns && ./write_patterns
// A  grid-stride: wave w writes 1 KB at ((it * G + w) * 1 KB)            (memset-like sliding window)
// B  one contiguous region per wave (G concurrent streams), 1 KB per instruction
// C  like B, but each half-wave writes its own 2 166-byte slot (512 B per half per instruction, 16-B aligned
//    vectors only, ragged edges as byte stores)                           (= what k_children2/3 do today)
// D  regions shared by K consecutive waves, chunk-interleaved: wave j of a group writes chunks j, j+K, ...
// E  like A, but at slot granularity: wave w writes slot pair (it * G + w) with the half-wave pattern of C
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((aligned(16))) V16 { uint32_t w[4]; };
constexpr int S = 2166, SLOTS = 362;
constexpr int64_t REGION = (int64_t)S * SLOTS;  // 784 092 B per parent

__global__ void kA(uint8_t *p, int64_t total) {
  const int64_t nchunk = total / 1024;
  const V16 z = {{0, 0, 0, 0}};
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) *reinterpret_cast<V16 *>(p + c * 1024 + threadIdx.x * 16) = z;
}
__global__ void kB(uint8_t *p, int64_t nregions) {
  const V16 z = {{0, 0, 0, 0}};
  const int64_t per = (REGION / 1024) * 1024;
  for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    uint8_t *q = p + r * per;
    for (int64_t o = 0; o < per; o += 1024) *reinterpret_cast<V16 *>(q + o + threadIdx.x * 16) = z;
  }
}
__device__ __forceinline__ void slot_zero(uint8_t *g, int hl) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  uint8_t *ga = g - mis;
  const int end = (int)mis + S;
  const int v0 = mis ? 1 : 0, v1 = end >> 4;
  const V16 z = {{0, 0, 0, 0}};
  for (int v = v0 + hl; v < v1; v += 32) *reinterpret_cast<V16 *>(ga + 16 * v) = z;
  const int head = mis ? 16 - (int)mis : 0, tail = end & 15;
  int j = -1;
  if (hl < 16) { if (hl < head) j = hl; }
  else if (hl - 16 < tail) j = S - tail + (hl - 16);
  if (j >= 0) g[j] = 0;
}
__global__ void kC(uint8_t *p, int64_t nregions) {
  const int h = threadIdx.x >> 5, hl = threadIdx.x & 31;
  for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    uint8_t *q = p + r * REGION;
    for (int a = 0; a < SLOTS; a += 2) slot_zero(q + (int64_t)(a + h) * S, hl);
  }
}
__global__ void kD(uint8_t *p, int64_t nregions, int K) {
  const V16 z = {{0, 0, 0, 0}};
  const int64_t per = (REGION / 1024) * 1024;
  const int64_t ngroups = (nregions + 0) ;
  for (int64_t i = blockIdx.x; i < ngroups * K; i += gridDim.x) {
    const int64_t r = i / K; const int j = (int)(i % K);
    uint8_t *q = p + r * per;
    for (int64_t o = (int64_t)j * 1024; o < per; o += (int64_t)K * 1024) *reinterpret_cast<V16 *>(q + o + threadIdx.x * 16) = z;
  }
}
__global__ void kE(uint8_t *p, int64_t npairs) {
  const int h = threadIdx.x >> 5, hl = threadIdx.x & 31;
  for (int64_t i = blockIdx.x; i < npairs; i += gridDim.x) slot_zero(p + (2 * i + h) * (int64_t)S, hl);
}
// F: like C but consecutive waves take consecutive slot pairs of the same parent group: K waves share a parent
__global__ void kF(uint8_t *p, int64_t nregions, int K) {
  const int h = threadIdx.x >> 5, hl = threadIdx.x & 31;
  for (int64_t i = blockIdx.x; i < nregions * K; i += gridDim.x) {
    const int64_t r = i / K; const int j = (int)(i % K);
    uint8_t *q = p + r * REGION;
    for (int a = 2 * j; a < SLOTS; a += 2 * K) slot_zero(q + (int64_t)(a + h) * S, hl);
  }
}
// G: B with non-temporal stores; H: B with 4 KB per wave iteration (4 stores in flight); I: 256-thread workgroups
__global__ void kG(uint8_t *p, int64_t nregions) {
  const int64_t per = (REGION / 1024) * 1024;
  for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    uint8_t *q = p + r * per;
    for (int64_t o = 0; o < per; o += 1024) {
      uint32_t *d = reinterpret_cast<uint32_t *>(q + o + threadIdx.x * 16);
      __builtin_nontemporal_store(0u, d); __builtin_nontemporal_store(0u, d + 1);
      __builtin_nontemporal_store(0u, d + 2); __builtin_nontemporal_store(0u, d + 3);
    }
  }
}
__global__ void kH(uint8_t *p, int64_t nregions) {
  const V16 z = {{0, 0, 0, 0}};
  const int64_t per = (REGION / 4096) * 4096;
  for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    uint8_t *q = p + r * per;
    for (int64_t o = 0; o < per; o += 4096) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<V16 *>(q + o + k * 1024 + threadIdx.x * 16) = z;
    }
  }
}
__global__ void kI(uint8_t *p, int64_t nregions) {   // 256 threads: 4 waves share a region, 4 KB per iteration
  const V16 z = {{0, 0, 0, 0}};
  const int64_t per = (REGION / 4096) * 4096;
  for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    uint8_t *q = p + r * per;
    for (int64_t o = 0; o < per; o += 4096) *reinterpret_cast<V16 *>(q + o + threadIdx.x * 16) = z;
  }
}
__global__ void kJ(uint8_t *p, int64_t nregions, int misalign) {   // B with the region start misaligned by `misalign` bytes (multiple of 16)
  const V16 z = {{0, 0, 0, 0}};
  const int64_t per = (REGION / 1024) * 1024;
  for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    uint8_t *q = p + r * per + misalign;
    for (int64_t o = 0; o < per - 1024; o += 1024) *reinterpret_cast<V16 *>(q + o + threadIdx.x * 16) = z;
  }
}
// K: NON-persistent: one workgroup of T threads per tile of T x 16 x U bytes, tiles in address order (what a fill kernel
// does: the dispatcher keeps a sliding window of tiles in flight); L: the same with single-wave workgroups
template <int U>
__global__ void kK(uint8_t *p, int64_t total) {
  const V16 z = {{0, 0, 0, 0}};
  const int64_t base = (int64_t)blockIdx.x * blockDim.x * 16 * U;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int64_t o = base + ((int64_t)k * blockDim.x + threadIdx.x) * 16;
    if (o + 16 <= total) *reinterpret_cast<V16 *>(p + o) = z;
  }
}
// M: persistent waves, but each wave takes the NEXT tile of 16 KB from a global atomic counter (dynamic sliding window)
__global__ void kM(uint8_t *p, int64_t total, unsigned long long *ctr) {
  const V16 z = {{0, 0, 0, 0}};
  const int64_t ntiles = total / 16384;
  for (;;) {
    unsigned long long t = 0;
    if (threadIdx.x == 0) t = atomicAdd(ctr, 1ull);
    t = __shfl(t, 0);
    if ((int64_t)t >= ntiles) break;
    uint8_t *q = p + t * 16384;
#pragma unroll
    for (int k = 0; k < 16; ++k) *reinterpret_cast<V16 *>(q + k * 1024 + threadIdx.x * 16) = z;
  }
}
// N: the 4 KB tiles of K, one per 256-thread workgroup, but WHICH tile a workgroup writes is remapped:
//   mode 0  tile = workgroup + shift             (shift 1..7: does it matter which XCD writes which tile? workgroups go
//                                                 round-robin over the 8 XCDs, memory is interleaved over the HBM stacks)
//   mode 1  tile = a pseudo-random permutation   (does the dense sliding window of in-flight stores matter?)
__global__ void kN(uint8_t *p, int64_t ntiles, int mode, int shift) {
  const V16 z = {{0, 0, 0, 0}};
  int64_t t = blockIdx.x;
  if (mode == 0) t = (t + shift) % ntiles;
  else t = (int64_t)(((unsigned long long)t * 0x9E3779B97F4A7C15ull) % (unsigned long long)ntiles);
  *reinterpret_cast<V16 *>(p + t * 4096 + threadIdx.x * 16) = z;
}
// O: PERSISTENT 256-thread workgroups that write 4 KB tiles grid-strided (tile = workgroup + k * grid): K's window, but
// the waves live on; P: the same, XCD-affine: a workgroup reads its XCC_ID and writes only tiles = xcc (mod 8)
__global__ void kO(uint8_t *p, int64_t ntiles) {
  const V16 z = {{0, 0, 0, 0}};
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) *reinterpret_cast<V16 *>(p + t * 4096 + threadIdx.x * 16) = z;
}
__global__ void kP(uint8_t *p, int64_t ntiles, unsigned long long *ctr, int shift) {
  const V16 z = {{0, 0, 0, 0}};
  unsigned int xc;
  asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xc));
  xc &= 7u;
  __shared__ unsigned long long slot;
  for (;;) {
    if (threadIdx.x == 0) slot = atomicAdd(ctr + xc * 16, 1ull);     // per-XCD tile counter (own cache line)
    __syncthreads();
    const int64_t t = (int64_t)slot * 8 + ((xc + shift) & 7);
    __syncthreads();
    if (t >= ntiles) break;
    *reinterpret_cast<V16 *>(p + t * 4096 + threadIdx.x * 16) = z;
  }
}
// Q: single-wave workgroups, one 1 KB tile each, non-persistent, in address order (K at wave granularity)
__global__ void kQ(uint8_t *p, int64_t total) {
  const V16 z = {{0, 0, 0, 0}};
  const int64_t o = (int64_t)blockIdx.x * 1024 + threadIdx.x * 16;
  if (o + 16 <= total) *reinterpret_cast<V16 *>(p + o) = z;
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  const int64_t P = 8192, total = P * REGION;
  uint8_t *p; hipMalloc(&p, total + 4096);
  {
    float t;
    unsigned long long *ctr; hipMalloc(&ctr, 8);
    t = timeit([&] { kK<1><<<(unsigned)((total + 4095) / 4096), 256>>>(p, total); }); printf("K tiles 4 KB, 256 thr      %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    t = timeit([&] { kK<4><<<(unsigned)((total + 16383) / 16384), 256>>>(p, total); }); printf("K tiles 16 KB, 256 thr     %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    t = timeit([&] { kK<16><<<(unsigned)((total + 65535) / 65536), 256>>>(p, total); }); printf("K tiles 64 KB, 256 thr     %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    t = timeit([&] { kK<4><<<(unsigned)((total + 4095) / 4096), 64>>>(p, total); }); printf("L tiles 4 KB, 64 thr       %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    t = timeit([&] { kK<16><<<(unsigned)((total + 16383) / 16384), 64>>>(p, total); }); printf("L tiles 16 KB, 64 thr      %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    t = timeit([&] { kK<64><<<(unsigned)((total + 65535) / 65536), 64>>>(p, total); }); printf("L tiles 64 KB, 64 thr      %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    if (false) for (int G : {3072, 8192, 16384}) {   // (measured: 1.3 TB/s - one global counter serialises the waves)
      t = timeit([&] { hipMemsetAsync(ctr, 0, 8, 0); kM<<<G, 64>>>(p, total, ctr); }); printf("M atomic 16 KB tiles G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    }
    {
      const int64_t nt = total / 4096;
      for (int sh : {0, 1, 2, 3, 4, 7}) { t = timeit([&] { kN<<<(unsigned)nt, 256>>>(p, nt, 0, sh); }); printf("N tiles 4 KB shifted by %d      %6.3f ms %6.2f TB/s\n", sh, t, nt * 4096.0 / t / 1e9); }
      t = timeit([&] { kN<<<(unsigned)nt, 256>>>(p, nt, 1, 0); }); printf("N tiles 4 KB permuted          %6.3f ms %6.2f TB/s\n", t, nt * 4096.0 / t / 1e9);
      for (int G : {2048, 4096, 8192}) { t = timeit([&] { kO<<<G, 256>>>(p, nt); }); printf("O persistent 4 KB tiles G=%5d %6.3f ms %6.2f TB/s\n", G, t, nt * 4096.0 / t / 1e9); }
      unsigned long long *c8; hipMalloc(&c8, 8 * 16 * 8);
      for (int sh : {0, 1, 3}) for (int G : {2048, 4096}) {
        t = timeit([&] { hipMemsetAsync(c8, 0, 8 * 16 * 8, 0); kP<<<G, 256>>>(p, nt, c8, sh); }); printf("P persistent XCD-affine sh=%d G=%5d %6.3f ms %6.2f TB/s\n", sh, G, t, nt * 4096.0 / t / 1e9);
      }
      t = timeit([&] { kQ<<<(unsigned)(total / 1024), 64>>>(p, total); }); printf("Q tiles 1 KB, 64 thr           %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    }
  }
  for (int G : {3072}) {
    float t;
    t = timeit([&] { hipMemsetAsync(p, 0, total, 0); }); printf("memset            %6.3f ms %6.2f TB/s\n", t, total / t / 1e9);
    t = timeit([&] { kA<<<G, 64>>>(p, total); }); printf("A grid-stride  G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    t = timeit([&] { kB<<<G, 64>>>(p, P); });     printf("B per-wave     G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    t = timeit([&] { kC<<<G, 64>>>(p, P); });     printf("C slots/wave   G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    for (int K : {2, 4, 8, 16}) { t = timeit([&] { kD<<<G, 64>>>(p, P, K); }); printf("D interleave K=%2d G=%5d %6.3f ms %6.2f TB/s\n", K, G, t, total / t / 1e9); }
    t = timeit([&] { kG<<<G, 64>>>(p, P); });     printf("G nontemporal  G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    t = timeit([&] { kH<<<G, 64>>>(p, P); });     printf("H 4KB/iter     G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    t = timeit([&] { kI<<<G / 4, 256>>>(p, P); }); printf("I 256 threads  G=%5d %6.3f ms %6.2f TB/s\n", G / 4, t, total / t / 1e9);
    for (int m : {16, 64, 128}) { t = timeit([&] { kJ<<<G, 64>>>(p, P, m); }); printf("J misalign %3d G=%5d %6.3f ms %6.2f TB/s\n", m, G, t, total / t / 1e9); }
    t = timeit([&] { kE<<<G, 64>>>(p, P * SLOTS / 2); }); printf("E slot-stride  G=%5d %6.3f ms %6.2f TB/s\n", G, t, total / t / 1e9);
    for (int K : {4, 16}) { t = timeit([&] { kF<<<G, 64>>>(p, P, K); }); printf("F slots K=%2d    G=%5d %6.3f ms %6.2f TB/s\n", K, G, t, total / t / 1e9); }
  }
  return 0;
}

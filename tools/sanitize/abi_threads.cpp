// abi_threads.cpp - the C-ABI of libgymgo_amd.so from two host threads at once, without Python or PyTorch in the process:
// the driver of the ThreadSanitizer pass (tools/sanitize.sh thread), whose runtime cannot host torch's GPU initialisation.
//
//   hipcc -O1 -g -std=c++17 -fsanitize=thread -I include tools/sanitize/abi_threads.cpp <tsan build of the library> -o abi_threads
//
// Each thread owns a stream and a batch of games and runs, from a common start barrier (cold caches: the first calls of
// the process race into the per-device CU cache and the mutex-guarded occupancy cache of gg_kernels.hip), a mix of entry
// points whose host side touches the library's mutable state: gg_batch_rollout (fused: FairShare board), gg_batch_invalid_mask,
// gg_batch_track_states, gg_batch_env_step (per-pair kernels: age split by occupancy), gg_batch_next_states.  The states each
// thread ends with must equal those of the same calls made alone.  Exit code 0 = identical and no sanitizer report.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "gymgo_amd.h"

#define CK(x) do { int e_ = (int)(x); if (e_ != 0) { fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, e_); return false; } } while (0)

static std::atomic<int> g_arrived{0};

static bool workload(uint64_t seed, int N, int64_t B, int rounds, bool wait_for_peer, std::vector<uint8_t> &result) {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const size_t S = (size_t)6 * N * N;
  uint8_t *st, *nxt, *mask, *dones;
  uint64_t *rng;
  int32_t *acts, *status, *taken;
  uint32_t *tracked;
  float *rewards;
  CK(hipMalloc(&st, B * S)); CK(hipMalloc(&nxt, B * S)); CK(hipMalloc(&mask, (size_t)B * N * N)); CK(hipMalloc(&dones, B));
  CK(hipMalloc(&rng, B * 8)); CK(hipMalloc(&acts, B * 4)); CK(hipMalloc(&status, B * 4)); CK(hipMalloc(&taken, B * 4));
  CK(hipMalloc(&tracked, (size_t)B * gg_tracked_words(N) * 4)); CK(hipMalloc(&rewards, B * 4));
  CK(hipMemsetAsync(st, 0, B * S, s));
  CK(gg_rng_seed(rng, seed, 0, B, s));
  if (wait_for_peer) {
    g_arrived.fetch_add(1);
    while (g_arrived.load() < 2) std::this_thread::yield();
  }
  for (int r = 0; r < rounds; ++r) {
    CK(gg_batch_rollout(st, rng, nullptr, nullptr, B, N, 24, 1, s));
    CK(gg_batch_invalid_mask(st, nullptr, mask, B, N, s));
    CK(gg_batch_sample_actions(st, rng, acts, B, N, s));
    CK(gg_batch_next_states(st, acts, nxt, status, B, N, 0, s));
    CK(gg_batch_track_states(nxt, tracked, B, N, s));
    CK(gg_batch_env_step(nxt, nullptr, rng, rewards, dones, status, taken, B, N, 7.5f, 0, 1, s));
    CK(hipMemcpyAsync(st, nxt, B * S, hipMemcpyDeviceToDevice, s));
  }
  result.resize(B * S + (size_t)B * N * N);
  CK(hipMemcpyAsync(result.data(), st, B * S, hipMemcpyDeviceToHost, s));
  CK(hipMemcpyAsync(result.data() + B * S, mask, (size_t)B * N * N, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  (void)hipFree(st); (void)hipFree(nxt); (void)hipFree(mask); (void)hipFree(dones); (void)hipFree(rng); (void)hipFree(acts);
  (void)hipFree(status); (void)hipFree(taken); (void)hipFree(tracked); (void)hipFree(rewards);
  (void)hipStreamDestroy(s);
  return true;
}

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { printf("abi_threads: no GPU visible - nothing to do\n"); return 0; }
  const int N = 19;
  const int64_t B0 = 40000, B1 = 20037;
  // the concurrent run FIRST (cold caches), the reference runs after it
  std::vector<uint8_t> got0, got1, want0, want1;
  bool ok0 = false, ok1 = false;
  std::thread t0([&] { ok0 = workload(101, N, B0, 3, true, got0); });
  std::thread t1([&] { ok1 = workload(202, N, B1, 3, true, got1); });
  t0.join();
  t1.join();
  if (!ok0 || !ok1) { printf("abi_threads: a call failed\n"); return 1; }
  if (!workload(101, N, B0, 3, false, want0) || !workload(202, N, B1, 3, false, want1)) return 1;
  const bool same = got0 == want0 && got1 == want1;
  size_t stones = 0;
  for (size_t i = 0; i < (size_t)N * N * 2; ++i) stones += want0[i];
  printf("abi_threads: two threads x two streams %s the sequential results (%zu + %zu bytes; first board holds %zu stones)\n",
         same ? "==" : "!=", want0.size(), want1.size(), stones);
  return same ? 0 : 1;
}

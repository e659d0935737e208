"""Where does a ONE-PLY byte-plane launch spend its time?  (-DGG_AB_PROF build: LIB=ab_libs/libgg_prof.so.)
For k_rollout2<R, PERPLY> and for k_rollout_lat forced onto the same launch: shader-clock cycles per wave by phase, the span of
the launch inside the machine (first wave in -> last wave out, 100 MHz wall clock) and the time per launch the host sees
(events around a hipGraph of 64 launches) - the difference is launch overhead nobody inside the kernel can shorten.
GGN / GGB / PLIES as the other scripts."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_libs/libgg_prof.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
for f in (L.gg_ab_prof_read_kernels, L.gg_ab_prof_read_lat):
    f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int32
NAMES = {'k_rollout2': ['draw', '-', '-', '-', 'ply (step_core2)', 'tables + bytes -> rows', '-', 'emit + stores'],
         'k_rollout_lat': ['draw', '-', 'tables + bytes -> rows', '-', 'ply (lat_play)', '-', 'first classes (11 floods)', 'emit + stores']}


def run(N, B, F, kernel):
    os.environ['GG_AB_LAT_MAX'] = '0' if kernel == 'k_rollout2' else str(1 << 30)
    os.environ['GG_AB_LAT_PLIES'] = '1'
    rd = L.gg_ab_prof_read_kernels if kernel == 'k_rollout2' else L.gg_ab_prof_read_lat
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    for _ in range(3): gogame.batch_rollout(st, rng, F, True)
    buf = (ctypes.c_ulonglong * 10)()
    rd(buf)
    reps, acc, spans = 32, [0] * 8, []
    for _ in range(reps):
        gogame.batch_rollout(st, rng, F, True)
        rd(buf)
        v = list(buf)
        for k in range(8): acc[k] += v[k]
        spans.append((v[9] - v[8]) * 0.01)
    waves = (B + 1) // 2 if kernel == 'k_rollout2' else (B + (4 if N <= 13 else 2) - 1) // (4 if N <= 13 else 2)
    # the host's view: a graph of 64 launches
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): gogame.batch_rollout(st, rng, F, True)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(64): gogame.batch_rollout(st, rng, F, True)
        g.replay(); s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(4): g.replay()
        b.record(s); s.synchronize()
    per = a.elapsed_time(b) / 256 * 1e3
    rd(buf)
    spans.sort()
    tot = sum(acc)
    print('%s N %d B %d F %d: %.2f us per launch in a graph (instrumented); span inside the machine: median %.2f us (min %.2f, max %.2f); %d waves'
          % (kernel, N, B, F, per, spans[len(spans) // 2], spans[0], spans[-1], waves))
    for n, x in zip(NAMES[kernel], acc):
        if x: print('    %-28s %5.1f %%  %8.0f cycles per wave' % (n, 100.0 * x / tot, x / (waves * reps)))
    print('    %-28s          %8.0f cycles per wave' % ('sum', tot / (waves * reps)), flush=True)


N, B = int(os.environ.get('GGN', 9)), int(os.environ.get('GGB', 4096))
for F in (1, 2):
    for kernel in ('k_rollout2', 'k_rollout_lat'):
        run(N, B, F, kernel)
# the floor: the smallest kernel of the library (gg_rng_seed, 16 workgroups) as a graph of 64 launches
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    rng = gogame.rng_seed(B, 1)
    s.synchronize()
    lib = _lib.lib()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(64): lib.gg_rng_seed(rng.data_ptr(), 1, 0, B, _lib.stream_ptr(rng.device))
    g.replay(); s.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(4): g.replay()
    b.record(s); s.synchronize()
print('gg_rng_seed (%d games, the smallest kernel): %.2f us per launch in a graph' % (B, a.elapsed_time(b) / 256 * 1e3))

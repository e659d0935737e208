"""Why does ONE launch shape (65 536 games of 19x19 x 256 plies, byte planes) read 1.85 ms in bench.py's batch sweep and 1.98 ms in
its timed region?  Candidates: (a) the clock - the sweep times 3 launches right after an idle gap (boost),
the timed region 20 launches after >= 0.35 s of back-to-back launches (sustained); (b) the amdsmi sampler thread polling inside
the timed region; (c) the positions - the sweep's batch has played ~500 - 1 100 plies, the timed batch ~50 000.
Every arm below is measured REPS times, interleaved, per-launch with HIP events:
    python tools/exp/timing_split.py            (prints a table; `gpurun_out/timing_split.txt` when OUT is set)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame

N, B, F = 19, 65536, 256
REPS = int(os.environ.get('REPS', '3'))
dev = torch.device('cuda:0')
out_lines = []


def say(s):
    print(s, flush=True)
    out_lines.append(s)


def batch(extra_launches):
    st = gogame.batch_init_state(B, N, device=dev)
    rng = gogame.rng_seed(B, 20260927, 0, dev)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * 40, True)
    for _ in range(extra_launches):
        gogame.batch_rollout(st, rng, F, True)
    torch.cuda.synchronize(dev)
    return st, rng


def launches(st, rng, n):
    """per-launch durations (ms) of n back-to-back launches"""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        gogame.batch_rollout(st, rng, F, True)
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


class Poller:
    def __init__(self, period=0.004):
        self.period, self.mhz = period, []
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                self.mhz.append(float(torch.cuda.clock_rate(dev)))
                torch.cuda.power_draw(dev)
            except Exception:
                return
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *e):
        self._stop.set()
        self._t.join(2.0)


def mean(x):
    return sum(x) / len(x)


def settle(st, rng, seconds):
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            gogame.batch_rollout(st, rng, F, True)
        torch.cuda.synchronize(dev)
        n += 4
    return n


say('timing_split: %d games of %dx%d x %d plies per launch, REPS %d' % (B, N, N, F, REPS))
fresh_st, fresh_rng = batch(1)          # what the batch sweep times: de-synchronised by 40 g plies + one launch
old_st, old_rng = batch(200)            # what the timed region times: + ~50 000 plies
say('arm                                              ' + '  '.join('rep%d' % i for i in range(REPS)) + '   (ms per launch)')
rows = {}


def arm(name, fn):
    rows.setdefault(name, []).append(fn())


for rep in range(REPS):
    # 1. sweep protocol: idle gap, one untimed launch, 3 timed - on both batches
    def sweep_like(st, rng):
        time.sleep(0.5)
        gogame.batch_rollout(st, rng, F, True); torch.cuda.synchronize(dev)
        return mean(launches(st, rng, 3))
    arm('idle 0.5 s -> 1 + 3 launches, ~50k-ply boards', lambda: sweep_like(old_st, old_rng))
    f2s, f2r = batch(1)
    arm('idle 0.5 s -> 1 + 3 launches, FRESH sweep boards', lambda: sweep_like(f2s, f2r))
    del f2s, f2r
    # 2. timed-region protocol: settle 0.35 s, 5 warm-up, 20 timed - sampler off / on
    def region(st, rng, settle_s, sampler):
        time.sleep(0.5)
        if settle_s:
            settle(st, rng, settle_s)
        if sampler:
            with Poller():
                for _ in range(5):
                    gogame.batch_rollout(st, rng, F, True)
                return mean(launches(st, rng, 20))
        for _ in range(5):
            gogame.batch_rollout(st, rng, F, True)
        return mean(launches(st, rng, 20))
    arm('settle 0.35 s + 5 + 20 launches, sampler OFF', lambda: region(old_st, old_rng, 0.35, False))
    arm('settle 0.35 s + 5 + 20 launches, sampler ON', lambda: region(old_st, old_rng, 0.35, True))
    arm('no settle, 5 + 20 launches, sampler OFF', lambda: region(old_st, old_rng, 0.0, False))
    arm('no settle, 5 + 20 launches, sampler ON', lambda: region(old_st, old_rng, 0.0, True))
    arm('settle 1.5 s + 5 + 20 launches, sampler OFF', lambda: region(old_st, old_rng, 1.5, False))
for k, v in rows.items():
    say('%-48s ' % k + '  '.join('%.4f' % x for x in v))

# 3. the curve: per-launch time against launch index after an idle gap (clock ramp / throttle), ~50k-ply boards
time.sleep(1.0)
cur = launches(old_st, old_rng, 400)
say('per-launch ms after 1 s idle, launches 0.. (mean of 10): ' + ' '.join('%.3f' % mean(cur[i:i + 10]) for i in range(0, 400, 10)))
with Poller(0.01) as p:
    time.sleep(1.0)
    n0 = len(p.mhz)
    cur = launches(old_st, old_rng, 300)
say('sclk MHz (amdsmi, 10 ms) idle: %s' % ' '.join('%d' % x for x in p.mhz[max(0, n0 - 5):n0]))
say('sclk MHz during 300 launches (every 5th sample): %s' % ' '.join('%d' % x for x in p.mhz[n0::5]))
say('per-launch ms with the 10 ms poller on (mean of 10): ' + ' '.join('%.3f' % mean(cur[i:i + 10]) for i in range(0, 300, 10)))
# 4. positions: the same curve on FRESH sweep boards (same protocol as 3: the difference at equal launch index is the positions)
del fresh_st, fresh_rng
fs, fr = batch(1)
say('  fresh boards: mean stones %.1f; ~50k-ply boards: %.1f' % (float(fs[:, :2].sum()) / B, float(old_st[:, :2].sum()) / B))
time.sleep(1.0)
cur = launches(fs, fr, 400)
say('FRESH boards, per-launch ms after 1 s idle (mean of 10): ' + ' '.join('%.3f' % mean(cur[i:i + 10]) for i in range(0, 400, 10)))
say('  after 400 launches: mean stones %.1f' % (float(fs[:, :2].sum()) / B))
if os.environ.get('OUT'):
    with open(os.environ['OUT'], 'w') as f:
        f.write('\n'.join(out_lines) + '\n')

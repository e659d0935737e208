"""small boards / small batches for a list of A/B libraries: fused rollout and one-ply launches
    python tools/exp/small_libs.py libgymgo_ab.so libgymgo_X.so ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] != 'run':
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, 'run', lib], capture_output=True, text=True)
            print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch, hashlib
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', sys.argv[2])
from gymgo_amd import gogame
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
out = []
for N, B in ((9, 4096), (9, 65536), (13, 65536)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    gogame.batch_rollout(st, rng, 300, True)
    ms = ev(lambda: gogame.batch_rollout(st, rng, 256, True), 12)
    m1 = ev(lambda: gogame.batch_rollout(st, rng, 1, True), 32)
    acts = gogame.batch_sample_actions(st, rng)
    mn = ev(lambda: gogame.batch_next_states(st, acts, check=False), 32)
    out.append('%dx%d B%d: fused %.3e  1-ply %.1f us  next_states %.1f us  %s' % (N, N, B, B * 256 / ms * 1e3, m1 * 1e3, mn * 1e3, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:8]))
print(' | '.join(out))

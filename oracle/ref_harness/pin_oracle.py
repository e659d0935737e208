"""Pin both oracles (C restatement, NumPy/SciPy restatement) against the REAL reference.

Runs only where /root/reference exists (the build container).  TEST INFRASTRUCTURE ONLY.
  python oracle/ref_harness/pin_oracle.py [--games-scale 1.0] [--speed]
Checks, bit-exact (reference float64 state cast to uint8):
  * gogame.next_state over seeded uniform-random full games on 3,5,7,9,13,19 (passes, captures,
    ko, game end, stepping past the end), canonical in {False, True};
  * every invalid action of sampled positions raises in the reference <=> oracle status 1;
  * state_utils.compute_invalid_moves directly (random ko points included);
  * gogame.areas, gogame.canonical_form, gogame.children(canonical in {F,T}, padded=True).
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refimport  # noqa: E402
from oracle import c_oracle, np_oracle  # noqa: E402


def u8(x):
    return np.asarray(x).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games-scale', type=float, default=1.0)
    ap.add_argument('--speed', action='store_true')
    ap.add_argument('--speed-reps', type=int, default=3)
    ap.add_argument('--speed-positions', type=int, default=2000)
    args = ap.parse_args()
    gym, gogame, govars, state_utils = refimport.load()
    rng = np.random.default_rng(20260927)
    plan = {3: 60, 5: 40, 7: 30, 9: 20, 13: 6, 19: 4}
    n_steps = n_inv = n_children = 0
    for size, games in plan.items():
        games = max(1, int(games * args.games_scale))
        for g in range(games):
            canonical = bool(g % 2)
            s = gogame.init_state(size)
            extra = 0
            for ply in range(4 * size * size):
                ended = gogame.game_ended(s)
                if ended:
                    extra += 1
                    if extra > 3:
                        break
                valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
                a = int(rng.choice(valid))
                if rng.random() < 0.08:
                    a = size * size
                ref = gogame.next_state(s, a, canonical)
                got_c = c_oracle.next_state(u8(s), a, canonical)
                got_np = np_oracle.next_state(s, a, canonical)
                assert np.array_equal(u8(ref), got_c), ('C oracle', size, g, ply, a)
                assert np.array_equal(u8(ref), u8(got_np)), ('np oracle', size, g, ply, a)
                n_steps += 1
                if ply % 7 == 0:
                    # invalid actions must be refused by both
                    for bad in np.flatnonzero(s[govars.INVD_CHNL].ravel() == 1)[:6]:
                        try:
                            gogame.next_state(s, int(bad))
                            raise SystemExit('reference accepted an invalid move?')
                        except AssertionError:
                            pass
                        out, st = c_oracle.batch_next_states(u8(s)[None], [int(bad)])
                        assert st[0] == 1 and np.array_equal(out[0], u8(s))
                        n_inv += 1
                    ba, wa = gogame.areas(s)
                    cb, cw = c_oracle.batch_areas(u8(s)[None])
                    nb, nw = np_oracle.areas(s)
                    assert (int(ba), int(wa)) == (int(cb[0]), int(cw[0])) == (int(nb), int(nw)), ('areas', size)
                    assert np.array_equal(u8(gogame.canonical_form(s)), c_oracle.canonical_form(u8(s)))
                    # compute_invalid_moves directly, with and without a ko point on an empty cell
                    player = int(rng.integers(0, 2))
                    empties = np.argwhere((s[0] + s[1]) == 0)
                    ko = tuple(empties[rng.integers(len(empties))]) if len(empties) and rng.random() < 0.5 else None
                    refm = state_utils.compute_invalid_moves(s, player, ko)
                    gotm = c_oracle.compute_invalid_moves(u8(s), player, -1 if ko is None else ko[0] * size + ko[1])
                    assert np.array_equal(u8(refm), gotm), ('compute_invalid_moves', size)
                    assert np.array_equal(refm, np_oracle.invalid_mask(s, player, ko))
                if ply % 23 == 0 and size <= 9 and not ended:  # reference children() raises on ended games
                    for canon in (False, True):
                        refc = gogame.children(s, canon, padded=True)
                        gotc = c_oracle.batch_children(u8(s)[None], canon)[0]
                        assert np.array_equal(u8(refc), gotc), ('children', size, canon)
                        n_children += 1
                s = gogame.next_state(s, a, False)
    # 19x19 children once mid-game
    s = gogame.init_state(19)
    for _ in range(150):
        valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)[:-1]
        s = gogame.next_state(s, int(rng.choice(valid)))
    assert np.array_equal(u8(gogame.children(s, True)), c_oracle.batch_children(u8(s)[None], True)[0])
    print('PINNED: %d steps, %d refused moves, %d children expansions bit-exact vs reference'
          % (n_steps, n_inv, n_children + 1))

    if args.speed:
        # Speed calibration of the NumPy/SciPy port against the real reference (19x19, same positions and actions):
        # `--speed-reps` interleaved repetitions (reference, port, reference, port ...) over `--speed-positions` positions
        # of uniform-random self-play.  Written to oracle/ref_harness/speed_calibration.json, which bench.py reads for
        # `cpu_baseline.port_vs_reference_speed` (the reference cannot travel to the GPU box, its ratio to the port can).
        acts, states = [], []
        s = gogame.init_state(19)
        for _ in range(max(2000, args.speed_positions)):
            if gogame.game_ended(s):
                s = gogame.init_state(19)
            valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
            a = int(rng.choice(valid))
            states.append(s)
            acts.append(a)
            s = gogame.next_state(s, a)
        ref_rates, port_rates, c_rates = [], [], []
        su = np.stack([u8(s) for s in states])
        for _ in range(max(3, args.speed_reps)):
            t0 = time.perf_counter()
            for s, a in zip(states, acts):
                gogame.next_state(s, a)
            ref_rates.append(len(acts) / (time.perf_counter() - t0))
            t0 = time.perf_counter()
            for s, a in zip(states, acts):
                np_oracle.next_state(s, a)
            port_rates.append(len(acts) / (time.perf_counter() - t0))
            t0 = time.perf_counter()
            c_oracle.batch_next_states(su, acts)
            c_rates.append(len(acts) / (time.perf_counter() - t0))
        ratios = [p / r for p, r in zip(port_rates, ref_rates)]
        import json
        import platform
        import scipy
        rec = {
            'what': 'gogame.next_state, 19x19, one core: the real reference vs oracle/np_oracle.py vs oracle/gg_oracle.c on the same '
                    'positions and actions (uniform-random self-play with restarts)',
            'positions': len(acts), 'repetitions': len(ratios),
            'reference_steps_per_s': {'mean': round(float(np.mean(ref_rates)), 1), 'min': round(min(ref_rates), 1), 'max': round(max(ref_rates), 1)},
            'port_steps_per_s': {'mean': round(float(np.mean(port_rates)), 1), 'min': round(min(port_rates), 1), 'max': round(max(port_rates), 1)},
            'c_oracle_steps_per_s': {'mean': round(float(np.mean(c_rates)), 1), 'min': round(min(c_rates), 1), 'max': round(max(c_rates), 1)},
            'port_vs_reference_speed': {'mean': round(float(np.mean(ratios)), 3), 'min': round(min(ratios), 3), 'max': round(max(ratios), 3)},
            'host': '%s, %d CPUs visible' % (platform.processor() or platform.machine(), os.cpu_count() or 0),
            'versions': {'python': platform.python_version(), 'numpy': np.__version__, 'scipy': scipy.__version__},
            'date': time.strftime('%Y-%m-%d'),
            'generated_by': 'python oracle/ref_harness/pin_oracle.py --speed --speed-reps %d --speed-positions %d' % (len(ratios), len(acts)),
        }
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'speed_calibration.json')
        with open(out, 'w') as f:
            json.dump(rec, f, indent=1)
            f.write('\n')
        print('speed 19x19 next_state, 1 core, %d positions x %d repetitions: reference %.0f steps/s (%.0f - %.0f), np port %.0f '
              '(%.2fx, %.2f - %.2f), C oracle %.0f  ->  %s'
              % (len(acts), len(ratios), np.mean(ref_rates), min(ref_rates), max(ref_rates), np.mean(port_rates),
                 np.mean(ratios), min(ratios), max(ratios), np.mean(c_rates), out))


if __name__ == '__main__':
    main()

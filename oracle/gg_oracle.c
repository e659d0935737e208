/*
 * gg_oracle.c - CPU restatement of huangeddie/GymGo's step hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and there only as the checker.  The product (gymgo_amd/, libgymgo_amd.so) never links, loads
 * or calls anything in oracle/.
 *
 * Parity pin: this restatement is checked in the build container against the real reference
 * imported from /root/reference (oracle/ref_harness/pin_oracle.py: randomized differential
 * over 3x3..19x19 full games + every scripted sequence of the reference's unit tests) and
 * against the golden vectors committed under tests/golden/ (generated from the reference by
 * tests/golden/make_golden.py).  Status: PINNED.
 *
 * It deliberately follows the reference's *algorithm* (label groups -> per-group liberty
 * sets -> possible_invalid / definite_valid / surrounded formula), not the closed-form
 * point-wise rule the HIP kernel uses, so kernel-vs-oracle parity is a genuine check.
 *
 * State layout (gym_go/gogame.py:7-19, gym_go/govars.py:4-11), stored here as uint8:
 *   [6][N][N]: 0 black, 1 white, 2 turn, 3 invalid moves, 4 previous-move-was-pass, 5 game over.
 * Citations are path:line relative to the reference root.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GG_MAXN 25
#define GG_MAXP (GG_MAXN * GG_MAXN)

enum { BLACK = 0, WHITE = 1, TURN_CHNL = 2, INVD_CHNL = 3, PASS_CHNL = 4, DONE_CHNL = 5, NUM_CHNLS = 6 };

static const int DR[4] = {-1, 1, 0, 0}; /* gym_go/state_utils.py:21 neighbor_deltas */
static const int DC[4] = {0, 0, -1, 1};

/* scipy.ndimage.label with the default 4-connectivity structure, as called at
 * gym_go/state_utils.py:48-49,166 and gym_go/gogame.py:283.  Labels are 1..n in raster order
 * of each component's first cell (the order only matters for iteration, never for results). */
static int label4(const uint8_t *plane, int N, int16_t *lab)
{
    int P = N * N, n = 0;
    int16_t stack[GG_MAXP];
    for (int i = 0; i < P; ++i) lab[i] = 0;
    for (int i = 0; i < P; ++i) {
        if (!plane[i] || lab[i]) continue;
        int sp = 0;
        ++n;
        lab[i] = (int16_t)n;
        stack[sp++] = (int16_t)i;
        while (sp) {
            int p = stack[--sp], r = p / N, c = p % N;
            for (int d = 0; d < 4; ++d) {
                int rr = r + DR[d], cc = c + DC[d];
                if (rr < 0 || rr >= N || cc < 0 || cc >= N) continue;
                int q = rr * N + cc;
                if (plane[q] && !lab[q]) {
                    lab[q] = (int16_t)n;
                    stack[sp++] = (int16_t)q;
                }
            }
        }
    }
    return n;
}

/* liberties of group `g` of labelling `lab`: empties * binary_dilation(group)
 * (gym_go/state_utils.py:61-62 with surround_struct; :173 with the default cross - the centre
 * never matters because a stone is not empty).  Writes the 0/1 liberty map, returns its sum. */
static int group_liberties(const int16_t *lab, int g, const uint8_t *empties, int N, uint8_t *libmap)
{
    int P = N * N, cnt = 0;
    memset(libmap, 0, (size_t)P);
    for (int p = 0; p < P; ++p) {
        if (lab[p] != g) continue;
        int r = p / N, c = p % N;
        for (int d = 0; d < 4; ++d) {
            int rr = r + DR[d], cc = c + DC[d];
            if (rr < 0 || rr >= N || cc < 0 || cc >= N) continue;
            int q = rr * N + cc;
            if (empties[q] && !libmap[q]) {
                libmap[q] = 1;
                ++cnt;
            }
        }
    }
    return cnt;
}

/* gym_go/state_utils.py:24-83 compute_invalid_moves(state, player, ko_protect).
 * `ko` is a flat index or -1.  Writes the N*N 0/1 mask (invalid for the OPPONENT of `player`). */
void gg_oracle_compute_invalid_moves(const uint8_t *state, int32_t N, int32_t player, int32_t ko, uint8_t *mask)
{
    int P = N * N;
    const uint8_t *own = state + (size_t)player * P, *opp = state + (size_t)(1 - player) * P;
    uint8_t all_pieces[GG_MAXP], empties[GG_MAXP], libmap[GG_MAXP];
    int possible_invalid[GG_MAXP], definite_valid[GG_MAXP];
    int16_t lab[GG_MAXP];
    for (int p = 0; p < P; ++p) { /* :39-40 */
        all_pieces[p] = (uint8_t)(state[p] + state[P + p]);
        empties[p] = (uint8_t)(1 - all_pieces[p]);
        possible_invalid[p] = 0; /* :43-44 */
        definite_valid[p] = 0;
    }
    /* own groups (:48, :61, :64, :70, :73): >1 liberty -> possible invalid, ==1 -> definite valid */
    int n = label4(own, N, lab);
    for (int g = 1; g <= n; ++g) {
        int cnt = group_liberties(lab, g, empties, N, libmap);
        if (cnt > 1)
            for (int p = 0; p < P; ++p) possible_invalid[p] += libmap[p];
        if (cnt == 1)
            for (int p = 0; p < P; ++p) definite_valid[p] += libmap[p];
    }
    /* opponent groups (:49, :62, :65, :71, :74): ==1 liberty -> possible invalid, >1 -> definite valid */
    n = label4(opp, N, lab);
    for (int g = 1; g <= n; ++g) {
        int cnt = group_liberties(lab, g, empties, N, libmap);
        if (cnt == 1)
            for (int p = 0; p < P; ++p) possible_invalid[p] += libmap[p];
        if (cnt > 1)
            for (int p = 0; p < P; ++p) definite_valid[p] += libmap[p];
    }
    /* :77 surrounded = convolve(all_pieces, surround_struct, mode='constant', cval=1) == 4 ; :78 */
    for (int p = 0; p < P; ++p) {
        int r = p / N, c = p % N, s = 0;
        for (int d = 0; d < 4; ++d) {
            int rr = r + DR[d], cc = c + DC[d];
            if (rr < 0 || rr >= N || cc < 0 || cc >= N) s += 1;
            else s += all_pieces[rr * N + cc];
        }
        int inv = all_pieces[p] + possible_invalid[p] * (definite_valid[p] == 0) * (s == 4);
        mask[p] = (uint8_t)(inv > 0);
    }
    if (ko >= 0) mask[ko] = 1; /* :81-82 */
}

/* gym_go/state_utils.py:214-223 adj_data: on-board neighbours of `a`; surrounded = all hold an
 * opponent stone. */
static int adj_data(const uint8_t *state, int N, int a, int player, int *nbr, int *surrounded)
{
    int P = N * N, r = a / N, c = a % N, k = 0;
    *surrounded = 1;
    for (int d = 0; d < 4; ++d) {
        int rr = r + DR[d], cc = c + DC[d];
        if (rr < 0 || rr >= N || cc < 0 || cc >= N) continue;
        nbr[k++] = rr * N + cc;
        if (!state[(size_t)(1 - player) * P + rr * N + cc]) *surrounded = 0;
    }
    return k;
}

/* gym_go/state_utils.py:159-180 update_pieces: remove the opponent groups adjacent to the move
 * that have no liberty.  Returns the number of killed groups; *single = flat index of the stone
 * when exactly one group of exactly one stone died, else -1. */
static int update_pieces_n(uint8_t *state, int N, const int *nbr, int k, int player, int *single)
{
    int P = N * N, killed = 0, killed_stones = 0, last = -1;
    uint8_t *oppp = state + (size_t)(1 - player) * P;
    uint8_t empties[GG_MAXP], libmap[GG_MAXP];
    int16_t lab[GG_MAXP];
    int seen[64], ns = 0;
    for (int p = 0; p < P; ++p) empties[p] = (uint8_t)(1 - (state[p] + state[P + p])); /* :163-164 */
    label4(oppp, N, lab);                                                              /* :166 */
    for (int i = 0; i < k; ++i) {                                                      /* :169-171 */
        int g = lab[nbr[i]], dup = 0;
        if (!g) continue;
        for (int j = 0; j < ns; ++j) dup |= (seen[j] == g);
        if (dup) continue;
        seen[ns++] = g;
        if (group_liberties(lab, g, empties, N, libmap) <= 0) { /* :172-174 */
            int sz = 0;
            for (int p = 0; p < P; ++p)
                if (lab[p] == g) { oppp[p] = 0; ++sz; last = p; } /* :176-178 */
            ++killed;
            killed_stones += sz;
        }
    }
    *single = (killed == 1 && killed_stones == 1) ? last : -1;
    return killed;
}

/* The same as a stand-alone entry (state_utils.update_pieces as the reference exposes it: any state, any list of
 * locations): planes 0/1 of `state` are edited in place, killed[p] = 1 for every removed stone (nullable).
 * adj: k flat indices, entries outside [0, N*N) are skipped.  Returns the number of killed groups. */
int32_t gg_oracle_update_pieces(uint8_t *state, int32_t N, const int32_t *adj, int32_t k, int32_t player, uint8_t *killed)
{
    int P = N * N, n = 0, single, nbr[64];
    uint8_t before[GG_MAXP];
    const uint8_t *oppp = state + (size_t)(1 - player) * P;
    for (int i = 0; i < k && n < 64; ++i)
        if (adj[i] >= 0 && adj[i] < P) nbr[n++] = adj[i];
    memcpy(before, oppp, (size_t)P);
    int groups = update_pieces_n(state, N, nbr, n, player, &single);
    if (killed)
        for (int p = 0; p < P; ++p) killed[p] = (uint8_t)(before[p] && !oppp[p]);
    return groups;
}

static int update_pieces(uint8_t *state, int N, const int *nbr, int k, int player, int *single)
{
    return update_pieces_n(state, N, nbr, k, player, single);
}

static int plane_max(const uint8_t *pl, int P)
{
    int m = 0;
    for (int p = 0; p < P; ++p) if (pl[p] > m) m = pl[p];
    return m;
}

/* gym_go/gogame.py:313-321 canonical_form (in place on a private copy). */
void gg_oracle_canonical_form(uint8_t *state, int32_t N)
{
    int P = N * N;
    if (plane_max(state + (size_t)TURN_CHNL * P, P) == WHITE) {
        for (int p = 0; p < P; ++p) {
            uint8_t t = state[p];
            state[p] = state[P + p];
            state[P + p] = t;
            state[(size_t)TURN_CHNL * P + p] = (uint8_t)(1 - state[(size_t)TURN_CHNL * P + p]);
        }
    }
}

/* gym_go/gogame.py:34-87 next_state.  Returns 0, or 1 when the reference would raise
 * AssertionError (:59 invalid move) - then `out` is an unchanged copy of `in`.
 * Out-of-range actions also return 1 (the reference would raise IndexError). */
int32_t gg_oracle_next_state(const uint8_t *in, int32_t action, uint8_t *out, int32_t N, int32_t canonical)
{
    int P = N * N, ko = -1;
    memcpy(out, in, (size_t)NUM_CHNLS * P);                           /* :36 */
    int player = plane_max(in + (size_t)TURN_CHNL * P, P);            /* :44, :241-246 */
    int prev_passed = 0;                                              /* :45, :200-201 */
    for (int p = 0; p < P; ++p) prev_passed |= (in[(size_t)PASS_CHNL * P + p] == 1);
    if (action < 0 || action > P) return 1;
    if (action == P) {                                                /* :48-53 */
        memset(out + (size_t)PASS_CHNL * P, 1, (size_t)P);
        if (prev_passed) memset(out + (size_t)DONE_CHNL * P, 1, (size_t)P);
    } else {
        if (in[(size_t)INVD_CHNL * P + action] != 0) return 1;        /* :59 */
        memset(out + (size_t)PASS_CHNL * P, 0, (size_t)P);            /* :56 */
        out[(size_t)player * P + action] = 1;                         /* :62 */
        int nbr[4], surrounded, single;
        int k = adj_data(out, N, action, player, nbr, &surrounded);   /* :65 */
        update_pieces(out, N, nbr, k, player, &single);               /* :68 */
        if (single >= 0 && surrounded) ko = single;                   /* :72-75 */
    }
    gg_oracle_compute_invalid_moves(out, N, player, ko, out + (size_t)INVD_CHNL * P); /* :78 */
    for (int p = 0; p < P; ++p)                                       /* :81, state_utils.py:235-241 */
        out[(size_t)TURN_CHNL * P + p] = (uint8_t)(1 - out[(size_t)TURN_CHNL * P + p]);
    if (canonical) gg_oracle_canonical_form(out, N);                  /* :83-85 */
    return 0;
}

/* Stacked gogame.next_state - the semantics gogame.batch_next_states (gym_go/gogame.py:90-150)
 * has whenever no game of the batch passes (with passes the reference's
 * batch_update_pieces mis-aligns games, gym_go/state_utils.py:187-193; not reproduced). */
void gg_oracle_batch_next_states(const uint8_t *in, const int32_t *actions, uint8_t *out, int32_t *status,
                                 int64_t B, int32_t N, int32_t canonical)
{
    size_t S = (size_t)NUM_CHNLS * N * N;
    for (int64_t b = 0; b < B; ++b) {
        int32_t st = gg_oracle_next_state(in + b * S, actions[b], out + b * S, N, canonical);
        if (status) status[b] = st;
    }
}

/* gym_go/gogame.py:275-300 areas (Tromp-Taylor): stones + empty regions bordered by one colour only. */
void gg_oracle_areas(const uint8_t *state, int32_t N, int32_t *black_area, int32_t *white_area)
{
    int P = N * N, b = 0, w = 0;
    uint8_t empties[GG_MAXP];
    int16_t lab[GG_MAXP];
    for (int p = 0; p < P; ++p) {
        empties[p] = (uint8_t)(1 - (state[p] + state[P + p])); /* :280-281 */
        b += state[p];                                         /* :285 */
        w += state[P + p];
    }
    int n = label4(empties, N, lab); /* :283 */
    for (int g = 1; g <= n; ++g) {   /* :286-298 */
        int bc = 0, wc = 0, sz = 0;
        for (int p = 0; p < P; ++p) {
            if (lab[p] != g) continue;
            ++sz;
            int r = p / N, c = p % N;
            for (int d = 0; d < 4; ++d) {
                int rr = r + DR[d], cc = c + DC[d];
                if (rr < 0 || rr >= N || cc < 0 || cc >= N) continue;
                bc |= state[rr * N + cc];
                wc |= state[P + rr * N + cc];
            }
        }
        if (bc && !wc) b += sz;
        else if (wc && !bc) w += sz;
    }
    *black_area = b;
    *white_area = w;
}

void gg_oracle_batch_areas(const uint8_t *states, int32_t *black, int32_t *white, int64_t B, int32_t N)
{
    size_t S = (size_t)NUM_CHNLS * N * N;
    for (int64_t b = 0; b < B; ++b) gg_oracle_areas(states + b * S, N, black + b, white + b);
}

/* gym_go/gogame.py:153-157 invalid_moves: plane 3 flattened + [0] for pass; all zeros once ended. */
static int game_ended(const uint8_t *state, int N) /* :208-214 */
{
    int P = N * N, cnt = 0;
    for (int p = 0; p < P; ++p) cnt += (state[(size_t)DONE_CHNL * P + p] == 1);
    return cnt == P;
}

/* gym_go/gogame.py:175-186 children(state, canonical, padded=True) -> [N*N+1][6][N][N], all-zero
 * slots for invalid actions. */
void gg_oracle_children(const uint8_t *state, uint8_t *children, int32_t N, int32_t canonical)
{
    int P = N * N;
    size_t S = (size_t)NUM_CHNLS * P;
    int ended = game_ended(state, N);
    memset(children, 0, S * (size_t)(P + 1));
    for (int a = 0; a <= P; ++a) {
        int invalid = (a < P && !ended) ? state[(size_t)INVD_CHNL * P + a] : 0; /* :153-161 */
        if (invalid) continue;
        int32_t st = gg_oracle_next_state(state, a, children + (size_t)a * S, N, canonical);
        if (st) memset(children + (size_t)a * S, 0, S); /* the reference would have raised */
    }
}

void gg_oracle_batch_children(const uint8_t *states, uint8_t *children, int64_t B, int32_t N, int32_t canonical)
{
    size_t S = (size_t)NUM_CHNLS * N * N;
    for (int64_t b = 0; b < B; ++b)
        gg_oracle_children(states + b * S, children + (size_t)b * S * (size_t)(N * N + 1), N, canonical);
}

/* ---- uniform-random rollout policy (GoEnv.uniform_random_action, gym_go/envs/go_env.py:78-81:
 * uniform over the valid actions including pass).  The reference draws from NumPy's global
 * generator; the build defines a counter-based generator instead so that the device and this
 * oracle pick identical actions: per game, x += 0x9E3779B97F4A7C15; u = splitmix64 finaliser(x);
 * action = the floor((u >> 32) * n_valid / 2^32)-th valid action in ascending index order. ---- */
static uint64_t splitmix_next(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t gg_oracle_rng_seed(uint64_t base_seed, uint64_t game_index)
{
    uint64_t x = base_seed ^ (game_index * 0xD1342543DE82EF95ull);
    splitmix_next(&x);
    return x;
}

/* One rollout ply for one game: a finished game is reset to zeros first when auto_reset (build-side
 * policy) or left frozen (returns -1, no RNG draw) otherwise; then draw a uniform valid
 * action and step.  Returns the action taken. */
int32_t gg_oracle_rollout_ply(uint8_t *state, uint64_t *rng, int32_t N, int32_t auto_reset)
{
    int P = N * N;
    size_t S = (size_t)NUM_CHNLS * P;
    uint8_t tmp[NUM_CHNLS * GG_MAXP];
    if (game_ended(state, N)) {
        if (!auto_reset) return -1;
        memset(state, 0, S);
    }
    int nvalid = 1;
    for (int p = 0; p < P; ++p) nvalid += (state[(size_t)INVD_CHNL * P + p] == 0);
    uint64_t u = splitmix_next(rng);
    uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)nvalid) >> 32);
    int32_t action = P;
    for (int p = 0; p < P; ++p) {
        if (state[(size_t)INVD_CHNL * P + p] == 0) {
            if (k == 0) { action = p; break; }
            --k;
        }
    }
    gg_oracle_next_state(state, action, tmp, N, 0);
    memcpy(state, tmp, S);
    return action;
}

void gg_oracle_batch_rollout(uint8_t *states, uint64_t *rng, int32_t *last_actions, int64_t B, int32_t N,
                             int32_t plies, int32_t auto_reset)
{
    size_t S = (size_t)NUM_CHNLS * N * N;
    for (int64_t b = 0; b < B; ++b) {
        int32_t last = -1; /* the last action applied in this call; -1 if the game was frozen throughout */
        for (int t = 0; t < plies; ++t) {
            int32_t a = gg_oracle_rollout_ply(states + b * S, rng + b, N, auto_reset);
            if (a >= 0) last = a;
        }
        if (last_actions) last_actions[b] = last;
    }
}

/* ---- policy-weighted action sampling: gogame.random_weighted_action (gym_go/gogame.py:385-392: L1-normalise the move
 * weights, draw from them) and gogame.random_action (:395-404: the same with weights 1 - invalid_moves).  The reference
 * normalises in float64 and draws from NumPy's global generator; the build defines an EXACT integer form so that the
 * device and this oracle pick identical actions from identical weights:
 *   1. v[a] = the float32 weight clamped to [+0, FLT_MAX] on its bit pattern (negative -> 0, NaN / +inf -> FLT_MAX),
 *      and 0 where plane 3 marks the point invalid (a < N*N; "assumes all invalid moves have weight 0", :387 - here
 *      enforced); the pass (a = N*N) is never masked;
 *   2. E = max(biased exponent of max v, 24), S = 2^(148 - E) (bits (275 - E) << 23), q[a] = trunc(v[a] * S) < 2^22:
 *      an exact power-of-two scaling, i.e. the weights as 22-bit fixed point relative to the largest one
 *      (weights below 2^-22 of the largest, or below 2^-103, count as 0);
 *   3. T = sum q (< 2^31); one draw of the game's generator: k = floor((u >> 32) * T / 2^32), uniform on [0, T);
 *   4. inverse CDF in the INTERLEAVED action order a = i + 16 j (i = 0..15 outer, j inner - one 16-lane DPP row per
 *      board on the device): the first a of that order whose running sum of q exceeds k.  Any fixed order gives
 *      P(a) = q[a] / T.
 * T == 0 (no positive weight on a valid action) returns -1: np.random.choice raises for such a vector.
 * The generator advances exactly once per call. ---- */
int32_t gg_oracle_sample_weighted(const uint8_t *invalid, const float *w, uint64_t *rng, int32_t N)
{
    int P = N * N, A = P + 1;
    uint32_t v[GG_MAXP + 1], q[GG_MAXP + 1], mx = 0;
    for (int a = 0; a < A; ++a) {
        int32_t bits;
        memcpy(&bits, &w[a], 4);
        if (bits < 0) bits = 0;
        if (bits > 0x7F7FFFFF) bits = 0x7F7FFFFF;
        if (a < P && invalid && invalid[a]) bits = 0;
        v[a] = (uint32_t)bits;
        if (v[a] > mx) mx = v[a];
    }
    uint32_t E = mx >> 23;
    if (E < 24) E = 24;
    uint32_t sbits = (275u - E) << 23;
    float S;
    memcpy(&S, &sbits, 4);
    uint32_t T = 0;
    for (int a = 0; a < A; ++a) {
        float f;
        memcpy(&f, &v[a], 4);
        volatile float prod = f * S; /* one IEEE single-precision product (exact: S is a power of two) */
        q[a] = (uint32_t)prod;
        T += q[a];
    }
    uint64_t u = splitmix_next(rng);
    if (T == 0) return -1;
    uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)T) >> 32);
    uint32_t run = 0;
    for (int i = 0; i < 16; ++i)
        for (int a = i; a < A; a += 16) {
            run += q[a];
            if (run > k) return a;
        }
    return -1; /* unreachable: run reaches T > k */
}

/* actions[b] = gg_oracle_sample_weighted(plane 3 of states[b], weights[b], &rng[b]); weights float32 [B][N*N+1].
 * A finished game masks nothing (gogame.invalid_moves returns zeros once the game has ended, gym_go/gogame.py:155-156). */
void gg_oracle_batch_sample_weighted(const uint8_t *states, const float *weights, uint64_t *rng, int32_t *actions,
                                     int64_t B, int32_t N)
{
    size_t P = (size_t)N * N, S = (size_t)NUM_CHNLS * P;
    for (int64_t b = 0; b < B; ++b)
        actions[b] = gg_oracle_sample_weighted((states && !game_ended(states + b * S, N)) ? states + b * S + (size_t)INVD_CHNL * P : NULL,
                                               weights + b * (P + 1), rng + b, N);
}

int32_t gg_oracle_version(void) { return 2; }

// xq_engine.hip — tree + rules kernels and the C ABI of the batched self-play engine (gfx950).
//
// Execution model: grid = G workgroups of ONE wavefront (64 lanes); workgroup g owns game g for
// the whole launch, so every tree update is wave-private (ordered fp64 adds, no atomics) and the
// 90-byte board plus move buffers live in LDS.  State between launches stays in HBM as SoA:
// nibble-packed boards (48 B), 16-byte scalar records, a flat per-game node arena
// (N:u32, W:f64, P:f32, move:u16, first_child:u16, n_child:u8, flags:u8) that is reset every ply
// because the reference never reuses its tree (self_play.py:98).
//
// Kernels (all wave-per-game):
//   k_new_games     reset + first root move list                         chess_env.py:14-67
//   k_set_roots     caller-provided root states (MCTS.search on an env)  self_play.py:156-175
//   k_search_round  [consume evaluator output of round r-1] + the sims of round r on the frozen
//                   tree, terminal leaves backed up in place, <= 1 pending leaf per game written
//                   with its network planes                              self_play.py:103-148
//   k_hashnet       exact dyadic evaluator for parity tests              SURVEY.md Appendix B
//   k_end_search    consume the last round                               self_play.py:146-154
//   k_play_move     visits -> pi -> np.random.choice -> make_move        self_play.py:219-256
//   k_finalize      z table                                              self_play.py:259-310
//   k_rules_*       batch entry points on caller boards (host mirror of ChineseChess)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (float results are part of the
// parity contract: PUCT is float32 stepwise, W is float64, rewards/z are float64).
#include "xq_device.hpp"
#include "../../include/xq_selfplay.h"
#include "../../include/xq_debug.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

using namespace xq;

// ------------------------------------------------------------------------------------------
// device-side data structures
// ------------------------------------------------------------------------------------------
struct __align__(16) GameS {
    int8_t  side;
    uint8_t move_count;
    int8_t  winner;
    uint8_t reason;
    int8_t  reason_side;
    uint8_t reason_count;
    int8_t  rk, bk;
    uint8_t nocap, cchk, n_hist, n_root;
    uint8_t done, n_samples, error, n_plies;
};
static_assert(sizeof(GameS) == 16, "GameS must be 16 bytes");
static_assert(sizeof(xq_sample_record) == 576, "xq_sample_record layout is part of the ABI");

enum : int { F_TERM = 2, F_VAL_NEG = 4, F_VAL_POS = 8, F_CHECK = 16 };
enum : int { PATH_CAP = 72, LEAF_NONE = 0xFFFF };

struct Eng {
    // (the kernel argument is ~0.5 KB, more than a wave's scalar registers: the fields are ordered by the phase of
    // k_search_round that reads them, so that each phase's pointers arrive in one or two wide scalar loads)
    // ---- phase A of a search round: everything addressed by the game alone
    GameS    *gs;             // [G]
    uint32_t *board;          // [G][12]
    uint16_t *leaf_node; uint8_t *leaf_n; uint8_t *leaf_mult; uint8_t *leaf_depth;      // pending leaf per slot
    uint16_t *leaf_moves;     // [slots][128]
    uint16_t *leaf_path;      // [slots][PATH_CAP]
    uint32_t *n_nodes;        // [G]
    // evaluator row compaction (xq_engine_set_row_compaction): only slots with a pending leaf become network rows.
    // k_assign_rows numbers them in slot order after every search round: leaf_row[slot] = row or -1,
    // row_src[row] = slot (where the search kernel wrote the planes), row_count[0] = rows of this round
    int32_t *leaf_row, *row_src, *row_count, *row_hist;
    // tree root per game: always node 0 (fresh tree every ply, self_play.py:98) unless the opt-in
    // tree reuse keeps the played child's subtree (extension)
    uint16_t *root_node;
    // opt-in: carry the played child's network evaluation over as the next root's (xq_engine_set_root_eval_carry)
    uint8_t *root_ready; int32_t *roots_not_ready;
    int G, ncap, compact, tree_reuse, eval_carry, dedupe, dd_mask;
    unsigned dd_tag;          // tag of the round being launched (set by the host before every k_search_round)
    // ---- phase B: node arena, [G][ncap] each; the compact policy layout
    uint32_t *nN; double *nW; float *nP; uint16_t *nMove; uint16_t *nFirst; uint8_t *nNc; uint8_t *nFlags;
    // optional compact policy layout: logits rows hold only n_cols columns, col_map[move] = column
    const int16_t *col_map; int n_cols;
    int sims, leaf_batch, nrounds, max_moves, opponent_mode, want_check;
    // ---- the rest of a round: leaf record, dedupe table
    uint32_t *leaf_board;     // [slots][12]
    int8_t   *leaf_side;      // [slots]
    uint16_t *root_moves;     // [G][128]
    // leaf dedupe (xq_engine_set_leaf_dedupe, needs compaction): slots whose pending leaves are the same position
    // (board + side to move) share one network row.  dd_tab = open-addressing table of (round tag << 32 | lowest
    // slot of the position), dd_mask + 1 entries; leaf_pos[slot] = the entry dedupe_insert found for the slot
    unsigned long long *dd_tab;
    int32_t *leaf_pos;
    uint8_t *leaf_dup;        // [slots] 1 = a lower slot holds the same position (cleared by k_assign_rows after reading)
    float    *priors;         // [slots][128]
    double   *values;         // [slots]
    // evaluation cache (xq_engine_set_eval_cache): the evaluator's answer - legal-move priors and value - for a position
    // (board + side to move = everything the network reads, neural_network.py:128-146) is kept for two plies and handed to
    // any later leaf that is the same position: no network row.  ec_state[i] = epoch << 32 | launch that filled the entry
    // (0xffffffff: reserved, being filled); ec_key[i] = 12 board dwords, side + 2, n, value bits, 0; ec_prior[i][128].
    // leaf_ec[slot]: -1 nothing, >= 0 the entry this pending leaf reads, <= -2 the entry -(i + 2) it fills when consumed
    int ec_on, ec_mask;
    unsigned ec_epoch, ec_seq;
    unsigned long long *ec_state;
    uint32_t *ec_key;
    float    *ec_prior;
    int32_t  *leaf_ec;
    uint32_t *ec_stats;               // [3][G * 64] per slot: hits, fills, mismatches found in verify mode (ec_on == 2)
    int ec_stat_stride;
    // ---- per ply
    double temperature;
    uint64_t *pos_hist;       // [G][PATH_CAP]
    uint8_t  *chk_hist;       // [G][PATH_CAP]
    double   *uniforms;       // [G][70]
    // samples, [G][70]
    uint32_t *s_board; int8_t *s_player; uint8_t *s_n; uint16_t *s_moves; uint16_t *s_counts; double *s_z;
    double   *step_reward; uint16_t *t_move;
    const double *pow_table; int pow_n;
    // opt-in search extensions with no counterpart in the reference (BASELINE C5);
    // noise_eps == 0 keeps the reference behaviour
    double noise_alpha, noise_eps;
    uint64_t noise_seed;
    // virtual loss (opt-in extension): K = leaf_slots pending leaves per game and round instead of one
    // (every leaf_* / priors / values array and the evaluator's rows are indexed by slot = g * K + k);
    // nVl counts the pending visits through a node and is folded into PUCT as N + vl, W - vl
    int vloss, leaf_slots;
    uint8_t *nVl;
};

__device__ __forceinline__ void eval_cache_fill(const Eng &E, int i, uint32_t board_dword, int side, int n, float p0, float p1, float value);
// (accessors of the evaluation cache: past the L2s - see eval_cache_fill)
__device__ __forceinline__ uint32_t ec_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ec_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ec_ld64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ec_st64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ec_ldf(const float *p) { return __uint_as_float(ec_ld(reinterpret_cast<const uint32_t *>(p))); }
// key and priors of an entry that was FILLED IN AN EARLIER LAUNCH (state word: filled < this launch's number): ordinary
// cached loads - a kernel boundary lies between the stores and these reads, and nobody refills an entry that is still
// valid - so a position many games hold at once is served by the L2s, not by one memory channel.  (Measured: no
// difference on BASELINE C3, where hits are spread; verify mode over whole games: 0 mismatches in 46 M leaves.)
__device__ __forceinline__ uint32_t ec_old(const uint32_t *p) { return *p; }
__device__ __forceinline__ float ec_oldf(const float *p) { return *p; }
// a 64-bit value of the first active lane, wave-uniform.  (__builtin_amdgcn_readfirstlane returns int: without the casts the
// low half is SIGN-extended into the high one - a state word that ends in 0xffffffff then reads as all ones.)
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

struct __align__(16) WaveLds {
    int8_t   root_bd[96];
    int8_t   bd[96];
    uint16_t cand[128];
    uint16_t legal[128];
    uint8_t  own_sq[32];
    AttackMaps maps;         // per-row / per-column piece masks of the position being generated (xq_attack.hpp)
    uint16_t path_node[PATH_CAP];
    uint16_t path_move[PATH_CAP];
    uint64_t path_key[PATH_CAP];
    uint8_t  path_chk[PATH_CAP];
    double   fbuf[128];
    int      ibuf[128];
};

__device__ __forceinline__ void mem_fence_wave()
{
    // tree arrays are written and re-read by different lanes of the same wave
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

__device__ __forceinline__ GameS load_gs(const GameS *p)
{
    union { uint4 u; GameS g; } x;
    x.u = *reinterpret_cast<const uint4 *>(p);
    return x.g;
}
__device__ __forceinline__ void store_gs(GameS *p, const GameS &g)
{
    union { uint4 u; GameS g; } x;
    x.g = g;
    if (XQ_LANE == 0) *reinterpret_cast<uint4 *>(p) = x.u;
}

__device__ __forceinline__ MState to_mstate(const GameS &g)
{
    MState s;
    s.side = g.side; s.move_count = g.move_count; s.winner = g.winner; s.reason = g.reason;
    s.reason_side = g.reason_side; s.reason_count = g.reason_count; s.rk = g.rk; s.bk = g.bk;
    s.nocap = g.nocap; s.cchk = g.cchk;
    return s;
}
__device__ __forceinline__ void from_mstate(GameS &g, const MState &s)
{
    g.side = (int8_t)s.side; g.move_count = (uint8_t)s.move_count; g.winner = (int8_t)s.winner;
    g.reason = (uint8_t)s.reason; g.reason_side = (int8_t)s.reason_side; g.reason_count = (uint8_t)s.reason_count;
    g.rk = (int8_t)s.rk; g.bk = (int8_t)s.bk; g.nocap = (uint8_t)(s.nocap > 255 ? 255 : s.nocap);
    g.cchk = (uint8_t)(s.cchk > 255 ? 255 : s.cchk);
}

// ---- history adapters for wave_make_move -------------------------------------------------
struct HistPath {               // in-search env: starts empty (self_play.py:173-174)
    uint64_t *keys; uint8_t *chk; int n; bool keys_on;
    __device__ bool want_keys() const { return keys_on; }
    __device__ void push(uint64_t key, int c)
    {
        if (XQ_LANE == 0 && n < PATH_CAP) { keys[n] = key; chk[n] = (uint8_t)c; }
        n++;
        wave_sync();
    }
    __device__ int count_key(uint64_t key) const
    {
        int cnt = 0;
        for (int b = 0; b < n && b < PATH_CAP; b += 64) {
            int i = b + XQ_LANE;
            cnt += __builtin_popcountll(__ballot(i < n && i < PATH_CAP && keys[i] == key));
        }
        return cnt;
    }
    __device__ bool perpetual() const
    {
        if (n < 12) return false;
        int i = n - 12 + XQ_LANE;
        return __builtin_popcountll(__ballot(XQ_LANE < 12 && i < PATH_CAP && chk[i] != 0)) >= 10;
    }
};

struct HistGlobal {             // real env: history kept by the caller (HBM), new entry returned
    const uint64_t *keys; const uint8_t *chk; int n_keys, n_chk;
    uint64_t new_key; int new_chk;
    __device__ bool want_keys() const { return true; }
    __device__ void push(uint64_t key, int c) { new_key = key; new_chk = c; }
    __device__ int count_key(uint64_t key) const
    {
        int cnt = (new_key == key) ? 1 : 0;
        for (int b = 0; b < n_keys; b += 64) {
            int i = b + XQ_LANE;
            cnt += __builtin_popcountll(__ballot(i < n_keys && keys[i] == key));
        }
        return cnt;
    }
    __device__ bool perpetual() const
    {
        if (n_chk + 1 < 12) return false;
        int i = n_chk - 11 + XQ_LANE;            // last 11 stored entries + the new one
        int c = __builtin_popcountll(__ballot(XQ_LANE < 11 && chk[i] != 0));
        return c + (new_chk ? 1 : 0) >= 10;
    }
};

// ---- network input planes (neural_network.py:128-146) ------------------------------------
__device__ __forceinline__ float plane_value(const int8_t *bd, int side, int c, int s)
{
    if (c == 14) return side == 1 ? 1.0f : 0.0f;
    int code = c < 7 ? c + 1 : -(c - 6);
    return bd[s] == code ? 1.0f : 0.0f;
}

__device__ void write_planes(const int8_t *bd, int side, void *planes, int fmt, int g)
{
    const int lane = XQ_LANE;
    if (fmt == XQ_PLANES_NCHW_F32) {
        float *o = reinterpret_cast<float *>(planes) + (size_t)g * 1350;
        for (int e = lane; e < 1350; e += 64) o[e] = plane_value(bd, side, e / 90, e % 90);
    } else if (fmt == XQ_PLANES_NCHW_BF16) {
        uint32_t *o = reinterpret_cast<uint32_t *>(planes) + (size_t)g * 675;
        for (int i = lane; i < 675; i += 64) {
            int e0 = 2 * i, e1 = 2 * i + 1;
            uint32_t lo = plane_value(bd, side, e0 / 90, e0 % 90) != 0.0f ? 0x3F80u : 0u;
            uint32_t hi = plane_value(bd, side, e1 / 90, e1 % 90) != 0.0f ? 0x3F80u : 0u;
            o[i] = lo | (hi << 16);
        }
    } else if (fmt == XQ_PLANES_NHWC16_BF16) {
        uint4 *o = reinterpret_cast<uint4 *>(planes) + (size_t)g * 180;     // 90 squares x 32 B
        const uint32_t w7 = side == 1 ? 0x3F80u : 0u;                       // channel 14 (15 is padding)
        for (int s = lane; s < 90; s += 64) {
            const int p = bd[s];
            const int ch = p > 0 ? p - 1 : (p < 0 ? 6 - p : -2);            // 0..13, none: -2 (word -1)
            const uint32_t val = 0x3F80u << ((ch & 1) * 16);
            const int wi = ch >> 1;
            // (selects, not an indexed local array: that would live in scratch memory - a round trip per store)
            o[2 * s] = make_uint4(wi == 0 ? val : 0u, wi == 1 ? val : 0u, wi == 2 ? val : 0u, wi == 3 ? val : 0u);
            o[2 * s + 1] = make_uint4(wi == 4 ? val : 0u, wi == 5 ? val : 0u, wi == 6 ? val : 0u, w7);
        }
    }
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// ---- tree primitives ---------------------------------------------------------------------
struct Tree {
    uint32_t *N; double *W; float *P; uint16_t *mv; uint16_t *first; uint8_t *nc; uint8_t *fl; uint8_t *vl;
};
__device__ __forceinline__ Tree tree_of(const Eng &E, int g)
{
    size_t o = (size_t)g * E.ncap;
    return Tree{ E.nN + o, E.nW + o, E.nP + o, E.nMove + o, E.nFirst + o, E.nNc + o, E.nFlags + o, E.nVl + o };
}

// self_play.py:40-59 — float32 stepwise PUCT (NumPy >= 2 scalar rules), first maximum wins
template <bool VL>
__device__ int select_child(const Tree &T, int node)
{
    const int lane = XQ_LANE;
    const int first = T.first[node], nc = T.nc[node];
    const float sq = (float)sqrt((double)(T.N[node] + (VL ? (uint32_t)T.vl[node] : 0u)));
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < nc; c += 64) {
        uint32_t n = T.N[first + c];
        double w = T.W[first + c];
        const float p = T.P[first + c];
        if (VL) {                                   // pending visits count as losses
            const uint32_t v = T.vl[first + c];
            n += v;
            w -= (double)v;
        }
        float q = n ? (float)(w / (double)n) : 0.0f;
        float t = 1.5f * p;
        t = t * sq;
        t = t / (float)(1u + n);
        float s = q + t;
        if (s > best) { best = s; bi = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        float os = __shfl_xor(best, d, 64);
        int oi = __shfl_xor(bi, d, 64);
        if (os > best || (os == best && oi < bi)) { best = os; bi = oi; }
    }
    return first + uni(bi);
}

// self_play.py:70-80: `mult` sequential updates of value v at the node on level `depth`
// (levels: 0 = root, i = path_node[i-1]); every ancestor alternates the sign.
__device__ void backup(const Tree &T, int root, const uint16_t *path_node, int depth, double v, int mult)
{
    const int lane = XQ_LANE;
    if (lane <= depth) {
        const int x = lane == 0 ? root : path_node[lane - 1];
        const double sv = ((depth - lane) & 1) ? -v : v;
        double w = T.W[x];
        for (int i = 0; i < mult; i++) w += sv;
        T.W[x] = w;
        T.N[x] += (uint32_t)mult;
    }
    mem_fence_wave();
}

// ---- Dirichlet root noise (AlphaZero extension; absent from the reference) ------------------
__device__ __forceinline__ float u01(uint64_t key)
{
    return ((float)(mix64(key) >> 40) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1)
}

// Gamma(alpha, 1) by Marsaglia-Tsang on alpha + 1 and the U^(1/alpha) boost; counter-based stream
__device__ float gamma_sample(float alpha, uint64_t key)
{
    const float d = alpha + 1.0f - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
    float g = d;
    for (int t = 0; t < 16; t++) {
        const float u1 = u01(key + 4 * t + 1), u2 = u01(key + 4 * t + 2), u3 = u01(key + 4 * t + 3);
        const float x = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
        const float v0 = 1.0f + c * x;
        if (v0 <= 0.0f) continue;
        const float v = v0 * v0 * v0;
        if (logf(u3) < 0.5f * x * x + d - d * v + d * logf(v)) { g = d * v; break; }
    }
    return g * powf(u01(key), 1.0f / alpha);
}

// P' = (1 - eps) P + eps * eta, eta ~ Dirichlet(alpha) over the n children of the root (lane j and
// lane j + 64 hold children j and j + 64); the stream is keyed by (seed, game, ply)
__device__ void root_noise(const Eng &E, int g, int ply, int n, float &p0, float &p1)
{
    const int lane = XQ_LANE;
    const uint64_t base = E.noise_seed ^ mix64(((uint64_t)g << 20) ^ ((uint64_t)ply << 8));
    float g0 = lane < n ? gamma_sample((float)E.noise_alpha, base + (uint64_t)lane * 1024u) : 0.f;
    float g1 = lane + 64 < n ? gamma_sample((float)E.noise_alpha, base + (uint64_t)(lane + 64) * 1024u) : 0.f;
    float gs_ = g0 + g1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) gs_ += __shfl_xor(gs_, d, 64);
    const float eps = (float)E.noise_eps;
    p0 = (1.0f - eps) * p0 + eps * (g0 / gs_);
    p1 = (1.0f - eps) * p1 + eps * (g1 / gs_);
}

// self_play.py:61-68 + 146-148: expand the pending leaf with its priors and apply its backups
template <bool VL>
__device__ void consume_eval(const Eng &E, int g, int slot, WaveLds &L, const Tree &T, int root, int eval_kind,
                             const void *ev_a, const void *ev_v, int ply)
{
    const int lane = XQ_LANE;
    const int node = E.leaf_node[slot];
    if (node == LEAF_NONE) return;
    const int n = E.leaf_n[slot], mult = E.leaf_mult[slot], depth = E.leaf_depth[slot];
    const uint16_t *lm = E.leaf_moves + (size_t)slot * MAXM;
    double v;
    float p0 = 0.f, p1 = 0.f;
    const int m0 = lane < n ? lm[lane] : 0, m1 = lane + 64 < n ? lm[lane + 64] : 0;
    const int ec = E.ec_on ? uni(E.leaf_ec[slot]) : -1;      // evaluation cache: >= 0 it answers, <= -2 this wave fills entry -(ec + 2)
    if (ec >= 0 && E.ec_on == 1) {
        const float *pr = E.ec_prior + (size_t)ec * MAXM;
        if (lane < n) p0 = ec_oldf(pr + lane);
        if (lane + 64 < n) p1 = ec_oldf(pr + lane + 64);
        v = (double)__uint_as_float(ec_old(&E.ec_key[(size_t)ec * 16 + 14]));
        if (lane == 0) E.ec_stats[slot] += 1u;
    } else if (eval_kind == XQ_EVAL_PRIORS) {
        const float *pr = reinterpret_cast<const float *>(ev_a) + (size_t)slot * MAXM;
        if (lane < n) p0 = pr[lane];
        if (lane + 64 < n) p1 = pr[lane + 64];
        v = reinterpret_cast<const double *>(ev_v)[slot];
    } else {
        // neural_network.py:148-169: gather the legal logits, float32 softmax over them
        float x0 = -INFINITY, x1 = -INFINITY;
        const int stride = E.col_map ? E.n_cols : XQ_POLICY_SIZE;
        const int row = E.compact ? E.leaf_row[slot] : slot;          // the network's output row of this leaf
        const int c0 = (E.col_map && lane < n) ? E.col_map[m0] : m0;
        const int c1 = (E.col_map && lane + 64 < n) ? E.col_map[m1] : m1;
        if (eval_kind == XQ_EVAL_LOGITS_F32) {
            const float *lg = reinterpret_cast<const float *>(ev_a) + (size_t)row * stride;
            if (lane < n) x0 = lg[c0];
            if (lane + 64 < n) x1 = lg[c1];
            v = (double)reinterpret_cast<const float *>(ev_v)[row];
        } else {
            const uint16_t *lg = reinterpret_cast<const uint16_t *>(ev_a) + (size_t)row * stride;
            if (lane < n) x0 = bf16_to_f32(lg[c0]);
            if (lane + 64 < n) x1 = bf16_to_f32(lg[c1]);
            v = (double)bf16_to_f32(reinterpret_cast<const uint16_t *>(ev_v)[row]);
        }
        float mx = fmaxf(x0, x1);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        float e0 = lane < n ? expf(x0 - mx) : 0.f, e1 = lane + 64 < n ? expf(x1 - mx) : 0.f;
        float sum = e0 + e1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        p0 = e0 / sum; p1 = e1 / sum;
        if (ec <= -2) {
            eval_cache_fill(E, -ec - 2, lane < 12 ? E.leaf_board[(size_t)slot * 12 + lane] : 0u, E.leaf_side[slot], n, p0, p1, (float)v);
            if (lane == 0) E.ec_stats[E.ec_stat_stride + slot] += 1u;
        } else if (ec >= 0) {                                 // verify mode
            const float *pr = E.ec_prior + (size_t)ec * MAXM;
            const bool bad = (lane < n && __float_as_uint(ec_oldf(pr + lane)) != __float_as_uint(p0)) ||
                             (lane + 64 < n && __float_as_uint(ec_oldf(pr + lane + 64)) != __float_as_uint(p1)) ||
                             ec_old(&E.ec_key[(size_t)ec * 16 + 14]) != __float_as_uint((float)v) || ec_old(&E.ec_key[(size_t)ec * 16 + 13]) != (uint32_t)n;
            const bool any = __ballot(bad) != 0ull;
            if (lane == 0) { E.ec_stats[slot] += 1u; if (any) E.ec_stats[2 * E.ec_stat_stride + slot] += 1u; }
        }
    }
    if (E.noise_eps > 0.0 && node == root) root_noise(E, g, ply, n, p0, p1);
    const int first = (int)E.n_nodes[g];
    if (first + n <= E.ncap) {
        for (int h = 0; h < 2; h++) {
            int j = lane + 64 * h;
            if (j < n) {
                int x = first + j;
                T.N[x] = 0; T.W[x] = 0.0; T.P[x] = h ? p1 : p0; T.mv[x] = (uint16_t)(h ? m1 : m0);
                T.first[x] = 0; T.nc[x] = 0; T.fl[x] = 0;
                if (VL) T.vl[x] = 0;
            }
        }
        if (lane == 0) { T.first[node] = (uint16_t)first; T.nc[node] = (uint8_t)n; E.n_nodes[g] = (uint32_t)(first + n); }
    }
    if (lane < depth) L.path_node[lane] = E.leaf_path[(size_t)slot * PATH_CAP + lane];
    wave_sync();
    mem_fence_wave();
    if (VL && lane <= depth) {                       // the pending visits of this leaf are real now
        const int x = lane == 0 ? root : L.path_node[lane - 1];
        T.vl[x] = (uint8_t)(T.vl[x] - mult);
    }
    backup(T, root, L.path_node, depth, v, mult);
    if (lane == 0) E.leaf_node[slot] = LEAF_NONE;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__device__ void init_board(int8_t *bd)
{
    const int lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) {
        int p = 0;
        if (s < 90) {
            const int r = s / 9, c = s % 9;
            const int back[9] = { ROOK, KNIGHT, BISHOP, ADVISOR, KING, ADVISOR, BISHOP, KNIGHT, ROOK };
            if (r == 9) p = back[c];
            else if (r == 0) p = -back[c];
            else if (r == 7 && (c == 1 || c == 7)) p = CANNON;
            else if (r == 2 && (c == 1 || c == 7)) p = -CANNON;
            else if (r == 6 && (c % 2 == 0)) p = PAWN;
            else if (r == 3 && (c % 2 == 0)) p = -PAWN;
        }
        bd[s] = (int8_t)p;
    }
}

// ChineseChess.reset (chess_env.py:14-67) for the game in slot g, + the first root's move list
__device__ void new_game(const Eng &E, int g, WaveLds &L)
{
    const int lane = XQ_LANE;
    init_board(L.bd);
    wave_sync();
    BoardView v = load_view(L.bd);
    const int rk = 9 * 9 + 4, bk = 4;
    const uint32_t kab = build_attack_maps(L.maps, v, -1);
    const int n = wave_movegen(L.bd, v, L.maps, kab, 1, rk, bk, L.cand, L.legal, L.own_sq);
    for (int j = lane; j < n; j += 64) E.root_moves[(size_t)g * MAXM + j] = L.legal[j];
    if (lane < 12) E.board[(size_t)g * 12 + lane] = pack_dword(L.bd, lane);
    GameS gs;
    gs.side = 1; gs.move_count = 0; gs.winner = WINNER_NONE; gs.reason = R_NONE; gs.reason_side = 0;
    gs.reason_count = 0; gs.rk = (int8_t)rk; gs.bk = (int8_t)bk; gs.nocap = 0; gs.cchk = 0; gs.n_hist = 0;
    gs.n_root = (uint8_t)n; gs.done = 0; gs.n_samples = 0; gs.error = 0; gs.n_plies = 0;
    store_gs(E.gs + g, gs);
    const int K = E.leaf_slots;
    for (int k = lane; k < K; k += 64) { E.leaf_node[(size_t)g * K + k] = LEAF_NONE; E.leaf_mult[(size_t)g * K + k] = 0; }
    if (lane == 0) {
        E.root_node[g] = 0;
        if (E.eval_carry) { E.root_ready[g] = 0; atomicAdd(E.roots_not_ready, 1); }
    }
}

__global__ __launch_bounds__(64) void k_new_games(Eng E)
{
    __shared__ WaveLds L;
    new_game(E, blockIdx.x, L);
}

// roots from caller-provided envs (one staged int8 board + int32 state row per game)
__global__ __launch_bounds__(64) void k_set_roots(Eng E, const int8_t *boards, const int32_t *state)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) L.bd[s] = s < 90 ? boards[(size_t)g * 90 + s] : 0;
    wave_sync();
    const int32_t *st = state + (size_t)g * XQ_STATE_WORDS;
    BoardView v = load_view(L.bd);
    const int side = st[XQ_S_PLAYER], rk = st[XQ_S_RED_KING], bk = st[XQ_S_BLACK_KING];
    const uint32_t kab = build_attack_maps(L.maps, v, -side);
    const int n = wave_movegen(L.bd, v, L.maps, kab, side, rk, bk, L.cand, L.legal, L.own_sq);
    for (int j = lane; j < n; j += 64) E.root_moves[(size_t)g * MAXM + j] = L.legal[j];
    if (lane < 12) E.board[(size_t)g * 12 + lane] = pack_dword(L.bd, lane);
    GameS gs;
    gs.side = (int8_t)side; gs.move_count = (uint8_t)st[XQ_S_MOVE_COUNT]; gs.winner = (int8_t)st[XQ_S_WINNER];
    gs.reason = R_NONE; gs.reason_side = 0; gs.reason_count = 0; gs.rk = (int8_t)rk; gs.bk = (int8_t)bk;
    gs.nocap = (uint8_t)st[XQ_S_NO_CAPTURE]; gs.cchk = 0; gs.n_hist = 0; gs.n_root = (uint8_t)n;
    gs.done = (n == 0) ? 1 : 0; gs.n_samples = 0; gs.error = 0; gs.n_plies = 0;
    store_gs(E.gs + g, gs);
    const int K = E.leaf_slots;
    for (int k = lane; k < K; k += 64) { E.leaf_node[(size_t)g * K + k] = LEAF_NONE; E.leaf_mult[(size_t)g * K + k] = 0; }
    if (lane == 0) {
        E.root_node[g] = 0;
        if (E.eval_carry) {                             // a caller-provided root has no carried evaluation
            E.root_ready[g] = 0;
            atomicAdd(E.roots_not_ready, 1);
        }
    }
}

// Leaf dedupe, first half (called by the wave that has just recorded a pending leaf in `slot`): look the position - the 12
// packed board dwords + the side to move = everything encode_board reads, neural_network.py:128-146 - up in an
// open-addressing table shared by all games.  An entry carries the tag of the round that wrote it, so the table is never
// cleared; the first wave to arrive claims a free entry, later waves with the same position (compared in full, no hash is
// trusted) lower the entry to the smallest slot of the group with atomicMin and mark the slot that lost - themselves or the
// previous holder - in leaf_dup: after the kernel exactly the lowest slot of every group is unmarked, whatever the arrival
// order.  k_assign_rows gives rows to unmarked slots only.  All games of a step start from the same position and share
// their first plies with many others, so over whole games this removes the network rows of several plies in 70; the
// results cannot change: the evaluator's output for a position does not depend on the row it sits in.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "dedupe_insert: cross-XCD visibility rests on gfx950's lowering of relaxed agent-scope atomics (write-through / L2 bypass) and on the gfx9 s_waitcnt encoding; validate before building for another target"
#endif
// 64-bit hash of a position (12 packed board dwords, lane < 12, + side to move): wave-uniform.  Nothing trusts it: the
// dedupe table and the evaluation cache compare boards in full.
__device__ __forceinline__ uint64_t position_hash(uint32_t my_dword, int side)
{
    const int lane = XQ_LANE;
    uint64_t h = 0;
    if (lane < 12) h = mix64(((uint64_t)(lane + 1) << 32) | my_dword);
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)h, d, 64), hi = __shfl_xor((uint32_t)(h >> 32), d, 64);
        h ^= ((uint64_t)hi << 32) | lo;
    }
    const uint32_t hlo = __builtin_amdgcn_readfirstlane((uint32_t)h), hhi = __builtin_amdgcn_readfirstlane((uint32_t)(h >> 32));
    return mix64((((uint64_t)hhi << 32) | hlo) ^ (0x9E3779B97F4A7C15ull * (uint64_t)(side + 2)));
}

// The evaluation cache is written and read by waves of every XCD: every access goes past the L2s (agent-scope atomics), like
// the dedupe table's, so that nothing rests on what a kernel boundary does to the eight L2s.

// Evaluation cache, fill side: the wave that consumes a leaf whose probe reserved entry i writes the position, its priors
// (before any root noise) and its value, then the state word.  Readers accept the entry from the next launch on.
__device__ __forceinline__ void eval_cache_fill(const Eng &E, int i, uint32_t board_dword /* lane < 12 */, int side, int n,
                                                float p0, float p1, float value)
{
    const int lane = XQ_LANE;
    uint32_t k = board_dword;
    if (lane == 12) k = (uint32_t)(side + 2);
    else if (lane == 13) k = (uint32_t)n;
    else if (lane == 14) k = __float_as_uint(value);
    else if (lane == 15) k = 0u;
    if (lane < 16) ec_st(&E.ec_key[(size_t)i * 16 + lane], k);
    if (lane < n) ec_st(reinterpret_cast<uint32_t *>(&E.ec_prior[(size_t)i * MAXM + lane]), __float_as_uint(p0));
    if (lane + 64 < n) ec_st(reinterpret_cast<uint32_t *>(&E.ec_prior[(size_t)i * MAXM + 64 + lane]), __float_as_uint(p1));
    // (readers trust the state word from the next launch on: by then every store of this launch has completed)
    if (lane == 0) ec_st64(&E.ec_state[i], ((unsigned long long)E.ec_epoch << 32) | E.ec_seq);
}

struct DedupeLook {            // what dedupe_prepare hands to dedupe_finish
    unsigned pos;
    unsigned long long ent;
};

// first half: publish the leaf's board and side (past the L2s), take a first look at the position's table entry.
// Nothing of this wave enters the table here: the caller may do other work (move generation) while the stores land.
__device__ __forceinline__ DedupeLook dedupe_prepare(const Eng &E, int slot, uint32_t my_dword /* lane < 12: packed board dword */, int side,
                                                     uint64_t h)
{
    const int lane = XQ_LANE;
    // This leaf's board and side must be visible to every other wave before the table can hand them the slot.  The L2s of
    // the 8 XCDs are not coherent with each other inside a kernel: an agent-scope release fence would write the whole L2
    // back (measured: k_search_round 87 -> 320 us); instead the few words other waves read are stored and loaded with
    // agent-scope atomics (write-through / L2 bypass) and only their completion is waited for (dedupe_finish).
    if (lane < 12) __hip_atomic_store(&E.leaf_board[(size_t)slot * 12 + lane], my_dword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) __hip_atomic_store(&E.leaf_side[slot], (int8_t)side, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DedupeLook k;
    k.pos = (unsigned)h & (unsigned)E.dd_mask;
    // the first look at the table travels together with the stores above (the look is only a hint: the CAS decides)
    k.ent = __hip_atomic_load(&E.dd_tab[k.pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return k;
}

// Evaluation cache, look-up side (called for a leaf about to wait for the network).  Returns the entry that holds this
// position's evaluation (>= 0: no network row needed), or -(i + 2) after reserving entry i for it (the wave that consumes
// the leaf fills it), or -1 (no luck: evaluated, not kept).  An entry is read only if it was filled in an EARLIER launch -
// kernel boundaries make it visible, nothing inside a launch has to - and within the last two plies (the epoch): older
// entries are free to be taken over, so the table needs no clearing and stays as small as two plies of evaluations.
//
// `first` / `have_first`: the state word of the first entry, loaded by the caller ahead of time (the search kernel sends the
// look on its way before the move generation: one of the probe's two memory round trips is then hidden).
__device__ __forceinline__ unsigned eval_cache_pos(const Eng &E, uint64_t h) { return (unsigned)(h >> 24) & (unsigned)E.ec_mask; }

//
// `crowded` (wave-uniform): the dedupe table already holds an entry of this round under this position's hash - most likely
// this very position, put there by a wave whose own probe has just reserved the cache entry (in the opening thousands of
// waves arrive with one position).  Such a leaf still takes an answer the cache HAS, but does not join the race for an
// entry that is free: 16,384 compare-and-swaps on one word cost more than the cache saves.
__device__ __forceinline__ int eval_cache_probe(const Eng &E, uint64_t h, uint32_t my_dword, int side,
                                                unsigned long long first = 0ull, bool have_first = false, bool crowded = false)
{
    const int lane = XQ_LANE;
    const unsigned mask = (unsigned)E.ec_mask, epoch = E.ec_epoch, seq = E.ec_seq;
    unsigned pos = eval_cache_pos(E, h);
    for (int probe = 0; probe < 4; probe++, pos = (pos + 1) & mask) {
        const unsigned long long raw = (probe == 0 && have_first) ? first : ec_ld64(&E.ec_state[pos]);
        const unsigned long long st = uni64(raw);
        const unsigned ep = (unsigned)(st >> 32), filled = (unsigned)st;
        if (epoch - ep > 1u) {                                  // older than the ply before this one (or never used): take it
            if (crowded) return -1;
            unsigned long long old = 0;
            if (lane == 0) old = atomicCAS(&E.ec_state[pos], st, ((unsigned long long)epoch << 32) | 0xffffffffull);
            old = uni64(old);
            return old == st ? -(int)pos - 2 : -1;              // (lost the race: most likely to a wave with this very position)
        }
        if (filled >= seq) return -1;                           // reserved or filled in this launch: contents not visible yet
        const uint32_t theirs = lane < 13 ? ec_old(&E.ec_key[(size_t)pos * 16 + lane]) : 0u;
        const uint32_t mine = lane < 12 ? my_dword : (uint32_t)(side + 2);
        if (__ballot(lane < 13 && theirs != mine) == 0ull) {
            return (int)pos;
        }
    }
    return -1;
}

// second half: once the stores of dedupe_prepare have landed, enter the table
__device__ __forceinline__ void dedupe_finish(const Eng &E, int slot, uint32_t my_dword, int side, DedupeLook k)
{
    const int lane = XQ_LANE;
    const unsigned mask = (unsigned)E.dd_mask, tag = E.dd_tag;
    unsigned pos = k.pos;
    unsigned long long ent = k.ent;
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned)slot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // (compiler ordering; a one-wave workgroup gets no wait from it)
    __builtin_amdgcn_s_waitcnt(0x0070);                                 // vmcnt(0) lgkmcnt(0): the stores have landed
    for (unsigned probes = 0; probes <= mask; probes++) {              // (the table holds >= 2 entries per slot: never full)
        if (probes) ent = __hip_atomic_load(&E.dd_tab[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(ent >> 32) != tag) {                             // left by an earlier round: free
            unsigned long long old = 0;
            if (lane == 0) old = atomicCAS(&E.dd_tab[pos], ent, mine);
            old = uni64(old);
            if (old == ent) break;                                      // claimed: lowest slot of its position so far
            ent = old;                                                  // a wave of this round got there first
        }
        const unsigned other = (unsigned)ent;
        uint32_t theirs = 0;
        if (lane < 12) theirs = __hip_atomic_load(&E.leaf_board[(size_t)other * 12 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int oside = (int8_t)__hip_atomic_load(reinterpret_cast<const uint8_t *>(E.leaf_side) + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool same = __ballot(lane < 12 && theirs != my_dword) == 0ull && oside == side;
        if (same) {
            if (lane == 0) {
                // an entry only ever goes down: under a lower slot this one has lost already, no atomic needed (in the
                // opening all 16,384 games meet in one entry; about ln(16,384) of them ever lower it)
                unsigned loser = (unsigned)slot;
                if (other > (unsigned)slot) {
                    const unsigned long long old = atomicMin(&E.dd_tab[pos], mine);     // same tag on both sides: the lower slot wins
                    if ((unsigned)old > (unsigned)slot) loser = (unsigned)old;
                }
                E.leaf_dup[loser] = 1;
            }
            break;
        }
        pos = (pos + 1) & mask;
    }
    if (lane == 0) E.leaf_pos[slot] = (int32_t)pos;
}

__device__ __forceinline__ void dedupe_insert(const Eng &E, int slot, uint32_t my_dword /* lane < 12: packed board dword */, int side)
{
    dedupe_finish(E, slot, my_dword, side, dedupe_prepare(E, slot, my_dword, side, position_hash(my_dword, side)));
}

__device__ __forceinline__ void record_leaf(const Eng &E, int slot, WaveLds &L, const int8_t *bd, int side, int node,
                            int depth, int mult, const uint16_t *moves, int n, void *planes, int fmt,
                            unsigned long long *st = nullptr)
{
    const int lane = XQ_LANE;
    for (int j = lane; j < n; j += 64) E.leaf_moves[(size_t)slot * MAXM + j] = moves[j];
    if (lane < depth) E.leaf_path[(size_t)slot * PATH_CAP + lane] = L.path_node[lane];
    const uint32_t my_dword = lane < 12 ? pack_dword(bd, lane) : 0u;
    if (!E.dedupe) {                                          // (with the dedupe: stored by dedupe_insert, past the L2)
        if (lane < 12) E.leaf_board[(size_t)slot * 12 + lane] = my_dword;
        if (lane == 0) E.leaf_side[slot] = (int8_t)side;
    }
    if (lane == 0) {
        E.leaf_node[slot] = (uint16_t)node; E.leaf_mult[slot] = (uint8_t)mult; E.leaf_n[slot] = (uint8_t)n;
        E.leaf_depth[slot] = (uint8_t)depth;
    }
    if (planes) write_planes(bd, side, planes, fmt, slot);
    if (st && lane == 0) st[8] = __builtin_amdgcn_s_memtime();
    const uint64_t h = (E.dedupe || E.ec_on) ? position_hash(my_dword, side) : 0ull;
    DedupeLook look{ 0u, 0ull };
    if (E.dedupe) look = dedupe_prepare(E, slot, my_dword, side, h);        // (board + side past the L2s, first look at the table)
    int ec = -1;
    if (E.ec_on) {
        ec = eval_cache_probe(E, h, my_dword, side, 0ull, false, E.dedupe && (unsigned)(look.ent >> 32) == E.dd_tag);
        if (lane == 0) E.leaf_ec[slot] = ec;
    }
    // (last: the planes' stores travel while it waits for the table; a leaf the cache answers needs no row at all)
    if (E.dedupe && (ec < 0 || E.ec_on == 2)) dedupe_finish(E, slot, my_dword, side, look);
    if (st && lane == 0) st[9] = __builtin_amdgcn_s_memtime();
}

// Phase stamps (probes library only, xq_engine_set_search_stamps): lane 0 of every wave stores s_memtime at the phase
// boundaries of a search round, 16 u64 per game: where a wave's ~50 k cycles go.
#if XQ_TOWER_PROBES
#define XQ_STAMP(i) do { if (STAMP && XQ_LANE == 0) stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XQ_STAMP(i) do { } while (0)
#endif

// OCC = minimum waves per SIMD requested from the register allocator.  The kernel is a chain of
// dependent global loads (tree walk) with ~1.6 k VALU per wave, i.e. latency-bound: occupancy is
// the lever (PMC: 62 % of wave cycles in s_waitcnt at 3 waves/SIMD).
// The round as rounds 1-4 ran it: every step loads what it needs when it needs it (~14 dependent memory round trips per
// wave).  Since round 5 only the virtual-loss build (K pending leaves per game, opt-in extension) runs this body; the
// reference's search - one pending leaf per game and round - runs search_round_one below.
template <bool VL>
__device__ void search_round_generic(const Eng &E, WaveLds &L, int round, int batch_count, int eval_kind,
                                     const void *ev_a, const void *ev_v, void *planes, int fmt)
{
    const int g = blockIdx.x, lane = XQ_LANE;
    const GameS gs = load_gs(E.gs + g);
    if (gs.done) return;
    const Tree T = tree_of(E, g);
    const int K = VL ? E.leaf_slots : 1;             // pending-leaf slots of this game: g * K + k
    int root = 0;

    if (round == 0 && E.eval_carry && E.root_ready[g]) return;     // k_play_move already built this ply's expanded root
    if (round == 0) {
        // tree reuse (opt-in): k_play_move left the played child as the root when it had been
        // expanded; keep its subtree while a whole ply's expansions still fit the arena
        if (E.tree_reuse) {
            root = E.root_node[g];
            if (root != 0 && (T.nc[root] == 0 || (int)E.n_nodes[g] + E.nrounds * K * MAXM > E.ncap)) root = 0;
        }
        if (root == 0) {
            if (lane == 0) {                         // fresh tree every ply (self_play.py:98)
                T.N[0] = 0; T.W[0] = 0.0; T.P[0] = 0.f; T.mv[0] = 0; T.first[0] = 0; T.nc[0] = 0; T.fl[0] = 0;
                if (VL) T.vl[0] = 0;
                E.n_nodes[g] = 1; E.root_node[g] = 0;
            }
            for (int k = lane; k < K; k += 64) E.leaf_node[(size_t)g * K + k] = LEAF_NONE;
        } else if (E.noise_eps > 0.0) {              // the kept root gets this ply's noise on its stored priors
            const int n = T.nc[root], first = T.first[root];
            float p0 = lane < n ? T.P[first + lane] : 0.f, p1 = lane + 64 < n ? T.P[first + lane + 64] : 0.f;
            root_noise(E, g, gs.n_plies, n, p0, p1);
            if (lane < n) T.P[first + lane] = p0;
            if (lane + 64 < n) T.P[first + lane + 64] = p1;
        }
        mem_fence_wave();
    } else {
        root = E.root_node[g];
        for (int k = 0; k < K; k++)
            consume_eval<VL>(E, g, g * K + k, L, T, root, eval_kind, ev_a, ev_v, gs.n_plies);
    }

    unpack_to_lds(E.board + (size_t)g * 12, L.root_bd);
    wave_sync();

    int sims_left = batch_count;
    int nslot = 0;                                   // VL: slots handed out in this round
    while (sims_left > 0) {
        // ---- select (self_play.py:117-119); the tree is frozen unless a terminal leaf updates it
        int node = root, depth = 0;
        while (T.nc[node] != 0 && depth < PATH_CAP) {
            const int child = select_child<VL>(T, node);
            if (lane == 0) { L.path_node[depth] = (uint16_t)child; L.path_move[depth] = T.mv[child]; }
            node = child;
            depth++;
        }
        wave_sync();
        int flags = T.fl[node];
        if (VL && !(flags & F_TERM) && T.vl[node] != 0) {
            // this leaf is already waiting for the network in one of this round's slots: one more
            // pending visit through the same path
            const int mine = (lane < nslot && E.leaf_node[(size_t)g * K + lane] == node) ? 1 : 0;
            const int k = __ffsll((unsigned long long)__ballot(mine)) - 1;
            if (lane == 0 && k >= 0) E.leaf_mult[(size_t)g * K + k] += 1;
            if (lane <= depth) { const int x = lane == 0 ? root : L.path_node[lane - 1]; T.vl[x] += 1; }
            mem_fence_wave();
            sims_left--;
            continue;
        }
        if (!(flags & F_TERM)) {
            if (node == root) {
                // the root is never terminal (the driver checked legal moves, self_play.py:205-208)
                record_leaf(E, g * K + nslot, L, L.root_bd, gs.side, root, 0, VL ? 1 : sims_left,
                            E.root_moves + (size_t)g * MAXM, gs.n_root, planes, fmt);
                if (!VL) return;
                if (lane == 0) T.vl[root] += 1;
                mem_fence_wave();
                nslot++;
                sims_left--;
                continue;
            }
            // ---- replay the path on a copy of the root env (_copy_env, self_play.py:156-175)
            for (int s = lane; s < 24; s += 64)
                reinterpret_cast<uint32_t *>(L.bd)[s] = reinterpret_cast<const uint32_t *>(L.root_bd)[s];
            wave_sync();
            MState st = to_mstate(gs);
            st.cchk = 0; st.reason = R_NONE;
            HistPath hist{ L.path_key, L.path_chk, 0, depth >= 6 };
            for (int i = 0; i + 1 < depth; i++) {
                // interior nodes were classified non-terminal when first reached; only the
                // state that later plies read is replayed (board, caches, counters, history)
                const int mv = L.path_move[i], from = mv / 90, to = mv % 90;
                const int captured = L.bd[to], moving = L.bd[from];
                wave_sync();
                if (lane == 0) { L.bd[to] = (int8_t)moving; L.bd[from] = 0; }
                wave_sync();
                st.rk = (captured == KING) ? NO_KING : ((moving == KING) ? to : st.rk);
                st.bk = (captured == -KING) ? NO_KING : ((moving == -KING) ? to : st.bk);
                st.nocap = (captured != 0) ? 0 : st.nocap + 1;
                const uint64_t key = hist.keys_on ? position_key(L.bd, st.side == 1 ? 0 : 1) : 0ull;
                const int chk = (T.fl[L.path_node[i]] & F_CHECK) ? 1 : 0;
                hist.push(key, chk);
                st.side = -st.side;
                st.move_count += 1;
            }
            MoveResult mr;
            if (E.want_check)
                mr = wave_make_move<false, true>(L.bd, st, L.path_move[depth - 1], hist, L.maps, L.cand, L.legal, L.own_sq);
            else
                mr = wave_make_move<false, false>(L.bd, st, L.path_move[depth - 1], hist, L.maps, L.cand, L.legal, L.own_sq);
            flags |= mr.is_check ? F_CHECK : 0;
            const bool terminal = (mr.n_legal == 0) || (st.winner != WINNER_NONE);     // self_play.py:126
            if (!terminal) {
                if (lane == 0) T.fl[node] = (uint8_t)flags;
                record_leaf(E, g * K + nslot, L, L.bd, st.side, node, depth, VL ? 1 : sims_left, L.legal, mr.n_legal,
                            planes, fmt);
                if (!VL) return;
                if (lane <= depth) { const int x = lane == 0 ? root : L.path_node[lane - 1]; T.vl[x] += 1; }
                mem_fence_wave();
                nslot++;
                sims_left--;
                continue;
            }
            // self_play.py:128-133, value from the side to move at the leaf
            flags |= F_TERM;
            if (st.winner == st.side) flags |= F_VAL_POS;
            else if (st.winner == -st.side) flags |= F_VAL_NEG;
            if (lane == 0) T.fl[node] = (uint8_t)flags;
        }
        const double v = (flags & F_VAL_POS) ? 1.0 : ((flags & F_VAL_NEG) ? -1.0 : 0.0);
        backup(T, root, L.path_node, depth, v, 1);                                     // self_play.py:135
        sims_left--;
    }
}

// self_play.py:40-59 on the root's children held in registers (lane j: children j and j + 64): the same float32 steps
// in the same order as select_child, first maximum wins
__device__ __forceinline__ int select_child_regs(int nc, uint32_t n_parent, uint32_t n0, uint32_t n1, double w0, double w1,
                                                 float p0, float p1)
{
    const int lane = XQ_LANE;
    const float sq = (float)sqrt((double)n_parent);
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int c = lane + 64 * h;
        if (c < nc) {
            const uint32_t n = h ? n1 : n0;
            const double w = h ? w1 : w0;
            const float p = h ? p1 : p0;
            float q = n ? (float)(w / (double)n) : 0.0f;
            float t = 1.5f * p;
            t = t * sq;
            t = t / (float)(1u + n);
            float sc = q + t;
            if (sc > best) { best = sc; bi = c; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        float os = __shfl_xor(best, d, 64);
        int oi = __shfl_xor(bi, d, 64);
        if (os > best || (os == best && oi < bi)) { best = os; bi = oi; }
    }
    return uni(bi);
}

// value of child `c` (wave-uniform) out of a pair of per-lane registers (lane j: children j and j + 64)
__device__ __forceinline__ int pick_child(int v0, int v1, int c)
{
    return (c < 64) ? __builtin_amdgcn_readlane(v0, c) : __builtin_amdgcn_readlane(v1, c - 64);
}

// One search round of one game (self_play.py:103-148), the reference's form: ONE pending leaf per game and round.
// Round 5 restructure of the latency chain (tools/stamps_search.py: of a wave's ~41 k cycles, 11 k were consume_eval's
// six dependent loads, 7.5 k the descent's three, 4 k the dedupe insert behind the leaf's stores):
//   (A) everything addressed by the game alone - game record, pending-leaf header, its moves and path, its evaluator
//       row, the arena fill, the root board, the root's node - is loaded in ONE round trip at the top;
//   (B) the second round trip carries the policy columns of the pending moves, the W / N of the path's nodes for the
//       backup AND the root's children (N, W, P, move, child block, flags) for this round's descent;
//   (C) the third the logits.  The expansion and the backup then update memory as before and the same values in the
//       registers, so the first level of the descent - with random-init priors almost always the only one - and the
//       chosen child's move / flags need no memory at all.  Deeper levels walk memory as before (select_child).
//   The leaf's packed board, its planes and the first look at the dedupe table leave right after the board is updated
//   (wave_make_move's hook), so the table insert no longer waits a round trip behind ~3 KB of stores.
template <bool STAMP>
__device__ void search_round_one(const Eng &E, WaveLds &L, int round, int batch_count, int eval_kind,
                                 const void *ev_a, const void *ev_v, void *planes, int fmt, unsigned long long *stamps)
{
    const int g = blockIdx.x, lane = XQ_LANE, slot = g;
    XQ_STAMP(0);
    const Tree T = tree_of(E, g);
    unsigned long long *const stp = STAMP ? stamps + (size_t)g * 16 : nullptr;

    // ---- (A): straight-line, no branch between the loads (a load under an `if` is issued when the branch is taken,
    // i.e. one round trip later); what a round does not need is loaded from a valid address and ignored
    union { uint4 u; GameS g; } gx;
    gx.u = *reinterpret_cast<const uint4 *>(E.gs + g);
    const uint32_t bw = lane < 12 ? E.board[(size_t)g * 12 + lane] : 0u;
    const uint8_t *ready_p = E.eval_carry ? E.root_ready + g : reinterpret_cast<const uint8_t *>(E.leaf_n + slot);
    const int32_t *row_p = E.compact ? E.leaf_row + slot : reinterpret_cast<const int32_t *>(E.n_nodes + g);
    const int ready_v = *ready_p;
    const int row_v = *row_p;
    const int root_v = E.root_node[g];
    int p_node = E.leaf_node[slot], p_n = E.leaf_n[slot], p_mult = E.leaf_mult[slot], p_depth = E.leaf_depth[slot];
    int p_first = (int)E.n_nodes[g];
    const int m0 = E.leaf_moves[(size_t)slot * MAXM + lane], m1 = E.leaf_moves[(size_t)slot * MAXM + 64 + lane];
    const int pth = E.leaf_path[(size_t)slot * PATH_CAP + lane];
    const int32_t *ec_p = E.ec_on ? E.leaf_ec + slot : reinterpret_cast<const int32_t *>(E.n_nodes + g);
    const int ec_v = *ec_p;
    const uint32_t lbw = lane < 12 ? E.leaf_board[(size_t)slot * 12 + lane] : 0u;      // (the pending leaf's position: key of the entry it fills)
    const int lside = E.leaf_side[slot];
    int root = (round > 0 && E.tree_reuse) ? root_v : 0;
    int r_nc = T.nc[root], r_first = T.first[root];
    uint32_t r_N = T.N[root];
    // (whole dwords of the game record: field-wise narrow loads would each be a request of their own)
    gx.u.x = (uint32_t)uni((int)gx.u.x); gx.u.y = (uint32_t)uni((int)gx.u.y);
    gx.u.z = (uint32_t)uni((int)gx.u.z); gx.u.w = (uint32_t)uni((int)gx.u.w);
    const GameS gs = gx.g;
    const int ready = (round == 0 && E.eval_carry) ? ready_v : 0;
    int p_row = E.compact ? row_v : slot;
    if (round == 0) p_node = LEAF_NONE;
    if (gs.done) return;
    if (ready) return;                               // k_play_move already built this ply's expanded root
    XQ_STAMP(1);
    if (lane < 12) {                                 // root board -> LDS (unpack_to_lds from the register)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t code = (bw >> (4 * j)) & 15u;
            L.root_bd[8 * lane + j] = (int8_t)(code <= 7 ? (int)code : 7 - (int)code);
        }
    }

    if (round == 0) {
        // tree reuse (opt-in): k_play_move left the played child as the root when it had been
        // expanded; keep its subtree while a whole ply's expansions still fit the arena
        if (E.tree_reuse) {
            root = E.root_node[g];
            if (root != 0 && (T.nc[root] == 0 || (int)E.n_nodes[g] + E.nrounds * MAXM > E.ncap)) root = 0;
        }
        if (root == 0) {
            if (lane == 0) {                         // fresh tree every ply (self_play.py:98)
                T.N[0] = 0; T.W[0] = 0.0; T.P[0] = 0.f; T.mv[0] = 0; T.first[0] = 0; T.nc[0] = 0; T.fl[0] = 0;
                E.n_nodes[g] = 1; E.root_node[g] = 0;
                E.leaf_node[slot] = LEAF_NONE;
            }
            r_nc = 0; r_first = 0; r_N = 0;
        } else {
            if (E.noise_eps > 0.0) {                 // the kept root gets this ply's noise on its stored priors
                const int n = T.nc[root], first = T.first[root];
                float q0 = lane < n ? T.P[first + lane] : 0.f, q1 = lane + 64 < n ? T.P[first + lane + 64] : 0.f;
                root_noise(E, g, gs.n_plies, n, q0, q1);
                if (lane < n) T.P[first + lane] = q0;
                if (lane + 64 < n) T.P[first + lane + 64] = q1;
            }
            r_nc = T.nc[root]; r_first = T.first[root]; r_N = T.N[root];
        }
        mem_fence_wave();
    }
    r_nc = uni(r_nc); r_first = uni(r_first); r_N = (uint32_t)uni((int)r_N);
    p_node = uni(p_node); p_n = uni(p_n); p_mult = uni(p_mult); p_depth = uni(p_depth); p_row = uni(p_row); p_first = uni(p_first);
    const bool pending = p_node != LEAF_NONE;        // (round 0: never)
    const int p_ec = E.ec_on ? uni(ec_v) : -1;       // >= 0: the evaluation cache answers for the pending leaf; <= -2: it fills entry -(p_ec + 2)

    // ---- (B): the root's children as they stand before this round's expansion / backup
    // (cM = move | n_child << 16 | flags << 24 of the child: what the descent needs of the child it picks)
    uint32_t cN0 = 0, cN1 = 0, cM0 = 0, cM1 = 0;
    double cW0 = 0.0, cW1 = 0.0;
    float cP0 = 0.f, cP1 = 0.f;
    if (lane < r_nc) {
        const int x = r_first + lane;
        cN0 = T.N[x]; cW0 = T.W[x]; cP0 = T.P[x];
        cM0 = (uint32_t)T.mv[x] | ((uint32_t)T.nc[x] << 16) | ((uint32_t)T.fl[x] << 24);
    }
    if (lane + 64 < r_nc) {
        const int x = r_first + lane + 64;
        cN1 = T.N[x]; cW1 = T.W[x]; cP1 = T.P[x];
        cM1 = (uint32_t)T.mv[x] | ((uint32_t)T.nc[x] << 16) | ((uint32_t)T.fl[x] << 24);
    }
    if (pending) {
        // ---- consume the evaluator's output for the pending leaf (self_play.py:61-68 + 146-148)
        const int n = p_n, mult = p_mult, depth = p_depth;
        const int prev = __shfl_up(pth, 1, 64);                              // lane l: node on level l of the path
        const int bx = lane == 0 ? root : prev;
        double bW = 0.0;
        uint32_t bN = 0;
        if (lane <= depth) { bW = T.W[bx]; bN = T.N[bx]; }
        if (lane < depth) L.path_node[lane] = (uint16_t)pth;
        double v;
        float p0 = 0.f, p1 = 0.f;
        if (p_ec >= 0 && E.ec_on == 1) {
            // the evaluation cache holds this position's priors and value (the same bits the network would return)
            const float *pr = E.ec_prior + (size_t)p_ec * MAXM;
            if (lane < n) p0 = ec_oldf(pr + lane);
            if (lane + 64 < n) p1 = ec_oldf(pr + lane + 64);
            v = (double)__uint_as_float(ec_old(&E.ec_key[(size_t)p_ec * 16 + 14]));
            if (lane == 0) E.ec_stats[slot] += 1u;
        } else if (eval_kind == XQ_EVAL_PRIORS) {
            const float *pr = reinterpret_cast<const float *>(ev_a) + (size_t)slot * MAXM;
            if (lane < n) p0 = pr[lane];
            if (lane + 64 < n) p1 = pr[lane + 64];
            v = reinterpret_cast<const double *>(ev_v)[slot];
        } else {
            // neural_network.py:148-169: gather the legal logits, float32 softmax over them
            float x0 = -INFINITY, x1 = -INFINITY;
            const int stride = E.col_map ? E.n_cols : XQ_POLICY_SIZE;
            const int c0 = (E.col_map && lane < n) ? E.col_map[m0] : m0;
            const int c1 = (E.col_map && lane + 64 < n) ? E.col_map[m1] : m1;
            // ---- (C)
            if (eval_kind == XQ_EVAL_LOGITS_F32) {
                const float *lg = reinterpret_cast<const float *>(ev_a) + (size_t)p_row * stride;
                if (lane < n) x0 = lg[c0];
                if (lane + 64 < n) x1 = lg[c1];
                v = (double)reinterpret_cast<const float *>(ev_v)[p_row];
            } else {
                const uint16_t *lg = reinterpret_cast<const uint16_t *>(ev_a) + (size_t)p_row * stride;
                if (lane < n) x0 = bf16_to_f32(lg[c0]);
                if (lane + 64 < n) x1 = bf16_to_f32(lg[c1]);
                v = (double)bf16_to_f32(reinterpret_cast<const uint16_t *>(ev_v)[p_row]);
            }
            float mx = fmaxf(x0, x1);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
            float e0 = lane < n ? expf(x0 - mx) : 0.f, e1 = lane + 64 < n ? expf(x1 - mx) : 0.f;
            float sum = e0 + e1;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
            p0 = e0 / sum; p1 = e1 / sum;
            if (p_ec <= -2) {
                eval_cache_fill(E, -p_ec - 2, lbw, lside, n, p0, p1, (float)v);
                if (lane == 0) E.ec_stats[E.ec_stat_stride + slot] += 1u;
            } else if (p_ec >= 0) {
                // verify mode (ec_on == 2): the leaf was evaluated although the cache holds its position - they must agree
                const float *pr = E.ec_prior + (size_t)p_ec * MAXM;
                const bool bad = (lane < n && __float_as_uint(ec_oldf(pr + lane)) != __float_as_uint(p0)) ||
                                 (lane + 64 < n && __float_as_uint(ec_oldf(pr + lane + 64)) != __float_as_uint(p1)) ||
                                 ec_old(&E.ec_key[(size_t)p_ec * 16 + 14]) != __float_as_uint((float)v) || ec_old(&E.ec_key[(size_t)p_ec * 16 + 13]) != (uint32_t)n;
                const bool any = __ballot(bad) != 0ull;
                if (lane == 0) { E.ec_stats[slot] += 1u; if (any) E.ec_stats[2 * E.ec_stat_stride + slot] += 1u; }
            }
        }
        if (E.noise_eps > 0.0 && p_node == root) root_noise(E, g, gs.n_plies, n, p0, p1);
        const bool fits = p_first + n <= E.ncap;
        if (fits) {
            if (lane < n) {
                const int x = p_first + lane;
                T.N[x] = 0; T.W[x] = 0.0; T.P[x] = p0; T.mv[x] = (uint16_t)m0; T.first[x] = 0; T.nc[x] = 0; T.fl[x] = 0;
            }
            if (lane + 64 < n) {
                const int x = p_first + lane + 64;
                T.N[x] = 0; T.W[x] = 0.0; T.P[x] = p1; T.mv[x] = (uint16_t)m1; T.first[x] = 0; T.nc[x] = 0; T.fl[x] = 0;
            }
            if (lane == 0) { T.first[p_node] = (uint16_t)p_first; T.nc[p_node] = (uint8_t)n; E.n_nodes[g] = (uint32_t)(p_first + n); }
        }
        // self_play.py:70-80: `mult` sequential updates of value v at the leaf, signs alternating towards the root
        const double sv = ((depth - lane) & 1) ? -v : v;
        double w = bW;
        for (int i = 0; i < mult; i++) w += sv;
        if (lane <= depth) { T.W[bx] = w; T.N[bx] = bN + (uint32_t)mult; }
        if (lane == 0) E.leaf_node[slot] = LEAF_NONE;
        // ... and the same updates on the registers the descent reads
        r_N += (uint32_t)mult;
        if (depth == 0) {                                                    // the pending leaf was the root: its children are the new edges
            if (fits) {
                r_nc = n; r_first = p_first;
                cN0 = cN1 = 0; cW0 = cW1 = 0.0; cP0 = p0; cP1 = p1; cM0 = (uint32_t)m0; cM1 = (uint32_t)m1;
            }
        } else {
            const int ci = __builtin_amdgcn_readlane(pth, 0) - r_first;      // the root's child on the path
            const double w1 = __shfl(w, 1, 64);                              // its updated W (lane 1 of the backup)
            if (lane == (ci & 63)) {
                if (ci < 64) { cW0 = w1; cN0 += (uint32_t)mult; } else { cW1 = w1; cN1 += (uint32_t)mult; }
                if (depth == 1 && fits) {                                    // ... which is the leaf that was just expanded
                    if (ci < 64) cM0 = (cM0 & 0xff00ffffu) | ((uint32_t)n << 16); else cM1 = (cM1 & 0xff00ffffu) | ((uint32_t)n << 16);
                }
            }
        }
        mem_fence_wave();
    }
    wave_sync();
    XQ_STAMP(2);
    XQ_STAMP(3);

    // ---- select (self_play.py:117-119), first simulation of the round: level 0 from the registers (done in front of
    // the loop, so that the children's registers are dead before the move generation needs its own)
    int node0 = root, depth0 = 0, flags0 = -1;
    {
        int nc_node = r_nc;
        if (r_nc != 0) {
            const int c = select_child_regs(r_nc, r_N, cN0, cN1, cW0, cW1, cP0, cP1);
            node0 = r_first + c;
            const uint32_t meta = (uint32_t)pick_child((int)cM0, (int)cM1, c);
            nc_node = (int)((meta >> 16) & 0xffu);
            flags0 = (int)(meta >> 24);
            if (lane == 0) { L.path_node[0] = (uint16_t)node0; L.path_move[0] = (uint16_t)(meta & 0xffffu); }
            depth0 = 1;
        }
        while (nc_node != 0 && depth0 < PATH_CAP) {
            const int child = select_child<false>(T, node0);
            if (lane == 0) { L.path_node[depth0] = (uint16_t)child; L.path_move[depth0] = T.mv[child]; }
            node0 = child;
            depth0++;
            nc_node = T.nc[node0];
            flags0 = -1;
        }
        wave_sync();
        XQ_STAMP(4);
        if (flags0 < 0) flags0 = T.fl[node0];
    }

    int sims_left = batch_count;
    bool first = true;
    while (sims_left > 0) {
        // the tree is frozen unless a terminal leaf updates it: later simulations of the round walk memory
        int node = root, depth = 0, flags;
        if (first) {
            node = node0; depth = depth0; flags = flags0;
            first = false;
        } else {
            while (T.nc[node] != 0 && depth < PATH_CAP) {
                const int child = select_child<false>(T, node);
                if (lane == 0) { L.path_node[depth] = (uint16_t)child; L.path_move[depth] = T.mv[child]; }
                node = child;
                depth++;
            }
            wave_sync();
            flags = T.fl[node];
        }
        if (!(flags & F_TERM)) {
            if (node == root) {
                // the root is never terminal (the driver checked legal moves, self_play.py:205-208)
                record_leaf(E, slot, L, L.root_bd, gs.side, root, 0, sims_left, E.root_moves + (size_t)g * MAXM, gs.n_root,
                            planes, fmt, stp);
                return;
            }
            // ---- replay the path on a copy of the root env (_copy_env, self_play.py:156-175)
            XQ_STAMP(5);
            for (int s = lane; s < 24; s += 64)
                reinterpret_cast<uint32_t *>(L.bd)[s] = reinterpret_cast<const uint32_t *>(L.root_bd)[s];
            wave_sync();
            MState st = to_mstate(gs);
            st.cchk = 0; st.reason = R_NONE;
            HistPath hist{ L.path_key, L.path_chk, 0, depth >= 6 };
            for (int i = 0; i + 1 < depth; i++) {
                // interior nodes were classified non-terminal when first reached; only the
                // state that later plies read is replayed (board, caches, counters, history)
                const int mv = L.path_move[i], from = mv / 90, to = mv % 90;
                const int captured = L.bd[to], moving = L.bd[from];
                wave_sync();
                if (lane == 0) { L.bd[to] = (int8_t)moving; L.bd[from] = 0; }
                wave_sync();
                st.rk = (captured == KING) ? NO_KING : ((moving == KING) ? to : st.rk);
                st.bk = (captured == -KING) ? NO_KING : ((moving == -KING) ? to : st.bk);
                st.nocap = (captured != 0) ? 0 : st.nocap + 1;
                const uint64_t key = hist.keys_on ? position_key(L.bd, st.side == 1 ? 0 : 1) : 0ull;
                const int chk = (T.fl[L.path_node[i]] & F_CHECK) ? 1 : 0;
                hist.push(key, chk);
                st.side = -st.side;
                st.move_count += 1;
            }
            XQ_STAMP(6);
            // the leaf's position is known the moment its last move is on the board: its packed board (for the dedupe:
            // past the L2s), its planes and the first look at its table entry leave before the move generation
            const int leaf_side = -st.side;
            uint32_t my_dword = 0;
            uint64_t h = 0;
            unsigned long long ec_first = 0ull;
            DedupeLook look{ 0u, 0ull };
            auto early = [&](const int8_t *bd) {
                my_dword = lane < 12 ? pack_dword(bd, lane) : 0u;
                if (E.dedupe || E.ec_on) h = position_hash(my_dword, leaf_side);
                if (E.ec_on) ec_first = ec_ld64(&E.ec_state[eval_cache_pos(E, h)]);
                if (E.dedupe) look = dedupe_prepare(E, slot, my_dword, leaf_side, h);
                else {
                    if (lane < 12) E.leaf_board[(size_t)slot * 12 + lane] = my_dword;
                    if (lane == 0) E.leaf_side[slot] = (int8_t)leaf_side;
                }
                if (planes) write_planes(bd, leaf_side, planes, fmt, slot);
            };
            const MoveResult mr = wave_make_move<false, true>(L.bd, st, L.path_move[depth - 1], hist, L.maps, L.cand, L.legal, L.own_sq,
                                                              early, E.want_check != 0);
            flags |= mr.is_check ? F_CHECK : 0;
            XQ_STAMP(7);
            const bool terminal = (mr.n_legal == 0) || (st.winner != WINNER_NONE);     // self_play.py:126
            if (!terminal) {
                if (lane == 0) {
                    T.fl[node] = (uint8_t)flags;
                    E.leaf_node[slot] = (uint16_t)node; E.leaf_mult[slot] = (uint8_t)sims_left; E.leaf_n[slot] = (uint8_t)mr.n_legal;
                    E.leaf_depth[slot] = (uint8_t)depth;
                }
                for (int j = lane; j < mr.n_legal; j += 64) E.leaf_moves[(size_t)slot * MAXM + j] = L.legal[j];
                if (lane < depth) E.leaf_path[(size_t)slot * PATH_CAP + lane] = L.path_node[lane];
                if (STAMP && lane == 0) stp[8] = __builtin_amdgcn_s_memtime();
                // (the packed board again from LDS: cheaper than a register held across the move generation)
                const uint32_t bdw = lane < 12 ? pack_dword(L.bd, lane) : 0u;
                int ec = -1;
                if (E.ec_on) {
                    ec = eval_cache_probe(E, h, bdw, leaf_side, ec_first, true, E.dedupe && (unsigned)(look.ent >> 32) == E.dd_tag);
                    if (lane == 0) E.leaf_ec[slot] = ec;
                }
                if (E.dedupe && (ec < 0 || E.ec_on == 2)) dedupe_finish(E, slot, bdw, leaf_side, look);
                if (STAMP && lane == 0) stp[9] = __builtin_amdgcn_s_memtime();
                return;
            }
            // self_play.py:128-133, value from the side to move at the leaf
            flags |= F_TERM;
            if (st.winner == st.side) flags |= F_VAL_POS;
            else if (st.winner == -st.side) flags |= F_VAL_NEG;
            if (lane == 0) T.fl[node] = (uint8_t)flags;
        }
        const double v = (flags & F_VAL_POS) ? 1.0 : ((flags & F_VAL_NEG) ? -1.0 : 0.0);
        backup(T, root, L.path_node, depth, v, 1);                                     // self_play.py:135
        sims_left--;
    }
}

// OCC = minimum waves per SIMD requested from the register allocator.
template <int OCC, bool VL, bool STAMP = false>
__global__ __launch_bounds__(64, OCC) void k_search_round(Eng E, int round, int batch_count, int eval_kind,
                                                          const void *ev_a, const void *ev_v, void *planes, int fmt,
                                                          unsigned long long *stamps = nullptr)
{
    // one wave = one game = one workgroup (packing 4 games into a 256-thread workgroup was
    // measured 20-40 % slower at the same register budget: a workgroup retires only with its
    // slowest game)
    __shared__ WaveLds L;
    if constexpr (VL) search_round_generic<true>(E, L, round, batch_count, eval_kind, ev_a, ev_v, planes, fmt);
    else search_round_one<STAMP>(E, L, round, batch_count, eval_kind, ev_a, ev_v, planes, fmt, stamps);
}

__constant__ uint32_t c_crc_table[256];

__device__ __forceinline__ uint32_t crc_byte(uint32_t crc, uint32_t b)
{
    return c_crc_table[(crc ^ b) & 0xffu] ^ (crc >> 8);
}

// exact dyadic evaluator: priors k/1024, values m/64 from CRC-32
__global__ __launch_bounds__(64) void k_hashnet(Eng E, int salt)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;             // one block per pending-leaf slot
    if (E.gs[g / E.leaf_slots].done || E.leaf_node[g] == LEAF_NONE) return;
    unpack_to_lds(E.leaf_board + (size_t)g * 12, L.bd);
    wave_sync();
    uint32_t crc = 0xffffffffu;
    for (int s = 0; s < 90; s++) crc = crc_byte(crc, (uint32_t)(uint8_t)L.bd[s]);
    crc = crc_byte(crc, (uint32_t)(uint8_t)E.leaf_side[g]);
    if (salt) crc = crc_byte(crc, (uint32_t)salt & 0xffu);
    const uint32_t h0 = ~crc;
    const int n = E.leaf_n[g];
    for (int j = lane; j < n; j += 64) {
        const int mv = E.leaf_moves[(size_t)g * MAXM + j], from = mv / 90, to = mv % 90;
        uint32_t c = ~h0;
        c = crc_byte(c, from / 9); c = crc_byte(c, from % 9); c = crc_byte(c, to / 9); c = crc_byte(c, to % 9);
        const uint32_t h = ~c;
        E.priors[(size_t)g * MAXM + j] = (float)((h >> 8) % 64u + 1u) / 1024.0f;
    }
    if (lane == 0) E.values[g] = ((double)((h0 >> 4) % 65u) - 32.0) / 64.0;
}

// Evaluator row compaction: number the slots that hold a pending leaf, in slot order (deterministic), so that the
// network kernels touch only those rows.  One workgroup of 16 waves walks the G * K slots 1,024 at a time (ballot
// prefix inside a wave, 16 partial sums through LDS).  row_src[row] = slot tells the trunk kernel where the search
// kernel wrote that row's planes; leaf_row[slot] tells consume_eval where the row's logits and value are.
constexpr int ROW_HIST = 65536;

__global__ __launch_bounds__(1024) void k_assign_rows(Eng E, int n_slots, unsigned seq)
{
    // 16 consecutive slots per thread and pass (two 16-byte loads of the u16 leaf_node entries), a shuffle scan inside the
    // wave, 16 wave totals through LDS: rows come out in slot order.  A slot counts when it holds a pending leaf; a game
    // that is over has none (its last leaves were consumed by k_end_search before k_play_move ended it).
    __shared__ int wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int start = 0; start < n_slots; start += 16 * 1024) {
        const int s0 = start + tid * 16;
        uint32_t w[8] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
        if (s0 + 16 <= n_slots) {
            const uint4 a = *reinterpret_cast<const uint4 *>(E.leaf_node + s0), b = *reinterpret_cast<const uint4 *>(E.leaf_node + s0 + 8);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        } else {
            for (int i = 0; i < 16; i++)
                if (s0 + i < n_slots) {
                    const uint32_t v = E.leaf_node[s0 + i];
                    w[i >> 1] = (i & 1) ? (w[i >> 1] & 0x0000ffffu) | (v << 16) : (w[i >> 1] & 0xffff0000u) | v;
                }
        }
        uint32_t flags = 0;                                  // bit i: slot s0 + i holds a pending leaf
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t v = (i & 1) ? w[i >> 1] >> 16 : w[i >> 1] & 0xffffu;
            if (v != LEAF_NONE) flags |= 1u << i;
        }
        if (E.ec_on == 1) {
            // a pending leaf the evaluation cache answers (leaf_ec >= 0) needs no row
            uint32_t cached = 0;
            if (s0 + 16 <= n_slots) {
#pragma unroll
                for (int i4 = 0; i4 < 4; i4++) {
                    const int4 c = reinterpret_cast<const int4 *>(E.leaf_ec + s0)[i4];
                    cached |= (c.x >= 0 ? 1u : 0u) << (4 * i4) | (c.y >= 0 ? 2u : 0u) << (4 * i4) | (c.z >= 0 ? 4u : 0u) << (4 * i4) |
                              (c.w >= 0 ? 8u : 0u) << (4 * i4);
                }
            } else {
                for (int i = 0; i < 16; i++)
                    if (s0 + i < n_slots && E.leaf_ec[s0 + i] >= 0) cached |= 1u << i;
            }
            flags &= ~cached;
        }
        uint32_t dups = 0;                                   // bit i: pending, but a lower slot holds the same position
        if (E.dedupe) {
            // the search kernel marked every slot that lost its position to a lower one (dedupe_insert); read the marks and
            // clear them for the next round
            if (s0 + 16 <= n_slots) {
                const uint4 d4 = *reinterpret_cast<const uint4 *>(E.leaf_dup + s0);
                const uint32_t dw[4] = { d4.x, d4.y, d4.z, d4.w };
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if ((dw[i >> 2] >> ((i & 3) * 8)) & 0xffu) dups |= 1u << i;
                if (dups) *reinterpret_cast<uint4 *>(E.leaf_dup + s0) = make_uint4(0, 0, 0, 0);
            } else {
                for (int i = 0; i < 16; i++)
                    if (s0 + i < n_slots && E.leaf_dup[s0 + i]) { dups |= 1u << i; E.leaf_dup[s0 + i] = 0; }
            }
            dups &= flags;
            flags &= ~dups;
        }
        const int mine = __popc(flags);
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int off = base + incl - mine, total = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { const int c = wtot[k]; off += k < wave ? c : 0; total += c; }
        int rows[16];
        int r = off;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool p = (flags >> i) & 1u;
            rows[i] = p ? r : -1;
            if (p) { E.row_src[r] = s0 + i; r++; }
        }
        if (s0 + 16 <= n_slots) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                reinterpret_cast<int4 *>(E.leaf_row + s0)[i] = make_int4(rows[4 * i], rows[4 * i + 1], rows[4 * i + 2], rows[4 * i + 3]);
        } else {
            for (int i = 0; i < 16; i++)
                if (s0 + i < n_slots) E.leaf_row[s0 + i] = rows[i];
        }
        base += total;
        __syncthreads();
        if (dups) {
            // the representative = the slot left in the group's table entry = the lowest slot of the group: its row was
            // written in this pass or an earlier one (before the barrier above; read past the L1 of this CU)
            const unsigned *tab_lo = reinterpret_cast<const unsigned *>(E.dd_tab);
#pragma unroll
            for (int i = 0; i < 16; i++)
                if ((dups >> i) & 1u) {
                    const unsigned rep = __hip_atomic_load(&tab_lo[2 * (size_t)(unsigned)E.leaf_pos[s0 + i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    E.leaf_row[s0 + i] = __hip_atomic_load(&E.leaf_row[rep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
    }
    if (tid == 0) { E.row_count[0] = base; E.row_hist[seq & (ROW_HIST - 1)] = base; }
}

__global__ __launch_bounds__(64) void k_end_search(Eng E, int eval_kind, const void *ev_a, const void *ev_v)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x;
    const GameS gs = load_gs(E.gs + g);
    if (gs.done) return;
    const Tree T = tree_of(E, g);
    const int root = E.root_node[g];
    if (E.vloss) {
        for (int k = 0; k < E.leaf_slots; k++)
            consume_eval<true>(E, g, g * E.leaf_slots + k, L, T, root, eval_kind, ev_a, ev_v, gs.n_plies);
    } else {
        consume_eval<false>(E, g, g, L, T, root, eval_kind, ev_a, ev_v, gs.n_plies);
    }
}

// NumPy float64 add.reduce (pairwise, 8 partial sums; n <= 128)
__device__ double np_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// self_play.py:219-256 for every game: pi from root visits, sample record, np.random.choice on the
// game's private MT19937 stream, make_move, and the next root's move list.
__global__ __launch_bounds__(64) void k_play_move(Eng E)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    GameS gs = load_gs(E.gs + g);
    if (gs.done) return;
    const Tree T = tree_of(E, g);
    const int root = E.root_node[g];
    const int nc = T.nc[root], first = T.first[root];
    if (nc == 0) { gs.done = 1; store_gs(E.gs + g, gs); return; }       // self_play.py:216-217

    for (int j = lane; j < nc; j += 64) L.ibuf[j] = (int)T.N[first + j];
    unpack_to_lds(E.board + (size_t)g * 12, L.bd);
    wave_sync();

    // sample record (self_play.py:234-239)
    const bool store = (gs.side == 1) || !E.opponent_mode;
    if (store) {
        const size_t si = (size_t)g * XQ_MAX_PLIES + gs.n_samples;
        if (lane < 12) E.s_board[si * 12 + lane] = E.board[(size_t)g * 12 + lane];
        for (int j = lane; j < MAXM; j += 64) {
            E.s_moves[si * MAXM + j] = j < nc ? T.mv[first + j] : 0;
            E.s_counts[si * MAXM + j] = j < nc ? (uint16_t)L.ibuf[j] : 0;
        }
        if (lane == 0) { E.s_player[si] = gs.side; E.s_n[si] = (uint8_t)nc; }
    }

    // move probabilities and np.random.choice (serial float64, order matters)
    int pick = -1;
    if (lane == 0) {
        if (E.temperature < 0.01) {                                      // self_play.py:224-227
            int am = 0;
            for (int j = 1; j < nc; j++) if (L.ibuf[j] > L.ibuf[am]) am = j;
            for (int j = 0; j < nc; j++) L.fbuf[j] = 0.0;
            L.fbuf[am] = 1.0;
        } else {                                                         // self_play.py:230-231
            for (int j = 0; j < nc; j++) {
                const int c = L.ibuf[j];
                L.fbuf[j] = (E.pow_table && c < E.pow_n) ? E.pow_table[c] : (double)c;
            }
            const double sum = np_sum(L.fbuf, nc);
            for (int j = 0; j < nc; j++) L.fbuf[j] = L.fbuf[j] / sum;
        }
        // RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, 'right')
        bool nan = false;
        double acc = 0;
        for (int j = 0; j < nc; j++) { nan |= (L.fbuf[j] != L.fbuf[j]); acc += L.fbuf[j]; L.fbuf[j] = acc; }
        if (!nan) {
            const double last = L.fbuf[nc - 1];
            const double u = E.uniforms[(size_t)g * XQ_MAX_PLIES + gs.n_plies];
            int lo = 0, hi = nc;
            while (lo < hi) {
                int mid = lo + ((hi - lo) >> 1);
                if (L.fbuf[mid] / last <= u) lo = mid + 1; else hi = mid;
            }
            pick = lo < nc ? lo : nc - 1;
        }
    }
    pick = uni(pick);
    if (pick < 0) {                      // ValueError: probabilities contain NaN (sims <= leaf_batch)
        gs.error = 1; gs.done = 1;
        store_gs(E.gs + g, gs);
        return;
    }
    const int move = T.mv[first + pick];
    if (E.tree_reuse && lane == 0)       // keep the played child's subtree if it was ever expanded
        E.root_node[g] = (uint16_t)(T.nc[first + pick] != 0 ? first + pick : 0);

    MState st = to_mstate(gs);
    HistGlobal hist{ E.pos_hist + (size_t)g * PATH_CAP, E.chk_hist + (size_t)g * PATH_CAP, gs.n_hist, gs.n_hist, 0ull, 0 };
    MoveResult mr = wave_make_move<true, true>(L.bd, st, move, hist, L.maps, L.cand, L.legal, L.own_sq);
    from_mstate(gs, st);
    if (lane == 0) {
        if (gs.n_hist < PATH_CAP) {
            E.pos_hist[(size_t)g * PATH_CAP + gs.n_hist] = hist.new_key;
            E.chk_hist[(size_t)g * PATH_CAP + gs.n_hist] = (uint8_t)hist.new_chk;
        }
        E.step_reward[(size_t)g * XQ_MAX_PLIES + gs.n_plies] = mr.reward;
        E.t_move[(size_t)g * XQ_MAX_PLIES + gs.n_plies] = (uint16_t)move;
    }
    gs.n_hist += 1;
    gs.n_plies += 1;
    if (store) gs.n_samples += 1;
    if (lane < 12) E.board[(size_t)g * 12 + lane] = pack_dword(L.bd, lane);
    for (int j = lane; j < mr.n_legal; j += 64) E.root_moves[(size_t)g * MAXM + j] = L.legal[j];
    gs.n_root = (uint8_t)mr.n_legal;
    // self_play.py:203,205-208,255-256: stop on done, on MAX_MOVES, or when nothing is legal
    gs.done = (mr.done || gs.n_plies >= E.max_moves || mr.n_legal == 0) ? 1 : 0;
    store_gs(E.gs + g, gs);
    if (E.eval_carry && !gs.done) {
        // The reference rebuilds its tree every ply and so evaluates the new root again, although the position
        // was evaluated during this ply's search: the played child has visits, hence was expanded from a network
        // row for exactly this (board, player).  The evaluator is deterministic and row-independent, so that
        // second evaluation returns the same priors bit for bit: take them from the child's expansion and leave
        // the tree as round 0 would (root.visit_count = first batch size, children unvisited, self_play.py:103-148).
        const int c = first + pick, n = T.nc[c], fc = T.first[c];
        const int j0 = lane, j1 = lane + 64;
        const uint16_t m0 = j0 < n ? T.mv[fc + j0] : (uint16_t)0, m1 = j1 < n ? T.mv[fc + j1] : (uint16_t)0;
        const float p0 = j0 < n ? T.P[fc + j0] : 0.f, p1 = j1 < n ? T.P[fc + j1] : 0.f;
        // (the child's move list is the legal-move list the leaf's make_move produced: it must equal this one)
        const bool same = (j0 >= n || m0 == L.legal[j0]) && (j1 >= n || m1 == L.legal[j1]);
        const bool ok = n != 0 && n == mr.n_legal && __all(same);
        wave_sync();
        if (ok) {
            if (j0 < n) { const int x = 1 + j0; T.N[x] = 0; T.W[x] = 0.0; T.P[x] = p0; T.mv[x] = m0; T.first[x] = 0; T.nc[x] = 0; T.fl[x] = 0; }
            if (j1 < n) { const int x = 1 + j1; T.N[x] = 0; T.W[x] = 0.0; T.P[x] = p1; T.mv[x] = m1; T.first[x] = 0; T.nc[x] = 0; T.fl[x] = 0; }
            if (lane == 0) {
                const int b0 = E.sims < E.leaf_batch ? E.sims : E.leaf_batch;
                T.N[0] = (uint32_t)b0; T.W[0] = 0.0; T.P[0] = 0.f; T.mv[0] = 0; T.first[0] = 1; T.nc[0] = (uint8_t)n; T.fl[0] = 0;
                E.n_nodes[g] = (uint32_t)(1 + n); E.root_node[g] = 0;
            }
        }
        if (lane == 0) {
            E.root_ready[g] = ok ? 1 : 0;
            if (!ok) atomicAdd(E.roots_not_ready, 1);
        }
    }
}

// self_play.py:259-310
__device__ void finalize_game(const Eng &E, int g, const GameS &gs)
{
    const int lane = XQ_LANE;
    const int winner = gs.winner == WINNER_NONE ? 0 : gs.winner;
    const int len = gs.n_samples;
    for (int i = lane; i < len; i += 64) {
        const size_t si = (size_t)g * XQ_MAX_PLIES + i;
        const int player = E.s_player[si];
        double fr;
        if (winner == 0) fr = len >= 60 ? (player == 1 ? -0.15 : 0.05) : (player == 1 ? -0.1 : 0.1);
        else if (winner == player) fr = 1.0 + (len <= 30 ? 0.5 : len <= 50 ? 0.3 : len <= 70 ? 0.1 : 0.0);
        else fr = len >= 60 ? -1.2 : -1.0;
        const double imm = i < gs.n_plies ? E.step_reward[si] : 0.0;   // indexed by sample index (A14 note)
        const double scaled = imm * 0.01;
        E.s_z[si] = fr + scaled;
    }
}

__global__ __launch_bounds__(64) void k_finalize(Eng E)
{
    const int g = blockIdx.x;
    finalize_game(E, g, load_gs(E.gs + g));
}

// the samples of the game in slot g as fixed-size records rec[0 .. 69]
__device__ void pack_game(const Eng &E, int g, const GameS &gs, xq_sample_record *rec)
{
    const int lane = XQ_LANE;
    for (int i = 0; i < XQ_MAX_PLIES; i++) {
        const size_t si = (size_t)g * XQ_MAX_PLIES + i;
        xq_sample_record *r = rec + i;
        const bool valid = i < gs.n_samples;
        if (lane < 12) r->board[lane] = valid ? E.s_board[si * 12 + lane] : 0u;
        for (int j = lane; j < MAXM; j += 64) {
            r->moves[j] = valid ? E.s_moves[si * MAXM + j] : 0;
            r->counts[j] = valid ? E.s_counts[si * MAXM + j] : 0;
        }
        if (lane == 0) {
            r->z = valid ? E.s_z[si] : 0.0;
            r->player = valid ? E.s_player[si] : 0;
            r->n_moves = valid ? E.s_n[si] : 0;
            r->valid = valid ? 1 : 0;
            r->pad = 0; r->pad2 = 0;
            r->chosen = valid ? E.t_move[si] : 0;
        }
    }
}

__global__ __launch_bounds__(64) void k_pack_samples(Eng E, xq_sample_record *rec)
{
    const int g = blockIdx.x;
    pack_game(E, g, load_gs(E.gs + g), rec + (size_t)g * XQ_MAX_PLIES);
}

// ------------------------------------------------------------------------------------------
// Refill: the reference's pool hands a worker its next game the moment one ends (imap_unordered,
// self_play.py:404-408).  Here a finished game's slot is retired into its own record block (z table + packed
// samples + outcome, all by game id) and restarted on the next unplayed game id; a game's result depends only
// on its seed (its private MT19937 doubles travel with the id), never on the slot or the ply it started at.
// ------------------------------------------------------------------------------------------
struct Refill {
    const double *uni_all;     // [total][70]
    int32_t *slot_game;        // [G]  game id in the slot, -1 = retired
    int32_t *next_game;        // next unplayed game id
    int32_t *active;           // out: slots still playing after this step
    GameS *out_gs;             // [total]
    xq_sample_record *records; // [total][70]
    int total;
};

__global__ __launch_bounds__(64) void k_refill(Eng E, Refill R)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    int id = R.slot_game[g];
    if (id < 0) return;
    const GameS gs = load_gs(E.gs + g);
    if (!gs.done) { if (lane == 0) atomicAdd(R.active, 1); return; }
    finalize_game(E, g, gs);
    mem_fence_wave();
    wave_sync();
    pack_game(E, g, gs, R.records + (size_t)id * XQ_MAX_PLIES);
    if (lane == 0) R.out_gs[id] = gs;
    int nid = 0;
    if (lane == 0) nid = atomicAdd(R.next_game, 1);
    nid = uni(nid);
    if (nid >= R.total) { if (lane == 0) R.slot_game[g] = -1; return; }
    for (int j = lane; j < XQ_MAX_PLIES; j += 64)
        E.uniforms[(size_t)g * XQ_MAX_PLIES + j] = R.uni_all[(size_t)nid * XQ_MAX_PLIES + j];
    new_game(E, g, L);
    if (lane == 0) { R.slot_game[g] = nid; atomicAdd(R.active, 1); }
}

// root children of every game (moves, visit counts, priors), zero-padded to 128
__global__ __launch_bounds__(64) void k_root_children(Eng E, uint16_t *moves, int32_t *visits, float *priors, int32_t *n_child)
{
    const int g = blockIdx.x, lane = XQ_LANE;
    const Tree T = tree_of(E, g);
    const int root = E.root_node[g];
    const int nc = T.nc[root], first = T.first[root];
    for (int j = lane; j < MAXM; j += 64) {
        const bool ok = j < nc;
        if (moves) moves[(size_t)g * MAXM + j] = ok ? T.mv[first + j] : (uint16_t)0;
        if (visits) visits[(size_t)g * MAXM + j] = ok ? (int32_t)T.N[first + j] : 0;
        if (priors) priors[(size_t)g * MAXM + j] = ok ? T.P[first + j] : 0.f;
    }
    if (lane == 0 && n_child) n_child[g] = nc;
}

__global__ void k_count_active(const GameS *gs, int G, int *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int a = (i < G && !gs[i].done) ? 1 : 0;
    unsigned long long m = __ballot(a);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __builtin_popcountll(m));
}

// ---- rules entry points on caller boards -------------------------------------------------
__global__ __launch_bounds__(64) void k_rules_legal(int n, const int8_t *boards, const int32_t *player,
                                                    const int32_t *rk, const int32_t *bk,
                                                    uint16_t *moves, int32_t *counts)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) L.bd[s] = s < 90 ? boards[(size_t)g * 90 + s] : 0;
    wave_sync();
    BoardView v = load_view(L.bd);
    const uint32_t kab = build_attack_maps(L.maps, v, -player[g]);
    const int cnt = wave_movegen(L.bd, v, L.maps, kab, player[g], rk[g], bk[g], L.cand, L.legal, L.own_sq);
    for (int j = lane; j < cnt; j += 64) moves[(size_t)g * MAXM + j] = L.legal[j];
    if (lane == 0) counts[g] = cnt;
}

// a8 (chess_env.py:550-596): the (attacker, victim) pairs `side` threatens.  In the executed reference a pair is
// every LEGAL move of `side` (generated with current_player = side: wave_movegen's contract) that captures an
// enemy piece other than the king: _is_protected can never answer True, because _get_piece_moves drops every
// move onto a square the mover's own side holds (:116) — pinned by tests/golden/rules_extra.json against the
// literal oracle restatement.  Order = legal-move order; compaction by ballot prefix.
__global__ __launch_bounds__(64) void k_rules_threats(int n, const int8_t *boards, const int32_t *side,
                                                      const int32_t *rk, const int32_t *bk,
                                                      uint16_t *pairs, int32_t *counts)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) L.bd[s] = s < 90 ? boards[(size_t)g * 90 + s] : 0;
    wave_sync();
    BoardView v = load_view(L.bd);
    const int p = side[g];
    const uint32_t kab = build_attack_maps(L.maps, v, -p);
    const int cnt = wave_movegen(L.bd, v, L.maps, kab, p, rk[g], bk[g], L.cand, L.legal, L.own_sq);
    int base = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        bool keep = false;
        int mv = 0;
        if (j < cnt) {
            mv = L.legal[j];
            const int t = L.bd[mv % 90];
            keep = t * p < 0 && (t < 0 ? -t : t) != 1;              // an enemy piece, not the king (code 1)
        }
        const unsigned long long m = __ballot(keep);
        if (keep) pairs[(size_t)g * MAXM + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)mv;
        base += __popcll(m);
    }
    if (lane == 0) counts[g] = base;
}

// _get_position_hash (chess_env.py:497-504) as the 64-bit key the engine's repetition rule compares
__global__ __launch_bounds__(64) void k_rules_position_key(int n, const int8_t *boards, const int32_t *player, uint64_t *keys)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) L.bd[s] = s < 90 ? boards[(size_t)g * 90 + s] : 0;
    wave_sync();
    const uint64_t k = position_key(L.bd, player[g] == 1 ? 0 : 1);
    if (lane == 0) keys[g] = k;
}

__global__ __launch_bounds__(64) void k_rules_query(int n, const int8_t *boards, const int32_t *player,
                                                    const int32_t *rk, const int32_t *bk,
                                                    int32_t *chk_red, int32_t *chk_black, int32_t *facing)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) L.bd[s] = s < 90 ? boards[(size_t)g * 90 + s] : 0;
    wave_sync();
    BoardView v = load_view(L.bd);
    uint32_t kab = build_attack_maps(L.maps, v, -1);
    const bool cr = in_check(L.maps, kab, rk[g], player[g]);
    const bool f = kings_facing(L.maps, rk[g], bk[g]);
    kab = build_attack_maps(L.maps, v, 1);
    const bool cb = in_check(L.maps, kab, bk[g], player[g]);
    if (lane == 0) { chk_red[g] = cr; chk_black[g] = cb; facing[g] = f; }
}

__global__ __launch_bounds__(64) void k_rules_make_move(int n, int8_t *boards, int32_t *state, const int32_t *move,
                                                        const uint64_t *pos_hist, const int32_t *n_hist,
                                                        const uint8_t *chk_hist, const int32_t *n_chk, int stride,
                                                        double *reward, int32_t *done, int32_t *is_check,
                                                        uint64_t *key_out, uint16_t *next_moves, int32_t *next_count)
{
    __shared__ WaveLds L;
    const int g = blockIdx.x, lane = XQ_LANE;
    for (int s = lane; s < 96; s += 64) L.bd[s] = s < 90 ? boards[(size_t)g * 90 + s] : 0;
    wave_sync();
    int32_t *sw = state + (size_t)g * XQ_STATE_WORDS;
    MState st;
    st.side = sw[XQ_S_PLAYER]; st.move_count = sw[XQ_S_MOVE_COUNT]; st.winner = sw[XQ_S_WINNER];
    st.rk = sw[XQ_S_RED_KING]; st.bk = sw[XQ_S_BLACK_KING]; st.nocap = sw[XQ_S_NO_CAPTURE];
    st.cchk = sw[XQ_S_CONSEC_CHECKS]; st.reason = sw[XQ_S_REASON]; st.reason_side = sw[XQ_S_REASON_SIDE];
    st.reason_count = sw[XQ_S_REASON_COUNT];
    HistGlobal hist{ pos_hist + (size_t)g * stride, chk_hist + (size_t)g * stride, n_hist[g], n_chk[g], 0ull, 0 };
    MoveResult mr = wave_make_move<true, true>(L.bd, st, move[g], hist, L.maps, L.cand, L.legal, L.own_sq);
    wave_sync();
    for (int s = lane; s < 90; s += 64) boards[(size_t)g * 90 + s] = L.bd[s];
    for (int j = lane; j < mr.n_legal; j += 64) next_moves[(size_t)g * MAXM + j] = L.legal[j];
    if (lane == 0) {
        sw[XQ_S_PLAYER] = st.side; sw[XQ_S_MOVE_COUNT] = st.move_count; sw[XQ_S_WINNER] = st.winner;
        sw[XQ_S_RED_KING] = st.rk; sw[XQ_S_BLACK_KING] = st.bk; sw[XQ_S_NO_CAPTURE] = st.nocap;
        sw[XQ_S_CONSEC_CHECKS] = st.cchk; sw[XQ_S_REASON] = st.reason; sw[XQ_S_REASON_SIDE] = st.reason_side;
        sw[XQ_S_REASON_COUNT] = st.reason_count;
        reward[g] = mr.reward; done[g] = mr.done; is_check[g] = mr.is_check; key_out[g] = hist.new_key;
        next_count[g] = mr.n_legal;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(XQ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

extern "C" const char *xq_last_error(void) { return g_err.c_str(); }

extern "C" int xq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int xq_device_ok(int device)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

static bool g_crc_ready[64] = { false };

static int ensure_crc_table(int device)
{
    if (device < 0 || device >= 64) return fail(XQ_E_INVALID, "device ordinal out of range");
    if (g_crc_ready[device]) return 0;
    uint32_t tab[256];
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        tab[i] = c;
    }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_crc_table), tab, sizeof tab));
    g_crc_ready[device] = true;
    return 0;
}

// RAII-less device buffer helper: every rules call stages through temporary buffers
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

static int need_gpu()
{
    if (xq_device_count() <= 0) return fail(XQ_E_NOGPU, "no HIP device visible (the HIP path has no CPU fallback)");
    return 0;
}

extern "C" int xq_rules_legal_moves(int n, const int8_t *boards, const int32_t *player, const int32_t *rk,
                                    const int32_t *bk, uint16_t *moves, int32_t *counts)
{
    if (n <= 0 || !boards || !player || !rk || !bk || !moves || !counts) return fail(XQ_E_INVALID, "bad argument");
    if (int rc = need_gpu()) return rc;
    DevBuf dB, dP, dR, dK, dM, dC;
    if (dB.alloc((size_t)n * 90) || dP.alloc((size_t)n * 4) || dR.alloc((size_t)n * 4) || dK.alloc((size_t)n * 4) ||
        dM.alloc((size_t)n * MAXM * 2) || dC.alloc((size_t)n * 4))
        return fail(XQ_E_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpy(dB.p, boards, (size_t)n * 90, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dP.p, player, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dR.p, rk, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dK.p, bk, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dM.p, 0, (size_t)n * MAXM * 2));
    hipLaunchKernelGGL(k_rules_legal, dim3(n), dim3(64), 0, 0, n, dB.as<int8_t>(), dP.as<int32_t>(), dR.as<int32_t>(),
                       dK.as<int32_t>(), dM.as<uint16_t>(), dC.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(moves, dM.p, (size_t)n * MAXM * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts, dC.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int xq_rules_threatened_pieces(int n, const int8_t *boards, const int32_t *side, const int32_t *rk,
                                         const int32_t *bk, uint16_t *pairs, int32_t *counts)
{
    if (n <= 0 || !boards || !side || !rk || !bk || !pairs || !counts) return fail(XQ_E_INVALID, "bad argument");
    if (int rc = need_gpu()) return rc;
    DevBuf dB, dP, dR, dK, dM, dC;
    if (dB.alloc((size_t)n * 90) || dP.alloc((size_t)n * 4) || dR.alloc((size_t)n * 4) || dK.alloc((size_t)n * 4) ||
        dM.alloc((size_t)n * MAXM * 2) || dC.alloc((size_t)n * 4))
        return fail(XQ_E_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpy(dB.p, boards, (size_t)n * 90, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dP.p, side, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dR.p, rk, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dK.p, bk, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dM.p, 0, (size_t)n * MAXM * 2));
    hipLaunchKernelGGL(k_rules_threats, dim3(n), dim3(64), 0, 0, n, dB.as<int8_t>(), dP.as<int32_t>(), dR.as<int32_t>(),
                       dK.as<int32_t>(), dM.as<uint16_t>(), dC.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(pairs, dM.p, (size_t)n * MAXM * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts, dC.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int xq_rules_position_key(int n, const int8_t *boards, const int32_t *player, uint64_t *keys)
{
    if (n <= 0 || !boards || !player || !keys) return fail(XQ_E_INVALID, "bad argument");
    if (int rc = need_gpu()) return rc;
    DevBuf dB, dP, dK;
    if (dB.alloc((size_t)n * 90) || dP.alloc((size_t)n * 4) || dK.alloc((size_t)n * 8)) return fail(XQ_E_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpy(dB.p, boards, (size_t)n * 90, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dP.p, player, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rules_position_key, dim3(n), dim3(64), 0, 0, n, dB.as<int8_t>(), dP.as<int32_t>(), dK.as<uint64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(keys, dK.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int xq_rules_query(int n, const int8_t *boards, const int32_t *player, const int32_t *rk, const int32_t *bk,
                              int32_t *chk_red, int32_t *chk_black, int32_t *facing)
{
    if (n <= 0 || !boards || !player || !rk || !bk || !chk_red || !chk_black || !facing)
        return fail(XQ_E_INVALID, "bad argument");
    if (int rc = need_gpu()) return rc;
    DevBuf dB, dP, dR, dK, d1, d2, d3;
    if (dB.alloc((size_t)n * 90) || dP.alloc((size_t)n * 4) || dR.alloc((size_t)n * 4) || dK.alloc((size_t)n * 4) ||
        d1.alloc((size_t)n * 4) || d2.alloc((size_t)n * 4) || d3.alloc((size_t)n * 4))
        return fail(XQ_E_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpy(dB.p, boards, (size_t)n * 90, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dP.p, player, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dR.p, rk, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dK.p, bk, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rules_query, dim3(n), dim3(64), 0, 0, n, dB.as<int8_t>(), dP.as<int32_t>(), dR.as<int32_t>(),
                       dK.as<int32_t>(), d1.as<int32_t>(), d2.as<int32_t>(), d3.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(chk_red, d1.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(chk_black, d2.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(facing, d3.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int xq_rules_make_move(int n, int8_t *boards, int32_t *state, const int32_t *move,
                                  const uint64_t *pos_hist, const int32_t *n_hist, const uint8_t *chk_hist,
                                  const int32_t *n_chk, int stride, double *reward, int32_t *done,
                                  int32_t *is_check, uint64_t *key_out, uint16_t *next_moves, int32_t *next_count)
{
    if (n <= 0 || !boards || !state || !move || !n_hist || !n_chk || !reward || !done || !is_check || !key_out ||
        !next_moves || !next_count || stride < 0)
        return fail(XQ_E_INVALID, "bad argument");
    for (int i = 0; i < n; i++)
        if (n_hist[i] < 0 || n_hist[i] > stride || n_chk[i] < 0 || n_chk[i] > stride)
            return fail(XQ_E_INVALID, "history length exceeds hist_stride");
    if (stride > 0 && (!pos_hist || !chk_hist)) return fail(XQ_E_INVALID, "history pointers missing");
    if (int rc = need_gpu()) return rc;
    const size_t hs = (size_t)n * (stride ? stride : 1);
    DevBuf dB, dS, dM, dPH, dNH, dCH, dNC, dRw, dDn, dIc, dKy, dNm, dNn;
    if (dB.alloc((size_t)n * 90) || dS.alloc((size_t)n * XQ_STATE_WORDS * 4) || dM.alloc((size_t)n * 4) ||
        dPH.alloc(hs * 8) || dNH.alloc((size_t)n * 4) || dCH.alloc(hs) || dNC.alloc((size_t)n * 4) ||
        dRw.alloc((size_t)n * 8) || dDn.alloc((size_t)n * 4) || dIc.alloc((size_t)n * 4) || dKy.alloc((size_t)n * 8) ||
        dNm.alloc((size_t)n * MAXM * 2) || dNn.alloc((size_t)n * 4))
        return fail(XQ_E_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpy(dB.p, boards, (size_t)n * 90, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dS.p, state, (size_t)n * XQ_STATE_WORDS * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dM.p, move, (size_t)n * 4, hipMemcpyHostToDevice));
    if (stride > 0) {
        HIPCHK(hipMemcpy(dPH.p, pos_hist, hs * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dCH.p, chk_hist, hs, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(dNH.p, n_hist, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dNC.p, n_chk, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dNm.p, 0, (size_t)n * MAXM * 2));
    hipLaunchKernelGGL(k_rules_make_move, dim3(n), dim3(64), 0, 0, n, dB.as<int8_t>(), dS.as<int32_t>(), dM.as<int32_t>(),
                       dPH.as<uint64_t>(), dNH.as<int32_t>(), dCH.as<uint8_t>(), dNC.as<int32_t>(), stride,
                       dRw.as<double>(), dDn.as<int32_t>(), dIc.as<int32_t>(), dKy.as<uint64_t>(), dNm.as<uint16_t>(),
                       dNn.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(boards, dB.p, (size_t)n * 90, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(state, dS.p, (size_t)n * XQ_STATE_WORDS * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(reward, dRw.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(done, dDn.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(is_check, dIc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(key_out, dKy.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(next_moves, dNm.p, (size_t)n * MAXM * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(next_count, dNn.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ---- engine ------------------------------------------------------------------------------
struct xq_engine {
    xq_config cfg;
    Eng E;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    double *pow_table_dev = nullptr;
    int pow_cap = 0;
    int *active_dev = nullptr;
    int32_t *active_host = nullptr;      // pinned: results of xq_engine_active_games_post
    hipEvent_t active_ev = nullptr;
    unsigned active_posted = 0;
    int8_t *stage_boards = nullptr;      // [G][90] staging for set_roots / read_leaves
    int32_t *stage_state = nullptr;      // [G][XQ_STATE_WORDS]
    uint16_t *stage_rmoves = nullptr;    // [G][128] staging for read_root_visits / read_root_priors
    int32_t *stage_rvisits = nullptr;    // [G][128]
    float *stage_rpriors = nullptr;      // [G][128]
    int32_t *stage_rn = nullptr;         // [G]
    // refill mode (xq_engine_refill_*)
    double *uni_all = nullptr; int32_t *slot_game = nullptr, *next_game = nullptr; GameS *out_gs = nullptr;
    int refill_total = 0, refill_cap = 0;     // games of the running session / games the session buffers hold
    unsigned row_seq = 0;                     // search rounds launched with row compaction (index into row_hist)
    bool row_map_fetched = false;             // xq_engine_row_map since row compaction was last switched: the evaluator knows the layout
    unsigned dd_tag = 0;                      // round tag of the leaf dedupe table (never 0: the cleared table's tag)
    unsigned ec_epoch = 16, ec_seq = 0;       // evaluation cache: ply counter (+2 at every new set of roots), launch counter
    // profiling
    int prof = 0;                             // 0 = off, N = HIP events around every N-th k_search_round (and every k_play_move)
    unsigned prof_seen = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_search, ev_play;
    size_t ev_search_used = 0, ev_play_used = 0;
    double search_ms = 0, play_ms = 0;
    int64_t search_n = 0, play_n = 0;
};

template <class T> static int dalloc(xq_engine *e, T *&p, size_t count)
{
    void *q = nullptr;
    if (hipMalloc(&q, count * sizeof(T) + 64) != hipSuccess) return -1;
    if (hipMemset(q, 0, count * sizeof(T) + 64) != hipSuccess) return -1;
    e->allocs.push_back(q);
    p = reinterpret_cast<T *>(q);
    return 0;
}

// leaf dedupe table: at least two entries per pending-leaf slot, a power of two
static int alloc_dedupe(xq_engine *e, size_t slots)
{
    size_t n = 1024;
    while (n < 2 * slots) n <<= 1;
    int bad = dalloc(e, e->E.dd_tab, n);
    bad |= dalloc(e, e->E.leaf_pos, slots); bad |= dalloc(e, e->E.leaf_dup, slots + 16);
    e->E.dd_mask = (int)(n - 1);
    return bad;
}

extern "C" int xq_engine_create(const xq_config *cfg, xq_engine **out)
{
    if (!cfg || !out) return fail(XQ_E_INVALID, "null argument");
    if (cfg->n_games <= 0 || cfg->sims <= 0 || cfg->leaf_batch <= 0 || cfg->leaf_batch > 255 || cfg->max_moves <= 0)
        return fail(XQ_E_INVALID, "n_games, sims, leaf_batch, max_moves must be positive (leaf_batch <= 255)");
    if (int rc = need_gpu()) return rc;
    HIPCHK(hipSetDevice(cfg->device));
    if (!xq_device_ok(cfg->device)) return fail(XQ_E_NOGPU, "device is not gfx950: this library carries gfx950 code only");
    if (int rc = ensure_crc_table(cfg->device)) return rc;
    const int nrounds = (cfg->sims + cfg->leaf_batch - 1) / cfg->leaf_batch;
    if (nrounds > 64) return fail(XQ_E_INVALID, "sims / leaf_batch too large (at most 64 rounds per move: one lane per path level)");
    const long ncap_l = 1 + (long)nrounds * MAXM;
    if (ncap_l > 65535) return fail(XQ_E_INVALID, "node arena exceeds 16-bit indices");
    xq_engine *e = new xq_engine();
    e->cfg = *cfg;
    Eng &E = e->E;
    memset(&E, 0, sizeof E);
    E.G = cfg->n_games; E.ncap = (int)((ncap_l + 63) / 64 * 64); E.sims = cfg->sims; E.leaf_batch = cfg->leaf_batch;
    E.nrounds = nrounds; E.max_moves = cfg->max_moves; E.opponent_mode = cfg->opponent_mode;
    E.want_check = nrounds >= 12 ? 1 : 0;       // check_history only matters once 12 plies fit a path
    E.temperature = cfg->temperature;
    E.leaf_slots = 1;
    const size_t G = (size_t)E.G, NC = (size_t)E.ncap;
    int bad = 0;
    bad |= dalloc(e, E.board, G * 12); bad |= dalloc(e, E.gs, G); bad |= dalloc(e, E.pos_hist, G * PATH_CAP);
    bad |= dalloc(e, E.chk_hist, G * PATH_CAP); bad |= dalloc(e, E.root_moves, G * MAXM);
    bad |= dalloc(e, E.uniforms, G * XQ_MAX_PLIES);
    bad |= dalloc(e, E.nN, G * NC); bad |= dalloc(e, E.nW, G * NC); bad |= dalloc(e, E.nP, G * NC);
    bad |= dalloc(e, E.nMove, G * NC); bad |= dalloc(e, E.nFirst, G * NC); bad |= dalloc(e, E.nNc, G * NC);
    bad |= dalloc(e, E.nFlags, G * NC); bad |= dalloc(e, E.nVl, G * NC); bad |= dalloc(e, E.n_nodes, G);
    bad |= dalloc(e, E.root_node, G);
    bad |= dalloc(e, E.leaf_node, G); bad |= dalloc(e, E.leaf_mult, G); bad |= dalloc(e, E.leaf_n, G);
    bad |= dalloc(e, E.leaf_depth, G); bad |= dalloc(e, E.leaf_moves, G * MAXM); bad |= dalloc(e, E.leaf_path, G * PATH_CAP);
    bad |= dalloc(e, E.leaf_board, G * 12); bad |= dalloc(e, E.leaf_side, G);
    bad |= dalloc(e, E.priors, G * MAXM); bad |= dalloc(e, E.values, G);
    bad |= dalloc(e, E.leaf_row, G); bad |= dalloc(e, E.row_src, G); bad |= dalloc(e, E.row_count, (size_t)1);
    bad |= dalloc(e, E.row_hist, (size_t)ROW_HIST); bad |= alloc_dedupe(e, G);
    bad |= dalloc(e, E.s_board, G * XQ_MAX_PLIES * 12); bad |= dalloc(e, E.s_player, G * XQ_MAX_PLIES);
    bad |= dalloc(e, E.s_n, G * XQ_MAX_PLIES); bad |= dalloc(e, E.s_moves, G * XQ_MAX_PLIES * MAXM);
    bad |= dalloc(e, E.s_counts, G * XQ_MAX_PLIES * MAXM); bad |= dalloc(e, E.s_z, G * XQ_MAX_PLIES);
    bad |= dalloc(e, E.step_reward, G * XQ_MAX_PLIES); bad |= dalloc(e, E.t_move, G * XQ_MAX_PLIES);
    bad |= dalloc(e, e->active_dev, 1); bad |= dalloc(e, e->stage_boards, G * 90);
    bad |= dalloc(e, e->stage_state, G * XQ_STATE_WORDS);
    bad |= dalloc(e, e->stage_rmoves, G * MAXM); bad |= dalloc(e, e->stage_rvisits, G * MAXM);
    bad |= dalloc(e, e->stage_rpriors, G * MAXM); bad |= dalloc(e, e->stage_rn, G);
    if (bad) {
        xq_engine_destroy(e);
        return fail(XQ_E_HIP, "hipMalloc failed while sizing the engine");
    }
    *out = e;
    return 0;
}

extern "C" void xq_engine_destroy(xq_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->active_host) (void)hipHostFree(e->active_host);
    if (e->active_ev) (void)hipEventDestroy(e->active_ev);
    for (auto &pr : e->ev_search) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : e->ev_play) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    delete e;
}

extern "C" int xq_engine_set_stream(xq_engine *e, void *s)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    e->stream = reinterpret_cast<hipStream_t>(s);
    return 0;
}

extern "C" int xq_engine_set_pow_table(xq_engine *e, const double *t, int n)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (!t || n <= 0) { e->E.pow_table = nullptr; e->E.pow_n = 0; return 0; }
    HIPCHK(hipSetDevice(e->cfg.device));
    if (n > e->pow_cap) {                    // one persistent buffer, regrown only when needed
        double *p = nullptr;
        const int cap = n > e->E.sims + 1 ? n : e->E.sims + 1;
        if (dalloc(e, p, (size_t)cap)) return fail(XQ_E_HIP, "hipMalloc failed");
        e->pow_table_dev = p;
        e->pow_cap = cap;
    }
    HIPCHK(hipMemcpyAsync(e->pow_table_dev, t, (size_t)n * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->E.pow_table = e->pow_table_dev; e->E.pow_n = n;
    return 0;
}

// np.random.seed(int) + 70 x random_sample() per game (MT19937, 53-bit doubles)
static void mt_uniforms(uint32_t seed, double *out, int count)
{
    static thread_local uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    auto next = [&]() -> uint32_t {
        if (idx >= 624) {
            int i;
            for (i = 0; i < 624 - 397; i++) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
                mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; i < 623; i++) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
                mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    };
    for (int k = 0; k < count; k++) {
        uint32_t a = next() >> 5, b = next() >> 6;
        out[k] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
}

// the MT19937 streams of many games on the host cores (each game's stream is independent): 16,384 games take
// 17-25 ms on one core, 1 % of a bench step during which the GPU has nothing to do
static void mt_uniforms_many(const uint32_t *seeds, size_t n, double *out)
{
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? (hw < 16 ? hw : 16) : 4;
    if (n < 512) nt = 1;
    if (nt <= 1) {
        for (size_t g = 0; g < n; g++) mt_uniforms(seeds[g], out + g * XQ_MAX_PLIES, XQ_MAX_PLIES);
        return;
    }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++)
        th.emplace_back([=]() {
            for (size_t g = n * t / nt; g < n * (t + 1) / nt; g++) mt_uniforms(seeds[g], out + g * XQ_MAX_PLIES, XQ_MAX_PLIES);
        });
    for (auto &x : th) x.join();
}

extern "C" int xq_engine_new_games(xq_engine *e, const uint32_t *seeds)
{
    if (!e || !seeds) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    std::vector<double> u(G * XQ_MAX_PLIES);
    mt_uniforms_many(seeds, G, u.data());
    HIPCHK(hipMemcpyAsync(e->E.uniforms, u.data(), u.size() * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));          // u is a local: copy must finish
    if (e->E.eval_carry) HIPCHK(hipMemsetAsync(e->E.roots_not_ready, 0, 4, e->stream));
    e->ec_epoch += 2;                                                    // new games, possibly new weights: nothing cached survives
    e->active_posted = 0;                                                // (a count posted for the last games says nothing about these)
    hipLaunchKernelGGL(k_new_games, dim3(e->E.G), dim3(64), 0, e->stream, e->E);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xq_engine_set_temperature(xq_engine *e, double temperature, const double *table, int n)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    e->E.temperature = temperature;
    e->cfg.temperature = temperature;
    return xq_engine_set_pow_table(e, table, n);
}

extern "C" int xq_engine_set_root_noise(xq_engine *e, double alpha, double epsilon, uint64_t seed)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (epsilon < 0.0 || epsilon > 1.0 || (epsilon > 0.0 && alpha <= 0.0)) return fail(XQ_E_INVALID, "bad alpha / epsilon");
    if (epsilon > 0.0 && e->E.eval_carry) return fail(XQ_E_INVALID, "root noise cannot be combined with root evaluation carry-over");
    e->E.noise_alpha = alpha; e->E.noise_eps = epsilon; e->E.noise_seed = seed;
    return 0;
}

extern "C" int xq_engine_read_root_priors(xq_engine *e, float *priors /*[G][128]*/)
{
    if (!e || !priors) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    hipLaunchKernelGGL(k_root_children, dim3(e->E.G), dim3(64), 0, e->stream, e->E, (uint16_t *)nullptr, (int32_t *)nullptr,
                       e->stage_rpriors, (int32_t *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(priors, e->stage_rpriors, G * MAXM * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// node arena for the extensions: one ply's worst case is nrounds * leaf_slots expansions of <= 128
// children; tree reuse keeps a whole game's.  16-bit node indices cap it at 65,472 nodes (1.5 MB per
// game); a leaf whose children no longer fit stays unexpanded (it is evaluated again when reached).
static int grow_nodes(xq_engine *e)
{
    Eng &E = e->E;
    long want = 1 + (long)E.nrounds * E.leaf_slots * MAXM;
    if (E.tree_reuse) want = 1 + (want - 1) * (long)(E.max_moves < XQ_MAX_PLIES ? E.max_moves : XQ_MAX_PLIES);
    if (want > 65472) want = 65472;
    want = (want + 63) / 64 * 64;
    if (want <= E.ncap) return 0;
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t G = (size_t)E.G, NC = (size_t)want;
    int bad = 0;
    bad |= dalloc(e, E.nN, G * NC); bad |= dalloc(e, E.nW, G * NC); bad |= dalloc(e, E.nP, G * NC);
    bad |= dalloc(e, E.nMove, G * NC); bad |= dalloc(e, E.nFirst, G * NC); bad |= dalloc(e, E.nNc, G * NC);
    bad |= dalloc(e, E.nFlags, G * NC); bad |= dalloc(e, E.nVl, G * NC);
    if (bad) return fail(XQ_E_HIP, "hipMalloc failed while sizing the node arena");
    E.ncap = (int)want;
    return 0;
}

// opt-in extension (no counterpart in the reference): keep the played child's subtree as the next
// ply's tree.  Re-sizes the node arena for a whole game's expansions (16-bit node indices: 65,472
// nodes = 1.4 MB per game); call before xq_engine_new_games / xq_engine_set_roots.
extern "C" int xq_engine_set_tree_reuse(xq_engine *e, int enable)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    Eng &E = e->E;
    if (enable && E.eval_carry) return fail(XQ_E_INVALID, "tree reuse cannot be combined with root evaluation carry-over");
    E.tree_reuse = enable ? 1 : 0;
    if (int rc = grow_nodes(e)) return rc;
    HIPCHK(hipMemsetAsync(E.root_node, 0, (size_t)E.G * 2, e->stream));
    HIPCHK(hipMemsetAsync(E.n_nodes, 0, (size_t)E.G * 4, e->stream));
    return 0;
}

// opt-in extension (no counterpart in the reference, whose rounds of 8 simulations all reach the same
// leaf): virtual loss.  Every simulation of a round counts as a pending loss along its path, so the
// round's simulations spread over up to leaf_batch distinct leaves, each evaluated once.  The
// pending-leaf arrays, xq_engine_priors_ptr / values_ptr and the evaluator's rows (input planes,
// logits, values) then hold leaf_batch slots per game: row = game * leaf_batch + slot.
// Call before xq_engine_new_games / xq_engine_set_roots.
extern "C" int xq_engine_set_virtual_loss(xq_engine *e, int enable)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    Eng &E = e->E;
    // every refusal comes before any state changes (an engine that answered XQ_E_INVALID is left as it was)
    const int K = enable ? E.leaf_batch : 1;
    if (K > 64) return fail(XQ_E_INVALID, "virtual loss needs leaf_batch <= 64 (one lane per pending slot)");
    if (enable && E.eval_carry) return fail(XQ_E_INVALID, "virtual loss cannot be combined with root evaluation carry-over");
    HIPCHK(hipSetDevice(e->cfg.device));
    if (K != E.leaf_slots) {
        HIPCHK(hipStreamSynchronize(e->stream));
        const size_t S = (size_t)E.G * (size_t)K;
        int bad = 0;
        bad |= dalloc(e, E.leaf_node, S); bad |= dalloc(e, E.leaf_mult, S); bad |= dalloc(e, E.leaf_n, S);
        bad |= dalloc(e, E.leaf_depth, S); bad |= dalloc(e, E.leaf_moves, S * MAXM); bad |= dalloc(e, E.leaf_path, S * PATH_CAP);
        bad |= dalloc(e, E.leaf_board, S * 12); bad |= dalloc(e, E.leaf_side, S);
        bad |= dalloc(e, E.priors, S * MAXM); bad |= dalloc(e, E.values, S);
        bad |= dalloc(e, E.leaf_row, S); bad |= dalloc(e, E.row_src, S); bad |= alloc_dedupe(e, S);
        if (bad) return fail(XQ_E_HIP, "hipMalloc failed while sizing the pending-leaf slots");
        E.leaf_slots = K;
        HIPCHK(hipMemsetAsync(E.leaf_node, 0xff, S * 2, e->stream));        // LEAF_NONE
    }
    E.vloss = enable ? 1 : 0;
    return grow_nodes(e);
}

extern "C" int xq_engine_leaf_slots(xq_engine *e) { return e ? e->E.leaf_slots : 0; }

// per game: nodes in the arena and the sum of pending (virtual-loss) visits
__global__ void k_tree_stats(Eng E, int32_t *n_nodes, int32_t *vl_sum)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.G) return;
    const int n = (int)E.n_nodes[g];
    int sum = 0;
    for (int i = 0; i < n; i++) sum += E.nVl[(size_t)g * E.ncap + i];
    n_nodes[g] = n;
    vl_sum[g] = sum;
}
extern "C" int xq_engine_tree_stats(xq_engine *e, int32_t *n_nodes_host, int32_t *vl_sum_host)
{
    if (!e || !n_nodes_host || !vl_sum_host) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    hipLaunchKernelGGL(k_tree_stats, dim3((e->E.G + 63) / 64), dim3(64), 0, e->stream, e->E, e->stage_rvisits,
                       e->stage_rvisits + G);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(n_nodes_host, e->stage_rvisits, G * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(vl_sum_host, e->stage_rvisits + G, G * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int xq_engine_set_logit_columns(xq_engine *e, const int16_t *map, int n_columns)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (!map || n_columns <= 0) { e->E.col_map = nullptr; e->E.n_cols = 0; return 0; }
    if (n_columns > 16384) return fail(XQ_E_INVALID, "n_columns > 16384");     // (rows may be padded past 8,100 columns)
    for (int i = 0; i < XQ_POLICY_SIZE; i++)
        if (map[i] < -1 || map[i] >= n_columns) return fail(XQ_E_INVALID, "column map entry out of range");
    HIPCHK(hipSetDevice(e->cfg.device));
    int16_t *p = nullptr;
    if (dalloc(e, p, (size_t)XQ_POLICY_SIZE)) return fail(XQ_E_HIP, "hipMalloc failed");
    HIPCHK(hipMemcpyAsync(p, map, XQ_POLICY_SIZE * 2, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->E.col_map = p; e->E.n_cols = n_columns;
    return 0;
}

extern "C" int xq_engine_set_uniforms(xq_engine *e, const double *u)
{
    if (!e || !u) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemcpyAsync(e->E.uniforms, u, (size_t)e->E.G * XQ_MAX_PLIES * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int xq_engine_set_roots(xq_engine *e, const int8_t *boards, const int32_t *state)
{
    if (!e || !boards || !state) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    HIPCHK(hipMemcpyAsync(e->stage_boards, boards, G * 90, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->stage_state, state, G * XQ_STATE_WORDS * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->E.eval_carry) HIPCHK(hipMemsetAsync(e->E.roots_not_ready, 0, 4, e->stream));     // k_set_roots counts them
    e->ec_epoch += 2;
    e->active_posted = 0;
    hipLaunchKernelGGL(k_set_roots, dim3(e->E.G), dim3(64), 0, e->stream, e->E, e->stage_boards, e->stage_state);
    HIPCHK(hipGetLastError());
    return 0;
}

static int planes_ok(int fmt) { return fmt >= XQ_PLANES_NONE && fmt <= XQ_PLANES_NHWC16_BF16; }

static std::pair<hipEvent_t, hipEvent_t> *next_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used)
{
    if (used == pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        pool.emplace_back(a, b);
    }
    return &pool[used++];
}

static int drain_events(xq_engine *e)
{
    HIPCHK(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < e->ev_search_used; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e->ev_search[i].first, e->ev_search[i].second));
        e->search_ms += ms; e->search_n++;
    }
    for (size_t i = 0; i < e->ev_play_used; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e->ev_play[i].first, e->ev_play[i].second));
        e->play_ms += ms; e->play_n++;
    }
    e->ev_search_used = e->ev_play_used = 0;
    return 0;
}

static int g_search_occ = 4;      // 128 VGPRs, 4 waves/SIMD: fastest of {3, 4, 5, 6, 8}: 0.097 ms (3: 0.118; 5: 0.101 with
                                  // 56 B/lane of scratch that also adds 46 MB of HBM writes per launch)
// diagnostic (include/xq_debug.h): register budget variant of k_search_round
extern "C" void xq_engine_set_search_occupancy(int waves_per_simd) { g_search_occ = waves_per_simd; }
// diagnostic (include/xq_debug.h): device buffer of 16 u64 per game for k_search_round's phase stamps (probes library
// only; NULL = off).  Every launch overwrites it: read it after the round of interest.
#if XQ_TOWER_PROBES
static void *g_search_stamps = nullptr;
#endif
extern "C" int xq_engine_set_search_stamps(void *dev_u64x16_per_game)
{
#if XQ_TOWER_PROBES
    g_search_stamps = dev_u64x16_per_game;
    return 0;
#else
    (void)dev_u64x16_per_game;
    return XQ_E_INVALID;
#endif
}

extern "C" int xq_engine_search_round(xq_engine *e, int round, int eval_kind, const void *ev_a, const void *ev_v,
                                      void *planes)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (round < 0 || round >= e->E.nrounds) return fail(XQ_E_INVALID, "round out of range");
    if (eval_kind < XQ_EVAL_PRIORS || eval_kind > XQ_EVAL_LOGITS_BF16) return fail(XQ_E_INVALID, "bad eval_kind");
    if (round > 0 && (!ev_a || !ev_v)) return fail(XQ_E_INVALID, "evaluator output missing");
    if (!planes_ok(e->cfg.planes_format)) return fail(XQ_E_INVALID, "bad planes_format");
    if (eval_kind != XQ_EVAL_PRIORS && ev_a && e->E.compact && !e->row_map_fetched)
        return fail(XQ_E_INVALID, "logits handed in by slot while row compaction is on: fetch xq_engine_row_map (rows are compacted) "
                                  "or switch xq_engine_set_row_compaction off");
    HIPCHK(hipSetDevice(e->cfg.device));
    const int start = round * e->E.leaf_batch;
    const int batch = (start + e->E.leaf_batch <= e->E.sims) ? e->E.leaf_batch : e->E.sims - start;
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (e->prof && (e->prof_seen++ % (unsigned)e->prof) == 0) {
        if (e->ev_search_used >= 4096) { if (int rc = drain_events(e)) return rc; }
        ev = next_events(e->ev_search, e->ev_search_used);
        if (ev) HIPCHK(hipEventRecord(ev->first, e->stream));
    }
    const int fmt = planes ? e->cfg.planes_format : XQ_PLANES_NONE;
    e->E.ec_epoch = e->ec_epoch; e->E.ec_seq = ++e->ec_seq;
    if (e->E.dedupe) {                                                    // a fresh tag for this round's table entries
        if (++e->dd_tag == 0) {                                           // tag wrap: start from a cleared table
            HIPCHK(hipMemsetAsync(e->E.dd_tab, 0, ((size_t)e->E.dd_mask + 1) * 8, e->stream));
            e->dd_tag = 1;
        }
        e->E.dd_tag = e->dd_tag;
    }
#define XQ_LAUNCH_SEARCH(OCC, VLB) hipLaunchKernelGGL((k_search_round<OCC, VLB>), dim3(e->E.G), dim3(64), 0, e->stream, e->E, \
                                                    round, batch, eval_kind, ev_a, ev_v, planes, fmt, (unsigned long long *)nullptr)
    if (e->E.vloss) XQ_LAUNCH_SEARCH(4, true);
#if XQ_TOWER_PROBES
    else if (g_search_stamps)
        hipLaunchKernelGGL((k_search_round<4, false, true>), dim3(e->E.G), dim3(64), 0, e->stream, e->E, round, batch, eval_kind,
                           ev_a, ev_v, planes, fmt, (unsigned long long *)g_search_stamps);
#endif
    else switch (g_search_occ) {
#if XQ_TOWER_PROBES                 // the other register budgets spill 27-58 VGPRs: comparison builds, probes library only
    case 3: XQ_LAUNCH_SEARCH(3, false); break;
    case 5: XQ_LAUNCH_SEARCH(5, false); break;
    case 6: XQ_LAUNCH_SEARCH(6, false); break;
    case 8: XQ_LAUNCH_SEARCH(8, false); break;
#endif
    default: XQ_LAUNCH_SEARCH(4, false); break;
    }
#undef XQ_LAUNCH_SEARCH
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->second, e->stream));
    if (e->E.compact) {
        // number the rows the evaluator has to run for this round (outside the k_search_round events)
        hipLaunchKernelGGL(k_assign_rows, dim3(1), dim3(1024), 0, e->stream, e->E, e->E.G * e->E.leaf_slots, e->row_seq);
        HIPCHK(hipGetLastError());
        e->row_seq++;
    }
    return 0;
}

// Evaluator row compaction (default off: row = slot, every slot is evaluated).  When on, every search round is
// followed by k_assign_rows, the logits / values of a pending leaf are read from row leaf_row[slot], and the
// evaluator is expected to (1) read the planes of row r at slot row_src[r], (2) write its outputs to row r,
// (3) stop at row_count[0] rows (xq_engine_row_map hands out the two device pointers; xq_tower_nhwc_bf16,
// xq_policy_fc_bf16 and xq_value_head_bf16 take them).  Rows are numbered in slot order, so a run is reproducible;
// results do not depend on the numbering (every kernel of the evaluator is row-independent).  What it buys: games
// that are over, rounds that found only terminal leaves, roots carried over by xq_engine_set_root_eval_carry and
// empty virtual-loss slots cost no network time - without a host round trip.  Evaluators that fill
// xq_engine_priors_ptr (XQ_EVAL_PRIORS) are indexed by slot either way.
extern "C" int xq_engine_set_row_compaction(xq_engine *e, int enable)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if ((enable ? 1 : 0) != e->E.compact) e->row_map_fetched = false;
    e->E.compact = enable ? 1 : 0;
    if (!enable) e->E.dedupe = 0;
    return 0;
}

// Leaf dedupe (default off; needs row compaction): pending leaves that are the same position - same board, same side
// to move, which is all the network input encodes (neural_network.py:128-146) - share one evaluator row
// (k_dedupe_leaves + k_assign_rows: leaf_row of every slot of the group = the row of the group's lowest slot).
// Result-identical for an evaluator whose output for a position depends on nothing else (not on the row, the batch
// size or the launch: true of xq_tower_nhwc_bf16 / xq_policy_fc_bf16 / xq_value_head_bf16, each output element is one
// fixed fp32 chain); the caller vouches for that by switching it on.  The reference has no counterpart: its workers
// evaluate every leaf of every game (self_play.py:137-139).
extern "C" int xq_engine_set_leaf_dedupe(xq_engine *e, int enable)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (enable && !e->E.compact) return fail(XQ_E_INVALID, "leaf dedupe needs row compaction (xq_engine_set_row_compaction)");
    e->E.dedupe = enable ? 1 : 0;
    return 0;
}

// Evaluation cache (default off): see Eng::ec_state.  log2_entries = 0 switches it off; otherwise the table has
// 2^log2_entries entries of 576 B (16 <= log2_entries <= 24; two plies of evaluations should fill it to a half at most).
// Result-identical for a deterministic evaluator whose answer depends on the position only (the caller vouches for that,
// as for the leaf dedupe); entries live for two plies and never survive xq_engine_new_games / set_roots / refill_begin.
extern "C" int xq_engine_set_eval_cache(xq_engine *e, int log2_entries)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    const bool verify = log2_entries < 0;          // (diagnostic: -log2_entries = the size; hits keep their rows and are compared)
    if (verify) log2_entries = -log2_entries;
    if (log2_entries == 0) { e->E.ec_on = 0; return 0; }
    if (log2_entries < 10 || log2_entries > 24) return fail(XQ_E_INVALID, "eval cache: 10 <= log2_entries <= 24");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t n = (size_t)1 << log2_entries;
    if (!e->E.ec_state || (size_t)e->E.ec_mask + 1 != n) {
        HIPCHK(hipStreamSynchronize(e->stream));
        int bad = dalloc(e, e->E.ec_state, n);
        bad |= dalloc(e, e->E.ec_key, n * 16); bad |= dalloc(e, e->E.ec_prior, n * MAXM);
        if (!e->E.leaf_ec) bad |= dalloc(e, e->E.leaf_ec, (size_t)e->E.G * 64);
        if (!e->E.ec_stats) bad |= dalloc(e, e->E.ec_stats, (size_t)3 * e->E.G * 64);
        if (bad) return fail(XQ_E_HIP, "hipMalloc failed for the evaluation cache");
        HIPCHK(hipMemsetAsync(e->E.leaf_ec, 0xff, (size_t)e->E.G * 64 * 4, e->stream));
        e->E.ec_mask = (int)(n - 1);
    }
    e->ec_epoch += 2;
    e->E.ec_on = verify ? 2 : 1;
    e->E.ec_stat_stride = e->E.G * 64;
    return 0;
}

// hits / fills / verify-mode mismatches since the last reset (diagnostic of the evaluation cache; counted per slot on the
// device - a shared counter would be 16,384 atomics on one address per launch - and summed here)
extern "C" int xq_engine_eval_cache_stats(xq_engine *e, uint64_t *hits_fills_host /*[3]*/, int reset)
{
    if (!e || !hits_fills_host) return fail(XQ_E_INVALID, "null argument");
    hits_fills_host[0] = hits_fills_host[1] = hits_fills_host[2] = 0;
    if (!e->E.ec_stats) return 0;
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t stride = (size_t)e->E.G * 64, slots = (size_t)e->E.G * (size_t)e->E.leaf_slots;
    std::vector<uint32_t> h(slots);
    for (int k = 0; k < 3; k++) {
        HIPCHK(hipMemcpyAsync(h.data(), e->E.ec_stats + k * stride, slots * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        for (size_t i = 0; i < slots; i++) hits_fills_host[k] += h[i];
        if (reset) HIPCHK(hipMemsetAsync(e->E.ec_stats + k * stride, 0, slots * 4, e->stream));
    }
    return 0;
}

extern "C" int xq_engine_row_map(xq_engine *e, const int32_t **row_src_dev, const int32_t **row_count_dev)
{
    if (!e || !row_src_dev || !row_count_dev) return fail(XQ_E_INVALID, "null argument");
    e->row_map_fetched = true;
    *row_src_dev = e->E.compact ? e->E.row_src : nullptr;
    *row_count_dev = e->E.compact ? e->E.row_count : nullptr;
    return 0;
}

// rows of the last `cap` search rounds (oldest first) and the number of rounds launched since the engine was
// created or the history was last reset (reset != 0 restarts the count after reading)
extern "C" int xq_engine_read_row_history(xq_engine *e, int32_t *rows, int cap, int64_t *n_rounds, int reset)
{
    if (!e || !n_rounds || cap < 0 || (cap > 0 && !rows)) return fail(XQ_E_INVALID, "bad argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const unsigned n = e->row_seq;
    *n_rounds = n;
    int take = (int)(n < (unsigned)cap ? n : (unsigned)cap);
    if (take > ROW_HIST) take = ROW_HIST;
    if (take > 0) {
        std::vector<int32_t> ring(ROW_HIST);
        HIPCHK(hipMemcpy(ring.data(), e->E.row_hist, (size_t)ROW_HIST * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < take; i++) rows[i] = ring[(n - (unsigned)take + (unsigned)i) & (ROW_HIST - 1)];
    }
    if (reset) e->row_seq = 0;
    return 0;
}

extern "C" int xq_engine_read_leaf_rows(xq_engine *e, int32_t *rows /*[G * leaf_slots]*/)
{
    if (!e || !rows) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t S = (size_t)e->E.G * e->E.leaf_slots;
    HIPCHK(hipMemcpyAsync(rows, e->E.leaf_row, S * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!e->E.compact) for (size_t i = 0; i < S; i++) rows[i] = (int32_t)i;
    return 0;
}

extern "C" int xq_engine_eval_hashnet(xq_engine *e, int salt)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(k_hashnet, dim3(e->E.G * e->E.leaf_slots), dim3(64), 0, e->stream, e->E, salt);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xq_engine_end_search(xq_engine *e, int eval_kind, const void *ev_a, const void *ev_v)
{
    if (!e || !ev_a || !ev_v) return fail(XQ_E_INVALID, "null argument");
    if (eval_kind < XQ_EVAL_PRIORS || eval_kind > XQ_EVAL_LOGITS_BF16) return fail(XQ_E_INVALID, "bad eval_kind");
    if (eval_kind != XQ_EVAL_PRIORS && e->E.compact && !e->row_map_fetched)
        return fail(XQ_E_INVALID, "logits handed in by slot while row compaction is on: fetch xq_engine_row_map or switch it off");
    HIPCHK(hipSetDevice(e->cfg.device));
    e->E.ec_epoch = e->ec_epoch; e->E.ec_seq = ++e->ec_seq;
    hipLaunchKernelGGL(k_end_search, dim3(e->E.G), dim3(64), 0, e->stream, e->E, eval_kind, ev_a, ev_v);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xq_engine_play_move(xq_engine *e)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (e->prof) {
        ev = next_events(e->ev_play, e->ev_play_used);
        if (ev) HIPCHK(hipEventRecord(ev->first, e->stream));
    }
    if (e->E.eval_carry) HIPCHK(hipMemsetAsync(e->E.roots_not_ready, 0, 4, e->stream));
    e->ec_epoch += 1;                                                    // the evaluation cache keeps two plies
    hipLaunchKernelGGL(k_play_move, dim3(e->E.G), dim3(64), 0, e->stream, e->E);
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->second, e->stream));
    return 0;
}

// Opt-in, result-identical work elimination (default off; no counterpart in the reference, which rebuilds its
// tree every ply, self_play.py:98): the network evaluation of the position a game moves into is carried over
// from this ply's search to the next ply's root instead of being computed a second time.  The played child has
// visits, so it was expanded from a network row for exactly that (board, player); the evaluator is
// deterministic and row-independent, so the reference's fresh evaluation of the new root would return the same
// priors bit for bit.  xq_engine_play_move then leaves every continuing game with the tree round 0 would have
// produced (root.visit_count = first batch, children unvisited); the caller skips round 0 — tree kernel and
// network forward — whenever xq_engine_roots_not_ready reports 0 (it reports the games that still need it:
// fresh games, and any game whose played child was not expanded).  Not combinable with tree reuse, virtual loss
// or root noise.  Call before xq_engine_new_games.
extern "C" int xq_engine_set_root_eval_carry(xq_engine *e, int enable)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    Eng &E = e->E;
    if (enable && (E.tree_reuse || E.vloss || E.noise_eps > 0.0 || E.opponent_mode))
        return fail(XQ_E_INVALID, "root evaluation carry-over cannot be combined with tree reuse, virtual loss, root noise or "
                                  "opponent mode (the carried priors are the mover's network's)");
    HIPCHK(hipSetDevice(e->cfg.device));
    if (enable && !E.root_ready) {
        if (dalloc(e, E.root_ready, (size_t)E.G) || dalloc(e, E.roots_not_ready, (size_t)1)) return fail(XQ_E_HIP, "hipMalloc failed");
    }
    E.eval_carry = enable ? 1 : 0;
    return 0;
}

extern "C" int xq_engine_roots_not_ready(xq_engine *e, int32_t *n)
{
    if (!e || !n) return fail(XQ_E_INVALID, "null argument");
    if (!e->E.eval_carry) { *n = e->E.G; return 0; }
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemcpyAsync(n, e->E.roots_not_ready, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int xq_engine_finalize(xq_engine *e)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(k_finalize, dim3(e->E.G), dim3(64), 0, e->stream, e->E);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xq_engine_refill_begin(xq_engine *e, const uint32_t *seeds, int total)
{
    if (!e || !seeds) return fail(XQ_E_INVALID, "null argument");
    if (total < e->E.G) return fail(XQ_E_INVALID, "refill needs at least as many games as slots");
    if (e->E.opponent_mode) return fail(XQ_E_INVALID, "refill is not available in opponent (arena) mode: slots are at different plies");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G, T = (size_t)total;
    if (total > e->refill_cap) {             // (buffers of an earlier, smaller session stay in e->allocs until destroy)
        if (dalloc(e, e->uni_all, T * XQ_MAX_PLIES) || dalloc(e, e->out_gs, T)) return fail(XQ_E_HIP, "hipMalloc failed");
        e->refill_cap = total;
    }
    e->refill_total = total;                 // the session's size, every session: k_refill deals game ids below it only
    if (!e->slot_game && (dalloc(e, e->slot_game, G) || dalloc(e, e->next_game, (size_t)1)))
        return fail(XQ_E_HIP, "hipMalloc failed");
    std::vector<double> u(T * XQ_MAX_PLIES);
    mt_uniforms_many(seeds, T, u.data());
    std::vector<int32_t> sg(G);
    for (size_t g = 0; g < G; g++) sg[g] = (int32_t)g;
    const int32_t next = (int32_t)G;
    HIPCHK(hipMemcpyAsync(e->uni_all, u.data(), u.size() * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->E.uniforms, e->uni_all, G * XQ_MAX_PLIES * 8, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->slot_game, sg.data(), G * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->next_game, &next, 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->out_gs, 0, T * sizeof(GameS), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));          // locals: the copies must finish
    if (e->E.eval_carry) HIPCHK(hipMemsetAsync(e->E.roots_not_ready, 0, 4, e->stream));
    e->ec_epoch += 2;                                                    // new games, possibly new weights: nothing cached survives
    e->active_posted = 0;                                                // (a count posted for the last games says nothing about these)
    hipLaunchKernelGGL(k_new_games, dim3(e->E.G), dim3(64), 0, e->stream, e->E);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xq_engine_refill_step(xq_engine *e, void *records, int32_t *active)
{
    if (!e || !records) return fail(XQ_E_INVALID, "null argument");
    if (!e->slot_game || e->refill_total <= 0) return fail(XQ_E_INVALID, "xq_engine_refill_begin first");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemsetAsync(e->active_dev, 0, 4, e->stream));
    Refill R{ e->uni_all, e->slot_game, e->next_game, e->active_dev, e->out_gs, (xq_sample_record *)records, e->refill_total };
    hipLaunchKernelGGL(k_refill, dim3(e->E.G), dim3(64), 0, e->stream, e->E, R);
    HIPCHK(hipGetLastError());
    if (active) {
        HIPCHK(hipMemcpyAsync(active, e->active_dev, 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return 0;
}

extern "C" int xq_engine_refill_read_slots(xq_engine *e, int32_t *slot_game /*[G]*/)
{
    if (!e || !slot_game) return fail(XQ_E_INVALID, "null argument");
    if (!e->slot_game || e->refill_total <= 0) return fail(XQ_E_INVALID, "xq_engine_refill_begin first");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemcpyAsync(slot_game, e->slot_game, (size_t)e->E.G * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int xq_engine_refill_read_games(xq_engine *e, int32_t *winner, int32_t *reason, int32_t *reason_side,
                                           int32_t *reason_count, int32_t *n_plies, int32_t *n_samples, int32_t *error)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (!e->out_gs || e->refill_total <= 0) return fail(XQ_E_INVALID, "xq_engine_refill_begin first");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t T = (size_t)e->refill_total;
    std::vector<GameS> gs(T);
    HIPCHK(hipMemcpyAsync(gs.data(), e->out_gs, T * sizeof(GameS), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (size_t g = 0; g < T; g++) {
        if (winner) winner[g] = gs[g].winner == WINNER_NONE ? 0 : gs[g].winner;       // self_play.py:259
        if (reason) reason[g] = gs[g].reason;
        if (reason_side) reason_side[g] = gs[g].reason_side;
        if (reason_count) reason_count[g] = gs[g].reason_count;
        if (n_plies) n_plies[g] = gs[g].n_plies;
        if (n_samples) n_samples[g] = gs[g].n_samples;
        if (error) error[g] = gs[g].error;
    }
    return 0;
}

extern "C" int xq_engine_active_games(xq_engine *e, int32_t *n)
{
    if (!e || !n) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemsetAsync(e->active_dev, 0, 4, e->stream));
    hipLaunchKernelGGL(k_count_active, dim3((e->E.G + 255) / 256), dim3(256), 0, e->stream, e->E.gs, e->E.G, e->active_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(n, e->active_dev, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// The same count without stopping the host: _post enqueues the count and its copy into pinned host memory behind
// everything enqueued so far; _poll says whether the newest posted count has arrived (1) or not yet (0) and hands it out.
// A driver loop that posts every few plies and polls before it posts learns "all games over" a few plies late - plies of
// finished games cost next to nothing - and never drains the stream (the blocking form leaves the GPU idle until the
// host has enqueued the next ply again: ~0.3 % of a lock-step epoch).
extern "C" int xq_engine_active_games_post(xq_engine *e)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    if (!e->active_host) {
        HIPCHK(hipHostMalloc((void **)&e->active_host, 2 * sizeof(int32_t)));
        HIPCHK(hipEventCreateWithFlags(&e->active_ev, hipEventDisableTiming));
    }
    // two dwords in turn, so that a count being written is never the one being read
    const int slot = e->active_posted & 1;
    HIPCHK(hipMemsetAsync(e->active_dev, 0, 4, e->stream));
    hipLaunchKernelGGL(k_count_active, dim3((e->E.G + 255) / 256), dim3(256), 0, e->stream, e->E.gs, e->E.G, e->active_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->active_host + slot, e->active_dev, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipEventRecord(e->active_ev, e->stream));
    e->active_posted++;
    return 0;
}

extern "C" int xq_engine_active_games_poll(xq_engine *e, int32_t *n, int32_t *ready)
{
    if (!e || !n || !ready) return fail(XQ_E_INVALID, "null argument");
    *ready = 0;
    if (!e->active_host || e->active_posted == 0) return 0;
    const hipError_t q = hipEventQuery(e->active_ev);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) return fail(XQ_E_HIP, "hipEventQuery failed");
    *n = e->active_host[(e->active_posted - 1) & 1];
    *ready = 1;
    return 0;
}

extern "C" void *xq_engine_priors_ptr(xq_engine *e) { return e ? e->E.priors : nullptr; }
extern "C" void *xq_engine_values_ptr(xq_engine *e) { return e ? e->E.values : nullptr; }
extern "C" int xq_engine_rounds_per_move(xq_engine *e) { return e ? e->E.nrounds : 0; }

template <class T> static int d2h(xq_engine *e, T *dst, const T *src, size_t count)
{
    if (!dst) return 0;
    HIPCHK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, e->stream));
    return 0;
}

static void unpack_board_host(const uint32_t *w, int8_t *out)
{
    for (int s = 0; s < 90; s++) {
        uint32_t code = (w[s / 8] >> (4 * (s % 8))) & 15u;
        out[s] = (int8_t)(code <= 7 ? (int)code : 7 - (int)code);
    }
}

extern "C" int xq_engine_read_leaves(xq_engine *e, int8_t *boards, int32_t *player, uint16_t *moves, int32_t *n_moves,
                                     int32_t *mult)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (e->E.leaf_slots != 1) return fail(XQ_E_INVALID, "host-side evaluators see one leaf per game: not available with virtual loss");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    std::vector<uint32_t> pb(G * 12);
    std::vector<int8_t> side(G);
    std::vector<uint8_t> n(G), m(G);
    std::vector<uint16_t> node(G);
    if (int rc = d2h(e, pb.data(), e->E.leaf_board, G * 12)) return rc;
    if (int rc = d2h(e, side.data(), e->E.leaf_side, G)) return rc;
    if (int rc = d2h(e, n.data(), e->E.leaf_n, G)) return rc;
    if (int rc = d2h(e, m.data(), e->E.leaf_mult, G)) return rc;
    if (int rc = d2h(e, node.data(), e->E.leaf_node, G)) return rc;
    if (int rc = d2h(e, moves, e->E.leaf_moves, G * MAXM)) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    for (size_t g = 0; g < G; g++) {
        const bool pending = node[g] != LEAF_NONE;
        if (boards) unpack_board_host(pb.data() + g * 12, boards + g * 90);
        if (player) player[g] = side[g];
        if (n_moves) n_moves[g] = pending ? n[g] : 0;
        if (mult) mult[g] = pending ? m[g] : 0;
    }
    return 0;
}

extern "C" int xq_engine_write_priors(xq_engine *e, const float *priors, const double *values)
{
    if (!e || !priors || !values) return fail(XQ_E_INVALID, "null argument");
    if (e->E.leaf_slots != 1) return fail(XQ_E_INVALID, "host-side evaluators see one leaf per game: not available with virtual loss");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    HIPCHK(hipMemcpyAsync(e->E.priors, priors, G * MAXM * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->E.values, values, G * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int xq_engine_read_root_visits(xq_engine *e, uint16_t *moves, int32_t *visits, int32_t *n_child)
{
    if (!e || !moves || !visits || !n_child) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    hipLaunchKernelGGL(k_root_children, dim3(e->E.G), dim3(64), 0, e->stream, e->E, e->stage_rmoves, e->stage_rvisits,
                       (float *)nullptr, e->stage_rn);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(moves, e->stage_rmoves, G * MAXM * 2, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(visits, e->stage_rvisits, G * MAXM * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(n_child, e->stage_rn, G * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

// the search tree of one game as it stands (after xq_engine_end_search: the finished tree of the ply): the node arena's
// slices, node by node in creation order - MCTSNode's fields (self_play.py:19-28), children = [first, first + n_child)
extern "C" int xq_engine_read_tree(xq_engine *e, int game, int cap, int32_t *n_nodes, int32_t *root, uint32_t *visit_count,
                                   double *value_sum, float *prior, uint16_t *move, uint16_t *first_child, uint8_t *n_child)
{
    if (!e || !n_nodes || !root) return fail(XQ_E_INVALID, "null argument");
    if (game < 0 || game >= e->E.G || cap < 0) return fail(XQ_E_INVALID, "xq_engine_read_tree: game out of range");
    HIPCHK(hipSetDevice(e->cfg.device));
    uint32_t nn = 0; uint16_t rt = 0;
    HIPCHK(hipMemcpyAsync(&nn, e->E.n_nodes + game, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(&rt, e->E.root_node + game, 2, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *n_nodes = (int32_t)nn; *root = (int32_t)rt;
    const size_t n = nn < (uint32_t)cap ? nn : (size_t)cap, off = (size_t)game * e->E.ncap;
    if (n && visit_count) HIPCHK(hipMemcpyAsync(visit_count, e->E.nN + off, n * 4, hipMemcpyDeviceToHost, e->stream));
    if (n && value_sum) HIPCHK(hipMemcpyAsync(value_sum, e->E.nW + off, n * 8, hipMemcpyDeviceToHost, e->stream));
    if (n && prior) HIPCHK(hipMemcpyAsync(prior, e->E.nP + off, n * 4, hipMemcpyDeviceToHost, e->stream));
    if (n && move) HIPCHK(hipMemcpyAsync(move, e->E.nMove + off, n * 2, hipMemcpyDeviceToHost, e->stream));
    if (n && first_child) HIPCHK(hipMemcpyAsync(first_child, e->E.nFirst + off, n * 2, hipMemcpyDeviceToHost, e->stream));
    if (n && n_child) HIPCHK(hipMemcpyAsync(n_child, e->E.nNc + off, n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int xq_engine_read_games(xq_engine *e, int32_t *winner, int32_t *reason, int32_t *reason_side,
                                    int32_t *reason_count, int32_t *n_plies, int32_t *n_samples, int32_t *error)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t G = (size_t)e->E.G;
    std::vector<GameS> gs(G);
    HIPCHK(hipMemcpyAsync(gs.data(), e->E.gs, G * sizeof(GameS), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (size_t g = 0; g < G; g++) {
        if (winner) winner[g] = gs[g].winner == WINNER_NONE ? 0 : gs[g].winner;       // self_play.py:259
        if (reason) reason[g] = gs[g].reason;
        if (reason_side) reason_side[g] = gs[g].reason_side;
        if (reason_count) reason_count[g] = gs[g].reason_count;
        if (n_plies) n_plies[g] = gs[g].n_plies;
        if (n_samples) n_samples[g] = gs[g].n_samples;
        if (error) error[g] = gs[g].error;
    }
    return 0;
}

extern "C" int xq_engine_read_samples(xq_engine *e, int8_t *boards, int8_t *player, uint8_t *n_moves, uint16_t *moves,
                                      uint16_t *counts, double *z, uint16_t *chosen, double *step_reward)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t GP = (size_t)e->E.G * XQ_MAX_PLIES;
    std::vector<uint32_t> pb;
    if (boards) {
        pb.resize(GP * 12);
        if (int rc = d2h(e, pb.data(), e->E.s_board, GP * 12)) return rc;
    }
    if (int rc = d2h(e, player, e->E.s_player, GP)) return rc;
    if (int rc = d2h(e, n_moves, e->E.s_n, GP)) return rc;
    if (int rc = d2h(e, moves, e->E.s_moves, GP * MAXM)) return rc;
    if (int rc = d2h(e, counts, e->E.s_counts, GP * MAXM)) return rc;
    if (int rc = d2h(e, z, e->E.s_z, GP)) return rc;
    if (int rc = d2h(e, chosen, e->E.t_move, GP)) return rc;
    if (int rc = d2h(e, step_reward, e->E.step_reward, GP)) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (boards)
        for (size_t i = 0; i < GP; i++) unpack_board_host(pb.data() + i * 12, boards + i * 90);
    return 0;
}

extern "C" int xq_engine_pack_samples(xq_engine *e, void *records)
{
    if (!e || !records) return fail(XQ_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(k_pack_samples, dim3(e->E.G), dim3(64), 0, e->stream, e->E,
                       reinterpret_cast<xq_sample_record *>(records));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xq_engine_profile(xq_engine *e, int enable)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (int rc = drain_events(e)) return rc;
    e->prof = enable > 0 ? enable : 0;        // (enable = N > 1: every N-th search launch is timed - an event pair is not free)
    e->prof_seen = 0;
    if (enable) { e->search_ms = e->play_ms = 0; e->search_n = e->play_n = 0; }
    return 0;
}

extern "C" int xq_engine_profile_read(xq_engine *e, double *search_ms, int64_t *search_n, double *play_ms, int64_t *play_n)
{
    if (!e) return fail(XQ_E_INVALID, "null engine");
    if (int rc = drain_events(e)) return rc;
    if (search_ms) *search_ms = e->search_ms;
    if (search_n) *search_n = e->search_n;
    if (play_ms) *play_ms = e->play_ms;
    if (play_n) *play_n = e->play_n;
    return 0;
}

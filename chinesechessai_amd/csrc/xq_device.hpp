// xq_device.hpp — wave-cooperative Xiangqi rules for gfx950 (CDNA4, wave64).
//
// One 64-lane wavefront owns one game.  The 10x9 board lives as 96 signed bytes in LDS for the
// duration of a kernel (nibble-packed, 48 B, in HBM); piece sets are held as wave-uniform 90-bit
// bitboards produced by __ballot, and per position the wave builds per-row / per-column bit masks
// of the occupancy and of the attackers by type (xq_attack.hpp: AttackMaps), so that ray scans and
// the king-centric legality filter are a handful of bit operations per lane instead of board walks.  Nothing here is a translation of the reference's Python: the behaviour
// (SURVEY.md Appendix A quirks included) is the contract, checked bit-for-bit against the oracle.
//
// Reference behaviour restated: chess_env.py:76-121 (legal moves, order), :123-251 (generators),
// :431-548 (suicide filter / kings facing / in-check), :253-406 (make_move), :683-737 (shaping).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "xq_attack.hpp"

namespace xq {

enum : int { KING = 1, ADVISOR = 2, BISHOP = 3, KNIGHT = 4, ROOK = 5, CANNON = 6, PAWN = 7 };
enum : int { WINNER_NONE = 2, NO_KING = -1, MAXM = 128 };
enum : int { R_NONE = 0, R_KING_CAPTURED = 1, R_CHECKMATE = 2, R_REPETITION = 3, R_FIFTY = 4,
             R_STALEMATE = 5, R_PERP_CHECK = 6, R_PERP_CHASE = 7, R_MOVE_CAP = 8 };

#define XQ_LANE ((int)(threadIdx.x & 63))

// Intra-wave LDS hand-off between lanes.  A wave's LDS instructions execute in issue order, so
// no s_barrier is needed: the fence only pins the compiler's ordering.  (Workgroups may hold
// several independent waves = games; they never exchange data.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------- 90-bit bitboards
struct BB {            // bit s of (lo,hi): lo = squares 0..63, hi = squares 64..89
    uint64_t lo, hi;
};

__device__ __forceinline__ int bb_test(const BB &b, int s)
{
    return (int)(((s < 64) ? (b.lo >> s) : (b.hi >> (s - 64))) & 1ull);
}
// Wave-uniform view of a board held in LDS (all fields identical in every lane).
struct BoardView {
    BB occ;              // occupancy, row-major
    BB red, blk;         // piece sets by colour (row-major)
    int pA, pB;          // per lane: piece on square lane, and on square 64+lane (lane < 26)
};

__device__ __forceinline__ BoardView load_view(const int8_t *bd)
{
    BoardView v;
    const int lane = XQ_LANE;
    v.pA = bd[lane];
    v.pB = (lane < 26) ? bd[64 + lane] : 0;
    v.occ.lo = __ballot(v.pA != 0);  v.occ.hi = __ballot(v.pB != 0);
    v.red.lo = __ballot(v.pA > 0);   v.red.hi = __ballot(v.pB > 0);
    v.blk.lo = __ballot(v.pA < 0);   v.blk.hi = __ballot(v.pB < 0);
    return v;
}

__device__ __forceinline__ bool in_palace(int X, int r, int c)
{
    return (X == 1 ? (r >= 7 && r <= 9) : (r >= 0 && r <= 2)) && c >= 3 && c <= 5;
}

// The per-row / per-column bit masks of xq_attack.hpp for the board of view `v`, attackers = the pieces of colour `att`:
// lane = square, one ds_or_b32 per table a piece belongs to (LDS atomics of one wave execute in issue order; the
// tables are wave-private).  Returns the rows that hold attacker kings / advisors / bishops (bit r + 2, wave-uniform):
// xq_attack.hpp's `kab` is a range test on it.
__device__ __forceinline__ uint32_t build_attack_maps(AttackMaps &M, const BoardView &v, int att)
{
    const int lane = XQ_LANE;
    uint32_t *w = reinterpret_cast<uint32_t *>(&M);
    w[lane] = 0u;
    if (lane < ATTACK_MAP_DWORDS - 64) w[64 + lane] = 0u;
    wave_sync();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = h ? v.pB : v.pA;
        if (p != 0) {
            const int s = lane + 64 * h, r = (s * 57) >> 9, c = s - 9 * r;
            atomicOr(&M.occrow[r + 2], 4u << c);
            atomicOr(&M.occcol[c], 1u << r);
            if ((p ^ att) >= 0) {                                    // a piece of the attackers' colour
                const int T = p < 0 ? -p : p, sh = attack_type_shift(T);
                atomicOr(&w[attack_row_table(T) + r + 2], (4u << c) << sh);
                if (T == ROOK || T == CANNON) atomicOr(&M.rc_col[c], (1u << r) << sh);
            }
        }
    }
    wave_sync();
    const uint32_t kab = lane < 14 ? (M.ka_row[lane] | M.b_row[lane]) : 0u;
    return (uint32_t)__ballot(kab != 0u);
}

// attacker kings / advisors / bishops anywhere in rows lo - 2 .. hi + 2 ?
__device__ __forceinline__ bool kab_in_rows(uint32_t kab_rows, int lo, int hi)
{
    return ((kab_rows >> lo) & ((1u << (hi - lo + 5)) - 1u)) != 0u;
}

// _is_in_check (chess_env.py:506-548) of the king cached on square k (wave-uniform) against the maps' attackers,
// evaluated with self.current_player == X
__device__ __forceinline__ bool in_check(const AttackMaps &M, uint32_t kab_rows, int k, int X)
{
    if (k < 0) return false;
    const int kr = (k * 57) >> 9;
    return king_attacked<false>(M, k, 0, 0, X, -1, kab_in_rows(kab_rows, kr, kr));
}

// chess_env.py:466-495 on cached king squares, on the board the maps were built from
__device__ __forceinline__ bool kings_facing(const AttackMaps &M, int rk, int bk)
{
    if (rk < 0 || bk < 0) return false;
    int rc = rk % 9, bc = bk % 9;
    if (rc != bc) return false;
    int rr = rk / 9, br = bk / 9;
    int lo = rr < br ? rr : br, hi = rr < br ? br : rr;
    if (hi - lo < 1) return true;          // same square: empty range -> "facing" (as the reference)
    return (M.occcol[rc] & (((1u << (hi - lo - 1)) - 1u) << (lo + 1))) == 0;
}

// ---------------------------------------------------------------- wave scan helper
__device__ __forceinline__ int wave_incl_scan(int x)
{
    const int lane = XQ_LANE;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// ---------------------------------------------------------------- legal move generation
// chess_env.py:76-121.  Output order is part of the contract: own pieces in row-major square
// order, per piece the generator's emission order, filters keep order.
// Lane mapping for candidates: lane = (own piece index & 15) * 4 + direction slot.
// cand / out: LDS u16[128]; own_sq: LDS u8[32].  M / kab_rows: build_attack_maps(M, v, -side) of this board.
// Returns the (wave-uniform) number of legal moves.
__device__ int wave_movegen(const int8_t *bd, const BoardView &v, const AttackMaps &M, uint32_t kab_rows, int side, int rk, int bk,
                            uint16_t *cand, uint16_t *out, uint8_t *own_sq)
{
    const int lane = XQ_LANE;
    const BB &own = (side == 1) ? v.red : v.blk;
    const int n_lo = __builtin_popcountll(own.lo);
    const int n_own = n_lo + __builtin_popcountll(own.hi);

    // own piece list in row-major order
    if (bb_test(own, lane) && lane < 64) {
        int rank = __builtin_popcountll(own.lo & ((1ull << lane) - 1ull));
        if (rank < 32) own_sq[rank] = (uint8_t)lane;
    }
    if (lane < 26 && ((own.hi >> lane) & 1ull)) {
        int rank = n_lo + __builtin_popcountll(own.hi & ((1ull << lane) - 1ull));
        if (rank < 32) own_sq[rank] = (uint8_t)(64 + lane);
    }
    wave_sync();

    int n_cand = 0;
    const int d = lane & 3;
    for (int chunk = 0; chunk < n_own && chunk < 32; chunk += 16) {
        const int pi = chunk + (lane >> 2);
        int cnt = 0, n1 = 0, x0 = -1, x1 = -1, delta = 0, sq = 0;
        if (pi < n_own && pi < 32) {
            sq = own_sq[pi];
            const int r = sq / 9, c = sq % 9;
            int tp = bd[sq];
            tp = tp < 0 ? -tp : tp;
            if (tp == ROOK || tp == CANNON) {
                // ray order right, left, down, up (chess_env.py:203,218)
                const bool horiz = d < 2, fwd = (d & 1) == 0;
                const uint32_t line = horiz ? M.occrow[r + 2] >> 2 : M.occcol[c];
                const int p = horiz ? c : r, len = horiz ? 9 : 10;
                delta = horiz ? (fwd ? 1 : -1) : (fwd ? 9 : -9);
                int n_empty, blk = -1, cap = -1;
                if (fwd) {
                    uint32_t x = line >> (p + 1);
                    if (x == 0) n_empty = len - 1 - p;
                    else {
                        int t = __builtin_ctz(x);
                        n_empty = t; blk = p + 1 + t;
                        uint32_t y = x >> (t + 1);
                        if (y) cap = blk + 1 + __builtin_ctz(y);
                    }
                } else {
                    uint32_t x = line & ((1u << p) - 1u);
                    if (x == 0) n_empty = p;
                    else {
                        int hb = 31 - __builtin_clz(x);
                        n_empty = p - 1 - hb; blk = hb;
                        uint32_t y = x & ((1u << hb) - 1u);
                        if (y) cap = 31 - __builtin_clz(y);
                    }
                }
                n1 = n_empty;
                if (tp == ROOK) {
                    if (blk >= 0) {
                        int bs = horiz ? r * 9 + blk : blk * 9 + c;
                        if (!bb_test(own, bs)) n1 += 1;      // contiguous with the empties
                    }
                } else if (cap >= 0) {
                    int cs = horiz ? r * 9 + cap : cap * 9 + c;
                    if (!bb_test(own, cs)) x0 = cs;
                }
            } else {
                int ar = -1, ac = -1, br2 = -1, bc2 = -1;       // up to two explicit targets
                bool okA = false, okB = false;
                if (tp == KNIGHT) {
                    // offsets (2,1),(2,-1) | (-2,1),(-2,-1) | (1,2),(-1,2) | (1,-2),(-1,-2); each
                    // pair shares its leg (chess_env.py:182-187)
                    int lr, lc;
                    if (d < 2) { int s = d == 0 ? 1 : -1; lr = r + s; lc = c; ar = r + 2 * s; ac = c + 1; br2 = ar; bc2 = c - 1; }
                    else { int s = d == 2 ? 1 : -1; lr = r; lc = c + s; ac = c + 2 * s; ar = r + 1; bc2 = ac; br2 = r - 1; }
                    bool leg_ok = lr >= 0 && lr < 10 && lc >= 0 && lc < 9 && !bb_test(v.occ, lr * 9 + lc);
                    okA = okB = leg_ok;
                } else if (tp == KING) {
                    ar = r + (d == 2 ? 1 : d == 3 ? -1 : 0);
                    ac = c + (d == 0 ? 1 : d == 1 ? -1 : 0);
                    okA = in_palace(side, ar, ac);
                } else if (tp == ADVISOR) {
                    ar = r + (d < 2 ? 1 : -1);
                    ac = c + ((d & 1) == 0 ? 1 : -1);
                    okA = in_palace(side, ar, ac);
                } else if (tp == BISHOP) {
                    int sr = d < 2 ? 1 : -1, sc = (d & 1) == 0 ? 1 : -1;
                    ar = r + 2 * sr; ac = c + 2 * sc;
                    okA = ar >= 0 && ar < 10 && ac >= 0 && ac < 9 &&
                          !(side == 1 ? (ar < 5) : (ar >= 4)) && !bb_test(v.occ, (r + sr) * 9 + c + sc);
                } else if (tp == PAWN) {
                    int fw = side == 1 ? -1 : 1;
                    bool crossed = side == 1 ? (r < 5) : (r >= 5);
                    if (d == 0) { ar = r + fw; ac = c; okA = true; }
                    else if (d == 1) { ar = r; ac = c - 1; okA = crossed; }
                    else if (d == 2) { ar = r; ac = c + 1; okA = crossed; }
                }
                okA = okA && ar >= 0 && ar < 10 && ac >= 0 && ac < 9 && !bb_test(own, ar * 9 + ac);
                okB = okB && br2 >= 0 && br2 < 10 && bc2 >= 0 && bc2 < 9 && !bb_test(own, br2 * 9 + bc2);
                int ta = okA ? ar * 9 + ac : -1, tb = okB ? br2 * 9 + bc2 : -1;
                x0 = okA ? ta : tb;
                x1 = okA ? tb : -1;
            }
            cnt = n1 + (x0 >= 0) + (x1 >= 0);
        }
        const int incl = wave_incl_scan(cnt);
        const int base = n_cand + incl - cnt;
        for (int k = 0; k < 10; k++) {
            if (k < cnt) {
                int t = (k < n1) ? sq + (k + 1) * delta : (k == n1 ? x0 : x1);
                if (base + k < MAXM) cand[base + k] = (uint16_t)(sq * 90 + t);
            }
        }
        n_cand += __shfl(incl, 63, 64);
    }
    if (n_cand > MAXM) n_cand = MAXM;
    wave_sync();

    // suicide filter (chess_env.py:431-464): the own king's square after the move (only a moving king moves it, A5)
    // must not be attacked on the occupancy the move leaves behind, nor face the other king's cached square
    const int K = (side == 1) ? rk : bk, O = (side == 1) ? bk : rk;
    // rows a tested king square can lie in: the cached one, and the palace rows a king move can reach
    int lo = (side == 1) ? 7 : 0, hi = lo + 2;
    if (K >= 0) { const int kr = (K * 57) >> 9; lo = kr < lo ? kr : lo; hi = kr > hi ? kr : hi; }
    const bool kab = kab_in_rows(kab_rows, lo, hi);
    int n_out = 0;
    for (int base = 0; base < n_cand; base += 64) {
        const int j = base + lane;
        bool legal = false;
        int mv = 0;
        if (j < n_cand) {
            mv = cand[j];
            const int f = mv / 90, t = mv - 90 * f;
            const int k = (bd[f] == side) ? t : K;                   // +-KING == +-1 == side
            legal = !king_attacked<true>(M, k, f, t, side, O, kab);
        }
        const uint64_t mask = __ballot(legal);
        if (legal) out[n_out + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint16_t)mv;
        n_out += __builtin_popcountll(mask);
    }
    wave_sync();
    return n_out;
}

// ---------------------------------------------------------------- board packing (HBM form)
// 4 bits per square: 0 empty, 1..7 red K,A,B,N,R,C,P, 8..14 black; 12 dwords = 48 B per board.
// The LDS board is 96 bytes (squares 90..95 hold 0) and 8-byte aligned: one ds_read_b64 per lane, four squares per SWAR step.
__device__ __forceinline__ uint32_t nibbles4(uint32_t w)         // 4 signed bytes (0, +-1..7) -> 4 nibble codes
{
    const uint32_t n = (w >> 7) & 0x01010101u;                    // 1 where the byte is negative
    uint32_t x = ((w ^ (n * 0xFFu)) & 0x07070707u) | (n << 3);    // p > 0: p;  p < 0: 8 + (~p & 7) = 7 - p
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}
__device__ __forceinline__ uint32_t pack_dword(const int8_t *bd, int i)      // squares 8i..8i+7, i < 12
{
    const uint2 v = *reinterpret_cast<const uint2 *>(bd + 8 * i);
    return nibbles4(v.x) | (nibbles4(v.y) << 16);
}

__device__ __forceinline__ void unpack_to_lds(const uint32_t *packed, int8_t *bd)
{
    const int lane = XQ_LANE;
    if (lane < 12) {
        uint32_t w = packed[lane];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t code = (w >> (4 * j)) & 15u;
            int p = code <= 7 ? (int)code : 7 - (int)code;
            bd[8 * lane + j] = (int8_t)p;
        }
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// Position key for the repetition rule (chess_env.py:497-504): the reference compares 64-bit
// Python hashes of board bytes + mover byte; only equality matters, so any 64-bit key of the
// same information is an equivalent contract.  Zobrist-style XOR over the 12 packed dwords.
__device__ __forceinline__ uint64_t position_key(const int8_t *bd, int player_byte)
{
    const int lane = XQ_LANE;
    uint64_t h = 0;
    if (lane < 12) h = mix64(((uint64_t)(lane + 1) << 32) | pack_dword(bd, lane));
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        uint32_t lo = __shfl_xor((uint32_t)h, d, 64), hi = __shfl_xor((uint32_t)(h >> 32), d, 64);
        h ^= ((uint64_t)hi << 32) | lo;
    }
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)h), hi = __builtin_amdgcn_readfirstlane((uint32_t)(h >> 32));
    h = ((uint64_t)hi << 32) | lo;
    return mix64(h ^ (0x9E3779B97F4A7C15ull * (uint64_t)(player_byte + 1)));
}

// ---------------------------------------------------------------- make_move
struct MState {          // wave-uniform scalar state of one env (chess_env.py:14-67 attributes)
    int side, move_count, winner, reason, reason_side, reason_count;
    int rk, bk, nocap, cchk;
};

// chess_env.py:683-737 (uses the board AFTER the move and the mover as current_player)
__device__ __forceinline__ double position_change(int type, int side, int from, int to, int rk, int bk)
{
    const int fr = from / 9, fc = from % 9, tr = to / 9, tc = to % 9;
    double score = 0;
    const int advance = side == 1 ? fr - tr : tr - fr;
    if (advance > 0) {
        if (type == PAWN) score += advance * 2.0;
        else if (type == ROOK || type == CANNON) score += advance * 1.5;
        else if (type == KNIGHT) score += advance * 1.0;
    }
    if (tc >= 3 && tc <= 5) {
        score += 1.5;
        if (tr >= 3 && tr <= 6) score += 1.0;
    }
    if (type == PAWN) {
        if (side == 1 && tr < 5) score += 3.0;
        else if (side == -1 && tr >= 5) score += 3.0;
    }
    const int ok = side == 1 ? bk : rk;
    if (ok >= 0) {
        const int kr = ok / 9, kc = ok % 9;
        const int od = abs(fr - kr) + abs(fc - kc), nd = abs(tr - kr) + abs(tc - kc);
        if (nd < od) score += (od - nd) * 0.5;
    }
    return score;
}

struct MoveResult {
    double reward;
    int done, is_check, n_legal;
    uint64_t key;          // position_history entry appended by this move
};

// History access is abstracted so that the same cascade serves the real game (HBM history), the
// in-search envs (path-local history, starts empty: self_play.py:173-174) and the C-ABI rules
// entry points (host-provided history).
//   hist.count_key(key)      -> occurrences of key among the entries recorded so far (incl. new one)
//   hist.perpetual(is_check) -> >= 10 checks among the last 12 entries incl. the new one, len >= 12
//
// The board in LDS is updated in place; `legal` receives the legal moves of the NEW side to move
// when the game is not over by king capture (they double as the next position's move list).
//
// `hook(bd)` runs once, wave-convergent, as soon as the board holds the new position - before the check test and the
// move generation: the search kernel uses it to send the leaf's board, planes and table look on their way early.
// `check_now` (wave-uniform, with WANT_CHECK): compute :317's is_checking; the search skips it while no path can
// hold the 12 plies the perpetual-check rule reads (one instantiation instead of two: half the kernel's code).
struct NoHook { __device__ void operator()(const int8_t *) const {} };

template <bool WANT_REWARD, bool WANT_CHECK, class Hist, class Hook = NoHook>
__device__ MoveResult wave_make_move(int8_t *bd, MState &s, int move, Hist &hist, AttackMaps &M,
                                     uint16_t *cand, uint16_t *legal, uint8_t *own_sq, Hook hook = Hook(), bool check_now = true)
{
    const int lane = XQ_LANE;
    const int from = move / 90, to = move % 90;
    const int captured = bd[to], moving = bd[from];
    wave_sync();
    if (lane == 0) { bd[to] = (int8_t)moving; bd[from] = 0; }
    wave_sync();
    hook(bd);

    // (value selects, not conditional stores: `if (a) s.rk = x; else if (b) s.bk = x;` becomes a store through a selected
    // pointer, which keeps the whole MState in scratch memory - a memory round trip per access)
    s.rk = (captured == KING) ? NO_KING : ((moving == KING) ? to : s.rk);          // :271-279
    s.bk = (captured == -KING) ? NO_KING : ((moving == -KING) ? to : s.bk);
    s.nocap = (captured != 0) ? 0 : s.nocap + 1;                                   // :282-285

    MoveResult res;
    res.reward = 0; res.done = 0; res.n_legal = 0;
    const int acap = captured < 0 ? -captured : captured;
    if (acap == KING) {                                                            // :292-297
        s.winner = s.side; res.reward = 100; res.done = 1;
        s.reason = R_KING_CAPTURED; s.reason_side = s.side;
    } else if (WANT_REWARD && captured != 0) {                                     // :300-314
        double base = acap == ROOK ? 9 : acap == CANNON ? 4.5 : acap == KNIGHT ? 4 :
                      acap == BISHOP ? 2 : acap == ADVISOR ? 2 : acap == PAWN ? 1 : 0;
        res.reward = base * 2.0;
        if (acap == ADVISOR || acap == BISHOP) res.reward += 3.0;
    }

    BoardView v = load_view(bd);
    // every attack question of this move has the mover's pieces as the attackers: the check it gives (:317), the
    // next side's suicide filter (:431-464) and that side's in-check test (:625,641) - one set of maps
    const uint32_t kab_rows = build_attack_maps(M, v, s.side);
    int is_checking = 0;
    if (WANT_CHECK && check_now) is_checking = uni(in_check(M, kab_rows, s.side == 1 ? s.bk : s.rk, s.side) ? 1 : 0);   // :317
    res.is_check = is_checking;
    if (!res.done && is_checking) {                                                // :318-327
        if (WANT_REWARD) {
            if (s.cchk == 0) res.reward += 15.0;
            else if (s.cchk == 1) res.reward += 10.0;
            else if (s.cchk == 2) res.reward += 5.0;
        }
        s.cchk += 1;
    } else {                                                                       // :328-335
        s.cchk = 0;
        if (WANT_REWARD && captured == 0 && !res.done) {
            int tp = moving < 0 ? -moving : moving;
            double pr = position_change(tp, s.side, from, to, s.rk, s.bk);
            res.reward += pr * 0.01;
        }
    }

    res.key = hist.want_keys() ? position_key(bd, s.side == 1 ? 0 : 1) : 0ull;    // :338 (mover byte)
    hist.push(res.key, is_checking);                                               // :338-345

    s.side = -s.side;                                                              // :348-349
    s.move_count += 1;

    if (!res.done) {                                                               // :352-397
        const int nl = wave_movegen(bd, v, M, kab_rows, s.side, s.rk, s.bk, cand, legal, own_sq);
        res.n_legal = nl;
        const bool in_chk = (nl == 0) ? uni(in_check(M, kab_rows, s.side == 1 ? s.rk : s.bk, s.side) ? 1 : 0) != 0 : false;
        if (nl == 0 && in_chk) {                                                   // :354-359
            res.done = 1; res.reward = 200; s.winner = -s.side;
            s.reason = R_CHECKMATE; s.reason_side = s.side;
        } else if (hist.want_keys() && hist.count_key(position_key(bd, s.side == 1 ? 0 : 1)) >= 3) {   // :362-366, 598-605
            res.done = 1; res.reward = 0; s.winner = 0; s.reason = R_REPETITION; s.reason_side = 0;
        } else if (s.nocap >= 100) {                                               // :369-373, 612
            res.done = 1; res.reward = 0; s.winner = 0; s.reason = R_FIFTY; s.reason_side = 0;
        } else if (nl == 0) {                                                      // :376-381
            res.done = 1; res.reward = 100; s.winner = -s.side;
            s.reason = R_STALEMATE; s.reason_side = s.side;
        } else if (hist.perpetual()) {                                             // :384-389, 646-662
            res.done = 1; res.reward = -10; s.winner = -s.side;
            s.reason = R_PERP_CHECK; s.reason_side = s.side;
        }
    }
    if (!res.done && s.move_count >= 70) {                                         // :400-404
        res.done = 1; res.reward = -2; s.winner = 0;
        s.reason = R_MOVE_CAP; s.reason_side = 0; s.reason_count = s.move_count;
    }
    return res;
}

}  // namespace xq

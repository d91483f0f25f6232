// xq_attack.hpp — king-centric attack test for the legality filter and the in-check test (gfx950; also plain C++).
//
// Reference behaviour restated: chess_env.py:431-464 (_is_move_suicide: apply the move on a copy, test the own king),
// :466-495 (_are_kings_facing on the cached king squares), :506-548 (_is_in_check: regenerate every enemy piece's
// pseudo-moves and look for the king square), with the generators of :123-251 keyed on current_player (SURVEY.md
// Appendix A1) and the cache rules of Appendix A5 / A6.
//
// The reference asks "which squares does every enemy piece reach?" (16 generator calls per candidate move); rounds 1-4
// asked the same question per enemy piece in closed form (a wave-uniform loop over ~16 pieces, one scalar branch and
// ~20 vector instructions each).  This file asks it from the king's side: from square k, on the occupancy the candidate
// move leaves behind, which squares could hold a rook / cannon / knight / pawn / king / advisor / bishop that reaches
// k - and is one there?  One pass, no loop over pieces, no scalar branches:
//   * sliders: the first and second occupied square on each of the four rays from k; a rook attacks from a first one, a
//     cannon from a second one when k is occupied and from a first one when it is empty (chess_env.py:199-235);
//   * knights: the 4 diagonal neighbours of k are the legs, each serving two knight squares (:178-197);
//   * pawns / kings / advisors / bishops under the rules of side X (A1: the generators read current_player, not the
//     piece's colour), bishops with the river bound of :159-170 and the eye = the same diagonal neighbour.
// The attackers are looked up in per-row bit masks by type (AttackMaps, built once per position from the board the
// candidate moves start from); the piece a candidate captures is taken out by clearing its bit per row.
//
// Everything here is plain integer code on arrays: tests/test_attack_cpu.py compiles it with g++ and checks it against
// the CPU oracle on random (also inconsistent: stale king caches, any piece counts) positions before a GPU sees it.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define XQ_HD __host__ __device__ __forceinline__
#else
#define XQ_HD static inline
#endif

namespace xq {

// Row r lives at index r + 2 (rows -2, -1, 10, 11 stay zero: probes beside the board read "nothing there"); column c
// at bit c + 2 (bits 0, 1, 11, 12 likewise); the second piece type of a table 16 bits up.  Column tables: column c at
// index c, row r at bit r.  All entries are dwords so that a wave can build them with ds_or_b32.
struct AttackMaps {
    uint32_t occrow[14];     // every piece
    uint32_t occcol[12];
    uint32_t rc_row[14];     // attackers: rooks | cannons << 16
    uint32_t rc_col[12];
    uint32_t np_row[14];     // attackers: knights | pawns << 16
    uint32_t ka_row[14];     // attackers: kings | advisors << 16
    uint32_t b_row[14];      // attackers: bishops
};
enum : int { ATTACK_MAP_DWORDS = sizeof(AttackMaps) / 4 };

// (table offset in dwords, bit shift) of the row table that holds attackers of type T = 1..7 (K A B N R C P)
XQ_HD int attack_row_table(int T)
{
    return T == 5 || T == 6 ? 26 : (T == 4 || T == 7 ? 52 : (T == 3 ? 80 : 66));
}
XQ_HD int attack_type_shift(int T) { return (T == 2 || T == 6 || T == 7) ? 16 : 0; }

// host-side builder (tests; the device builds the same tables with LDS atomics, xq_device.hpp)
static inline void build_attack_maps_host(AttackMaps &M, const int8_t *bd, int att_colour)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(&M);
    for (int i = 0; i < ATTACK_MAP_DWORDS; i++) w[i] = 0;
    for (int s = 0; s < 90; s++) {
        const int p = bd[s];
        if (!p) continue;
        const int r = s / 9, c = s - 9 * r;
        M.occrow[r + 2] |= 4u << c;
        M.occcol[c] |= 1u << r;
        if (p * att_colour > 0) {
            const int T = p < 0 ? -p : p;
            if (T < 1 || T > 7) continue;
            w[attack_row_table(T) + r + 2] |= (4u << c) << attack_type_shift(T);
            if (T == 5 || T == 6) M.rc_col[c] |= (1u << r) << attack_type_shift(T);
        }
    }
}

XQ_HD uint32_t top_bit(uint32_t x)          // highest set bit of x as a mask, 0 for 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return x ? 0x80000000u >> __builtin_clz(x) : 0u;
#else
    return x ? 0x80000000u >> __builtin_clz(x) : 0u;
#endif
}

// first | (cannon's choice << 16) for one line: `line` = occupancy bits of the line, `pos` = bit index of k on it.
// A rook reaches k from the first occupied square on either side; a cannon from the second one when k itself is
// occupied (exactly one screen between) and from the first one when k is empty (chess_env.py:215-235: the ray walks
// over empty squares until it meets the screen, then captures the next piece).
XQ_HD uint32_t slider_sources(uint32_t line, int pos)
{
    const uint32_t below = line & ((1u << pos) - 1u), above = line & ~((2u << pos) - 1u);
    const uint32_t r1 = above & (0u - above), a2 = above ^ r1, r2 = a2 & (0u - a2);
    const uint32_t l1 = top_bit(below), l2 = top_bit(below ^ l1);
    const uint32_t first = l1 | r1, second = l2 | r2;
    return first | (((line >> pos) & 1u ? second : first) << 16);
}

// Is square k attacked by a piece of the maps' colour, other than the one on `t`, after the piece on `f` has moved to
// `t` (MOVE) / on the board as it stands (!MOVE: f, t ignored), under the generator rules of side X?  `other_king` =
// the cached square of the other king or -1: with MOVE the answer also covers _are_kings_facing between k and it
// (k is the mover's cached king square, already moved when the king is the piece that moves - A5).
// `kab` (wave-uniform on the device): kings / advisors / bishops of the attackers stand where they could matter; when
// false their tables are not read (never true in a game that started from the initial position: A1 makes them harmless).
template <bool MOVE>
XQ_HD bool king_attacked(const AttackMaps &M, int k, int f, int t, int X, int other_king, bool kab)
{
    if (k < 0) return false;                                    // no king: nothing to attack, nothing to face
    const int kr = (k * 57) >> 9, kc = k - 9 * kr, cb = kc + 2;
    uint32_t rowU = M.occrow[kr + 1], rowK = M.occrow[kr + 2], rowD = M.occrow[kr + 3], colK = M.occcol[kc];
    int dtr = -100, tr = 0, tc = 0;
    uint32_t tb = 0;
    if (MOVE) {
        const int fr = (f * 57) >> 9, fc = f - 9 * fr;
        tr = (t * 57) >> 9; tc = t - 9 * tr;
        const uint32_t fb = 4u << fc;
        tb = 4u << tc;
        const int dfr = fr - kr;
        dtr = tr - kr;
        rowU = (rowU & ~(dfr == -1 ? fb : 0u)) | (dtr == -1 ? tb : 0u);
        rowK = (rowK & ~(dfr == 0 ? fb : 0u)) | (dtr == 0 ? tb : 0u);
        rowD = (rowD & ~(dfr == 1 ? fb : 0u)) | (dtr == 1 ? tb : 0u);
        colK = (colK & ~(fc == kc ? 1u << fr : 0u)) | (tc == kc ? 1u << tr : 0u);
    }
    // sliders
    uint32_t h0 = M.rc_row[kr + 2] & slider_sources(rowK, cb);
    uint32_t hc = M.rc_col[kc] & slider_sources(colK, kr);
    // knights: legs = the diagonal neighbours; bit 0 / bit 2 of fu, fd = left / right leg of the row above / below is free
    const uint32_t fu = (~rowU >> (cb - 1)) & 5u, fd = (~rowD >> (cb - 1)) & 5u;
    const uint32_t su = (fu & 1u) | ((fu & 4u) << 2), sd = (fd & 1u) | ((fd & 4u) << 2);      // bits 0 / 4: two columns out
    // pawns (A1: under X's rules whatever their colour): the square "behind" k, and beside it once X has crossed
    const uint32_t pf = 0x10000u << cb;
    const bool crossed = (X == 1) ? (kr < 5) : (kr >= 5);
    uint32_t hu2 = M.np_row[kr] & (fu << (cb - 1));
    uint32_t hu1 = M.np_row[kr + 1] & ((su << (cb - 2)) | (X == 1 ? 0u : pf));
    h0 |= M.np_row[kr + 2] & (crossed ? 0x50000u << (cb - 1) : 0u);
    uint32_t hd1 = M.np_row[kr + 3] & ((sd << (cb - 2)) | (X == 1 ? pf : 0u));
    uint32_t hd2 = M.np_row[kr + 4] & (fd << (cb - 1));
    if (kab) {
        const bool pal = ((X == 1) ? (kr >= 7) : (kr <= 2)) && kc >= 3 && kc <= 5;             // k inside X's palace
        const uint32_t kbit = pal ? 1u << cb : 0u, side3 = pal ? 5u << (cb - 1) : 0u;
        h0 |= M.ka_row[kr + 2] & side3;                                                         // kings beside k
        hu1 |= M.ka_row[kr + 1] & (kbit | (side3 << 16));                                      // king above, advisors diagonal
        hd1 |= M.ka_row[kr + 3] & (kbit | (side3 << 16));
        const bool river = !((X == 1) ? (kr < 5) : (kr >= 4));                                  // chess_env.py:159-170
        hu2 |= M.b_row[kr] & (river ? su << (cb - 2) : 0u);                                     // eye = the leg square
        hd2 |= M.b_row[kr + 4] & (river ? sd << (cb - 2) : 0u);
    }
    bool facing = false;
    if (MOVE) {
        // the captured piece no longer attacks: one square, i.e. one bit of one row (and of the column table)
        const uint32_t tb2 = tb | (tb << 16);
        hu2 &= ~(dtr == -2 ? tb2 : 0u);
        hu1 &= ~(dtr == -1 ? tb2 : 0u);
        h0 &= ~(dtr == 0 ? tb2 : 0u);
        hd1 &= ~(dtr == 1 ? tb2 : 0u);
        hd2 &= ~(dtr == 2 ? tb2 : 0u);
        hc &= ~(tc == kc ? 0x10001u << tr : 0u);
        // _are_kings_facing (chess_env.py:466-495) on the cached squares: same column, nothing between
        if (other_king >= 0) {
            const int orow = (other_king * 57) >> 9, ocol = other_king - 9 * orow;
            const int lo = orow < kr ? orow : kr, hi = orow < kr ? kr : orow;
            const uint32_t between = hi - lo > 1 ? ((1u << (hi - lo - 1)) - 1u) << (lo + 1) : 0u;
            facing = ocol == kc && (colK & between) == 0u;
        }
    }
    return ((hu2 | hu1 | h0 | hd1 | hd2 | hc) != 0u) || facing;
}

}  // namespace xq

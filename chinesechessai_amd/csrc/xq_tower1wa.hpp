// xq_tower1wa.hpp - k_tower1wa: the trunk kernel with ONE wave per SIMD whose residual tower is hand-written gfx950 assembly.
// Included by xq_tower.hip inside its translation unit (TowerArgs, the LDS constants, xq_mfma.hpp).
//
// A 256-thread workgroup carries 4 boards; each wave owns a whole board and all 128 output channels of it (wave tile 128
// channels x 96 pixels = 8 x 6 tiles of v_mfma_f32_16x16x32_bf16, 192 accumulator registers on a[0:191]), one workgroup per
// CU, one weight stream per 4 boards through a ring of 4 stages.  Round 3 built this tile in HIP (k_tower1w,
// xq_tower_probes.hpp): bit-identical to k_tower16b, 2.15-2.31 GHz, and level with the default because nothing covers an
// epilogue when a SIMD has one wave.  Here the residual tower - every layer's main loop, epilogue, weight DMA, stage
// barriers and the loop over the blocks - is one asm statement (xq_tower1wa_body.inc, generated and checked by
// tools/gen_tower1wa.py), in which a layer's epilogue runs UNDER the next layer's first tap: the MFMAs of tap 0 issue pair
// by pair as the accumulators they write are drained and the rows they read are stored.  Every accumulator still sees
// (bias [+ x], tap 0: ks 0..3, tap 1: ...) in k_tower16b's order: the results are the same bits.
// The parts outside the statement (input convolution, heads) are k_tower1w's HIP code: accumulators on literal AGPRs.
// The lane-dependent addresses the assembly needs (54 tap addresses, store addresses, selectors, DMA source offsets) are a
// compile-time table (g_lane_tab), 76 dwords per thread, loaded once per workgroup.
// Reference ops: neural_network.py:54-66,181-187 with eval-mode BatchNorm folded.
#pragma once
#include "xq_tower1wa_body.inc"
#if XQ_TOWER_PROBES
#include "xq_tower1wa_abl.inc"      // timing-only bodies (wrong results): python tools/gen_tower1wa.py --ablations
#endif

namespace {

// ---- accumulators on literal AGPRs (shared with the round-3 experiment k_tower1w, xq_tower_probes.hpp) -----------------------
// hipcc cannot keep 192 accumulator registers in AGPRs by itself: for __builtin_amdgcn_mfma_* it selects untied AGPR-form
// MFMAs and rotates the accumulators through staging tuples.  In the HIP parts of these kernels every MFMA, accumulator
// read / write and bias load is therefore an asm statement on LITERAL registers: tile t lives in a[4t : 4t + 3], the compiler
// never sees these registers; XQ_AGPR_ALL (one statement at kernel entry) makes it account for them in the kernel
// descriptor.  Nothing stops hipcc from spilling its own values into the same registers if it ran out of VGPRs, so every
// build is checked by tools/scan_tower1w_isa.py (no compiler-generated AGPR use while the accumulators are live).
// What hipcc does NOT do for an asm MFMA: insert the wait states a VALU-written source needs (amfma_guarded) or the ones
// between an MFMA and a v_accvgpr_read of its result (s_nop block at the head of an epilogue).
#define XQ_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define XQ_AGPR_ALL() asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", XQ_A8(1), XQ_A8(2), XQ_A8(3), XQ_A8(4), \
    XQ_A8(5), XQ_A8(6), XQ_A8(7), XQ_A8(8), XQ_A8(9), XQ_A8(10), XQ_A8(11), XQ_A8(12), XQ_A8(13), XQ_A8(14), XQ_A8(15), XQ_A8(16), \
    XQ_A8(17), XQ_A8(18), "a190", "a191")
// (the f32x4 & parameter is the compiler-visible stand-in of the tile: unused by this form)
template <int T> __device__ __forceinline__ void amfma(f32x4 &, const bf16x8 &a, const bf16x8 &b)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(4 * T), "n"(4 * T + 3), "v"(a), "v"(b));
}
template <int T> __device__ __forceinline__ void amfma_guarded(f32x4 &, const bf16x8 &a, const bf16x8 &b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(4 * T), "n"(4 * T + 3), "v"(a), "v"(b));
}
template <int T> __device__ __forceinline__ void aset(f32x4 &, const f32x4 &v)          // a[4T .. 4T+3] = v
{
    asm volatile("v_accvgpr_write_b32 a[%0], %4\n\tv_accvgpr_write_b32 a[%1], %5\n\tv_accvgpr_write_b32 a[%2], %6\n\tv_accvgpr_write_b32 a[%3], %7"
                 : : "n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
}
template <int T> __device__ __forceinline__ f32x4 aget(const f32x4 &)
{
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3));
    return f32x4{ x0, x1, x2, x3 };
}
// a[4T .. 4T+3] = 16 bytes of LDS at addr + OFF, straight into the accumulator (a layer's bias as its start value, no VALU
// instruction).  An asm load is outside hipcc's s_waitcnt bookkeeping: the caller waits with await_lds() before the
// registers are used; the compiler's own counted waits stay correct (extra operations in the in-order LDS queue only make
// them conservative).
template <int T, int OFF> __device__ __forceinline__ void aload(f32x4 &, int addr)
{
    asm volatile("ds_read_b128 a[%0:%1], %2 offset:%3" : : "n"(4 * T), "n"(4 * T + 3), "v"(addr), "n"(OFF) : "memory");
}
// compile-time loops over tiles
template <int N0, int... Ms> __device__ __forceinline__ void amfma_col(std::integer_sequence<int, Ms...>, f32x4 *acc, const bf16x8 *fa, const bf16x8 &fb)
{
    (amfma<Ms * 6 + N0>(acc[Ms * 6 + N0], fa[Ms], fb), ...);
}
template <int... Ts> __device__ __forceinline__ void aset_all(std::integer_sequence<int, Ts...>, f32x4 *acc, const f32x4 &v) { (aset<Ts>(acc[Ts], v), ...); }

// one MFMA of weight tile M on pixel tile n (n is a constant after unrolling: the switch folds away)
template <int M> __device__ __forceinline__ void amfma_n(f32x4 *acc, int n, const bf16x8 &a, const bf16x8 &b)
{
    switch (n) {
    case 0: amfma<M * 6 + 0>(acc[M * 6 + 0], a, b); break;
    case 1: amfma<M * 6 + 1>(acc[M * 6 + 1], a, b); break;
    case 2: amfma<M * 6 + 2>(acc[M * 6 + 2], a, b); break;
    case 3: amfma<M * 6 + 3>(acc[M * 6 + 3], a, b); break;
    case 4: amfma<M * 6 + 4>(acc[M * 6 + 4], a, b); break;
    default: amfma<M * 6 + 5>(acc[M * 6 + 5], a, b); break;
    }
}

template <int N0, int... Ms> __device__ __forceinline__ void aset_col(std::integer_sequence<int, Ms...>, f32x4 *acc, const f32x4 *b)
{
    (aset<Ms * 6 + N0>(acc[Ms * 6 + N0], b[Ms]), ...);
}
// two floats -> packed bf16 + ReLU without an asm statement (hipcc selects v_cvt_pk_bf16_f32 for the vector conversion and
// can schedule it; the asm form of xq_mfma.hpp costs a boundary s_nop per use)
__device__ __forceinline__ uint32_t pack_relu_bf16x2(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) short s16x2_t;
    const bf16x2_t h = __builtin_convertvector((f32x2_t){ a, b }, bf16x2_t);
    s16x2_t v = *reinterpret_cast<const s16x2_t *>(&h);
    v = __builtin_elementwise_max(v, (s16x2_t){ 0, 0 });
    return *reinterpret_cast<const uint32_t *>(&v);
}
__device__ __forceinline__ void await_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// epilogue of pixel tile N of channel pair J: the lane's 8 consecutive channels (tiles 2J, 2J + 1) -> bf16 -> ReLU -> 16 bytes
template <int J, int N> __device__ __forceinline__ uint4 epi_get(const f32x4 *acc)
{
    const f32x4 v0 = aget<(2 * J) * 6 + N>(acc[(2 * J) * 6 + N]), v1 = aget<(2 * J + 1) * 6 + N>(acc[(2 * J + 1) * 6 + N]);
    return make_uint4(pack_relu_bf16x2(v0[0], v0[1]), pack_relu_bf16x2(v0[2], v0[3]),
                      pack_relu_bf16x2(v1[0], v1[1]), pack_relu_bf16x2(v1[2], v1[3]));
}
// first half of a pair's epilogue: [x fragments in,] results out, next layer's bias into the accumulators
template <int J, bool READ_X, int... Ns>
__device__ __forceinline__ void epi_pair(std::integer_sequence<int, Ns...>, f32x4 *acc, const int *sb, bool tail_ok, int lbq, bf16x8 *xf)
{
    if constexpr (READ_X) ((xf[Ns] = lds_ld128(sb[Ns] ^ (J << 5))), ...);          // the block input x: the chunk this lane overwrites
    uint4 pk[6];
    ((pk[Ns] = epi_get<J, Ns>(acc)), ...);
    ((Ns < 5 || tail_ok ? lds_st128(sb[Ns] ^ (J << 5), pk[Ns]) : (void)0), ...);
    ((aload<(2 * J) * 6 + Ns, J * 128>(acc[(2 * J) * 6 + Ns], lbq), aload<(2 * J + 1) * 6 + Ns, J * 128 + 16>(acc[(2 * J + 1) * 6 + Ns], lbq)), ...);
}
// second half (first convolution of a block): + x through the matrix pipe, S . X with a 0/1 selector S
template <int J, int... Ns>
__device__ __forceinline__ void epi_skip(std::integer_sequence<int, Ns...>, f32x4 *acc, const bf16x8 *sel, const bf16x8 *xf)
{
    ((amfma_guarded<(2 * J) * 6 + Ns>(acc[(2 * J) * 6 + Ns], sel[0], xf[Ns]), amfma_guarded<(2 * J + 1) * 6 + Ns>(acc[(2 * J + 1) * 6 + Ns], sel[1], xf[Ns])), ...);
}


constexpr int RING1WA = 4;
constexpr int LDS_BYTES1WA = XQ_1WA_LDS_BYTES;
static_assert(XQ_1WA_LDS_BYTES <= 163840, "LDS image of k_tower1wa");
static_assert(XQ_1WA_JUNK >= 4 * WBUF_BYTES + 4 * ACT_BYTES + 256 + 2 * 512 && (XQ_1WA_JUNK & 1023) == 0, "junk rows behind the bias table");

struct LaneTab1WA { int32_t v[256][XQ_1WA_TAB_DWORDS]; };

// the per-thread constants of the assembly body, in the register order the body loads them (tools/gen_tower1wa.py:
// V_TA .. V_L16); formulas = k_tower1w's (tap_addr, abase, sb, lbq, sel, wsrc)
constexpr LaneTab1WA make_lane_tab_1wa()
{
    LaneTab1WA T{};
    constexpr int ACT0 = 4 * WBUF_BYTES, ZROW = ACT0 + 4 * ACT_BYTES, BIAS = ZROW + 256;
    for (int tid = 0; tid < 256; tid++) {
        const int wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
        const int act_off = ACT0 + wave * ACT_BYTES;
        int32_t *o = T.v[tid];
        int k = 0;
        const bool real5 = r16 < PIX - 80, yu0 = r16 >= 9, yd5 = r16 == 0;
        const int Rrow = act_off + r16 * 256, r5 = r16 << 4, q4 = (((q & 1) << 3) | (q >> 1)) << 4;
        for (int tap = 0; tap < 9; tap++)
            for (int nt = 0; nt < 6; nt++) {
                const int p = nt * 16 + r16, xx = p % 9;
                const bool xl = xx != 0 && p < PIX, xr = xx != 8 && p < PIX;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1, off = dy * 9 + dx;
                const int slot = ((r5 + off * 16) & 0x70) ^ q4;
                const int aok = Rrow + off * 256 + slot;
                const bool sel = dx != 0 || (nt == 0 && dy < 0) || nt == 5;
                int a = aok;
                if (sel) {
                    bool ok = dx < 0 ? xl : dx > 0 ? xr : real5;
                    if (dx == 0 && nt == 0) ok = yu0;
                    else if (nt == 0 && dy < 0) ok = ok && yu0;
                    if (nt == 5 && dy > 0) ok = ok && yd5;
                    a = ok ? aok : slot + (ZROW - nt * 4096);
                }
                o[k++] = a;
            }
        const int abase = r16 * 128 + ((q ^ ((r16 >> 1) & 7)) << 4);
        o[k++] = abase;
        o[k++] = abase ^ 64;
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r16 < PIX ? nt * 16 + r16 : 0;
            o[k++] = act_off + p * 256 + (((((q & 1) << 3) | (q >> 1)) ^ (p & 7)) << 4);
        }
        o[k] = 80 + r16 < PIX ? o[k - 1] : XQ_1WA_JUNK + lane * 16;      // store address of pixel tile 5
        k++;
        o[k++] = BIAS + q * 32;
        for (int sidx = 0; sidx < 2; sidx++)
            for (int d = 0; d < 4; d++) {
                uint32_t w = 0;
                for (int h = 0; h < 2; h++) {
                    const int j = 2 * d + h;
                    if (q == (r16 >> 2) && j == sidx * 4 + (r16 & 3)) w |= 0x3F80u << (16 * h);
                }
                o[k++] = (int32_t)w;
            }
        const int wch = (lane & 7) ^ (lane >> 4);
        const int wsrc_even = ((lane >> 5) * 8 + ((lane >> 3) & 3)) * 256 + (wch << 4);
        o[k++] = wsrc_even;
        o[k++] = wsrc_even ^ 64;
        o[k++] = lane * 16;
        o[k++] = 0;
    }
    return T;
}
__device__ const LaneTab1WA g_lane_tab_1wa = make_lane_tab_1wa();

#define XQ_S8(b) "s" #b "0", "s" #b "1", "s" #b "2", "s" #b "3", "s" #b "4", "s" #b "5", "s" #b "6", "s" #b "7", "s" #b "8", "s" #b "9"
#define XQ_V8(b) "v" #b "0", "v" #b "1", "v" #b "2", "v" #b "3", "v" #b "4", "v" #b "5", "v" #b "6", "v" #b "7", "v" #b "8", "v" #b "9"
#define XQ_1WA_CLOBBERS                                                                                                        \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", XQ_V8(1), XQ_V8(2), XQ_V8(3), XQ_V8(4), XQ_V8(5), XQ_V8(6), XQ_V8(7),        \
    XQ_V8(8), XQ_V8(9), XQ_V8(10), XQ_V8(11), XQ_V8(12), XQ_V8(13), XQ_V8(14), XQ_V8(15), XQ_V8(16), XQ_V8(17), XQ_V8(18), XQ_V8(19),   \
    XQ_V8(20), XQ_V8(21), XQ_V8(22), XQ_V8(23), XQ_V8(24), "v250", "v251", "v252", "v253",                                           \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", XQ_A8(1), XQ_A8(2), XQ_A8(3), XQ_A8(4), XQ_A8(5), XQ_A8(6), XQ_A8(7),        \
    XQ_A8(8), XQ_A8(9), XQ_A8(10), XQ_A8(11), XQ_A8(12), XQ_A8(13), XQ_A8(14), XQ_A8(15), XQ_A8(16), XQ_A8(17), XQ_A8(18), XQ_A8(19), \
    XQ_A8(20), XQ_A8(21), "a220", "a221", "a222", "a223",                                                                        \
    "s36", "s37", "s38", "s39", XQ_S8(4), XQ_S8(5), "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "vcc", "scc", "memory"
static_assert(XQ_1WA_V_LAST == 253 && XQ_1WA_A_LAST == 223 && XQ_1WA_S_FIRST == 36 && XQ_1WA_S_LAST == 69, "clobber list of the layer body");

// ABL (probes builds, stamped, wrong results): one of the timing-only bodies of xq_tower1wa_abl.inc
template <bool STAMP, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_tower1wa(TowerArgs A)
{
    constexpr int NB = 4, PPW = 4;                                       // boards = waves, weight pieces per wave and stage
    constexpr int ACT0 = RING1WA * WBUF_BYTES, ZROW = ACT0 + NB * ACT_BYTES, BIAS = ZROW + 256;   // bias: [2][128] f32
    using seq8 = std::make_integer_sequence<int, 8>;

    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) A.stamps[(size_t)blockIdx.x * 64 + slot] = t;
            if (slot == 0 || slot == 61) {
                const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
                if (threadIdx.x == 0) A.stamps[(size_t)blockIdx.x * 64 + (slot == 0 ? 62 : 63)] = rt;
            }
        }
    };
    stamp(0);
    // clock sample (xq_tower_set_clock_sample; product build only): one workgroup in 64 adds its shader cycles and its
    // 100 MHz ticks to three counters; bench.py divides them - the clock the chip held while this kernel ran
    unsigned long long clk0 = 0, rt0 = 0;
    const bool clk_sample = !STAMP && A.stamps != nullptr && (blockIdx.x & 63) == 0;
    if (clk_sample) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    XQ_AGPR_ALL();
    f32x4 acc[48];                                                     // stand-ins of the tiles: tile (mt, n) = acc[mt * 6 + n] lives in a[4t : 4t + 3]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto chan_row = [&](int mt, int i) { return (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3); };
    int nrows = A.G;
    if (A.n_rows) { const int n = *A.n_rows; nrows = n < A.G ? n : A.G; }
    const int act_off = ACT0 + wave * ACT_BYTES;
    const int r16 = lane & 15, q = lane >> 4;
    const int pl_off = act_off + ACT_BYTES - PIX * 32;
    const int nlayers = 2 * A.nblocks, nstages = nlayers * 18;
    const int quad = blockIdx.x;
    if (quad * NB >= nrows) return;
    const int board = quad * NB + wave;
    const bool board_ok = board < nrows;

    // ---------------------------------------------------------------- input conv (16 -> 128): k_tower1w's
    if (tid < 16) lds_st128(ZROW + tid * 16, make_uint4(0, 0, 0, 0));
    if (wave == 1 && lane < 32 && nstages > 0) dma16_abs(A.bias + 128 + lane * 4, BIAS + 512);   // bias[1] -> slot 1
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int piece = j * 4 + wave;
        dma16_abs(reinterpret_cast<const uint8_t *>(A.w1) + piece * 1024 + lane * 16, piece * 1024);
    }
    if (board_ok) {
        const int srow = A.row_src ? A.row_src[board] : board;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.planes) + (size_t)srow * PIX * 32;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int idx = j * 64 + lane;
            if (idx < PIX * 2) dma16_abs(src + idx * 16, pl_off + j * 1024);
        }
    }
    {
        f32x4 b8[8];
#pragma unroll
        for (int mt = 0; mt < 8; mt++) b8[mt] = *reinterpret_cast<const f32x4 *>(A.bias + chan_row(mt, 4 * q));
        aset_col<0>(seq8{}, acc, b8); aset_col<1>(seq8{}, acc, b8); aset_col<2>(seq8{}, acc, b8);
        aset_col<3>(seq8{}, acc, b8); aset_col<4>(seq8{}, acc, b8); aset_col<5>(seq8{}, acc, b8);
    }
    barrier_dma();
    {
        uint32_t vm[2] = { 0, 0 };        // tap validity of the 6 pixels of this lane, 9 bits each
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int o = nt * 16 + r16;
            uint32_t m = 0;
            if (o < PIX) {
                const int yy = o / 9, xx = o % 9;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    if (yy + dy >= 0 && yy + dy < 10 && xx + dx >= 0 && xx + dx < 9) m |= 1u << t;
                }
            }
            vm[nt / 3] |= m << ((nt % 3) * 9);
        }
#pragma unroll
        for (int pair = 0; pair < 5; pair++) {                       // K-step = taps (2 pair, 2 pair + 1) x 16 planes
            const int tap = 2 * pair + (q >> 1);
            const bool tap_real = tap < 9;
            const int tp = tap_real ? tap : 8;
            const int off = (tp / 3 - 1) * 9 + (tp % 3 - 1);
            bf16x8 bf[6], af[8];
#pragma unroll
            for (int nt = 0; nt < 6; nt++) {
                const bool ok = tap_real && ((vm[nt / 3] >> ((nt % 3) * 9 + tp)) & 1u);
                const int sp = nt * 16 + r16 + off;
                bf[nt] = lds_ld128((ok ? pl_off + sp * 32 : ZROW + (sp & 7) * 32) + (q & 1) * 16);
            }
#pragma unroll
            for (int mt = 0; mt < 8; mt++)
                af[mt] = lds_ld128((tp * COUT + chan_row(mt, r16)) * 32 + (q & 1) * 16);
            __builtin_amdgcn_sched_barrier(0);
            amfma_col<0>(seq8{}, acc, af, bf[0]); amfma_col<1>(seq8{}, acc, af, bf[1]); amfma_col<2>(seq8{}, acc, af, bf[2]);
            amfma_col<3>(seq8{}, acc, af, bf[3]); amfma_col<4>(seq8{}, acc, af, bf[4]); amfma_col<5>(seq8{}, acc, af, bf[5]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    barrier_dma();                                                 // every wave is done with the tap slices (ring area)
    stamp(1);

    // weight stream: stage g = (layer, tap, K-slice) -> ring slot g & 3, 16 pieces of 1 KB, 4 per wave; stages 0..2 from here,
    // the rest from the layer body
    if (nstages > 0) {
        const rsrc_t wrsrc = make_rsrc(A.wt, nlayers * 9 * COUT * COUT * 2);
        const int wch = (lane & 7) ^ (lane >> 4);
        const int wsrc_even = ((lane >> 5) * 8 + ((lane >> 3) & 3)) * 256 + (wch << 4), wsrc_odd = wsrc_even ^ 64;
        auto piece_off = [&](int P) { return ((P >> 3) * 64 + ((P >> 2) & 1) * 32 + (P & 1) * 16 + ((P >> 1) & 1) * 4) * 256; };
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int j = 0; j < PPW; j++) {
                const int soff = (g >> 1) * (COUT * COUT * 2) + (g & 1) * 128 + piece_off(wave * PPW + j);
                dma16_buf_abs(wrsrc, (j & 1) ? wsrc_odd : wsrc_even, soff, (g & 3) * WBUF_BYTES + (wave * PPW + j) * 1024);
            }
    }
    // Round 5: with a tower behind it, the input convolution's epilogue is the assembly body's first drain - under tap 0 of
    // layer 0, like every other layer boundary (tools/gen_tower1wa.py sec_pro) - and the statement starts right here: the
    // accumulators hold conv + bias[0], stages 0..2 are in flight (its head waits for this wave's pieces, its first barrier
    // publishes everybody's), bias[1] landed before the barrier above.  Without a tower (0 blocks) the heads read the
    // input convolution's output: the HIP epilogue below.
    if (nstages == 0) {
        // epilogue of the input convolution (wave-local): acc -> bf16 -> ReLU -> LDS rows
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int r = ln & 15, qq = ln >> 4;
        int sb[6];
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r < PIX ? nt * 16 + r : 0;
            sb[nt] = act_off + p * 256 + (((((qq & 1) << 3) | (qq >> 1)) ^ (p & 7)) << 4);
        }
        const int lbq = BIAS + 512 + qq * 8 * 4;
        const bool tail_ok = r < PIX - 80;
        using seq6 = std::make_integer_sequence<int, 6>;
        asm volatile("s_nop 15\n\ts_nop 15");                          // the last MFMAs -> v_accvgpr_read (no interlock for asm)
        epi_pair<0, false>(seq6{}, acc, sb, tail_ok, lbq, nullptr);
        epi_pair<1, false>(seq6{}, acc, sb, tail_ok, lbq, nullptr);
        epi_pair<2, false>(seq6{}, acc, sb, tail_ok, lbq, nullptr);
        epi_pair<3, false>(seq6{}, acc, sb, tail_ok, lbq, nullptr);
        await_lds();
        barrier_dma();
    }
    stamp(2);

    // ---------------------------------------------------------------- residual tower: the assembly body
    if (nstages > 0) {
        const int32_t *tab = &g_lane_tab_1wa.v[0][0];
        const float *bias2 = A.bias + 256;                              // row 2 = tower layer 1
        const unsigned long long *st = STAMP ? A.stamps + (size_t)blockIdx.x * 64 + 3 : nullptr;
        const int wt_bytes = nlayers * 9 * COUT * COUT * 2;
#if XQ_TOWER_PROBES
#define XQ_1WA_RUN_ABL(K) asm volatile(XQ_1WA_BODY_ABL##K : : "s"(tab), "s"(A.wt), "s"(wt_bytes), "s"(bias2), "s"(A.nblocks), "s"(wave), "s"(st), "v"(tid) : XQ_1WA_CLOBBERS)
        if constexpr (ABL == 1) XQ_1WA_RUN_ABL(1);
        else if constexpr (ABL == 2) XQ_1WA_RUN_ABL(2);
        else if constexpr (ABL == 3) XQ_1WA_RUN_ABL(3);
        else if constexpr (ABL == 4) XQ_1WA_RUN_ABL(4);
        else if constexpr (ABL == 5) XQ_1WA_RUN_ABL(5);
        else if constexpr (ABL == 6) XQ_1WA_RUN_ABL(6);
        else if constexpr (ABL == 7) XQ_1WA_RUN_ABL(7);
        else if constexpr (ABL == 8) XQ_1WA_RUN_ABL(8);
        else if constexpr (ABL == 9) XQ_1WA_RUN_ABL(9);
        else if constexpr (ABL == 10) XQ_1WA_RUN_ABL(10);
        else if constexpr (ABL == 11) XQ_1WA_RUN_ABL(11);
        else if constexpr (ABL == 12) XQ_1WA_RUN_ABL(12);
        else if constexpr (ABL == 13) XQ_1WA_RUN_ABL(13);
        else
#endif
        if constexpr (STAMP)
            asm volatile(XQ_1WA_BODY_STAMPED
                         :
                         : "s"(tab), "s"(A.wt), "s"(wt_bytes), "s"(bias2), "s"(A.nblocks), "s"(wave), "s"(st), "v"(tid)
                         : XQ_1WA_CLOBBERS);
        else
            asm volatile(XQ_1WA_BODY
                         :
                         : "s"(tab), "s"(A.wt), "s"(wt_bytes), "s"(bias2), "s"(A.nblocks), "s"(wave), "s"(st), "v"(tid)
                         : XQ_1WA_CLOBBERS);
    }
    stamp(59);

    // ---------------------------------------------------------------- heads (1x1, 128 -> 32 + 8): k_tower1w's
    __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0): no tower DMA in flight, own reads done
    __syncthreads();                                                 // ... for every wave: the ring area is free
    {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.wh);     // [64][256 B], chunk ^ ((row & 7) << 1)
#pragma unroll
        for (int j = 0; j < PPW; j++) {
            const int q0 = (wave * PPW + j) * 64, idx = q0 + lane, row = idx >> 4, cp = idx & 15;
            dma16_abs(src + row * 256 + ((cp ^ ((row & 7) << 1)) * 16), q0 * 16);
        }
    }
    f32x4 hacc[3][6];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int nt = 0; nt < 6; nt++)
#pragma unroll
            for (int i = 0; i < 4; i++) hacc[m][nt][i] = 0.f;
    barrier_dma();
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        bf16x8 hb[6], ha[3];
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r16 < PIX ? nt * 16 + r16 : 0;
            hb[nt] = lds_ld128(act_off + p * 256 + (((((q & 1) << 3) | (ks << 1) | (q >> 1)) ^ (p & 7)) << 4));
        }
#pragma unroll
        for (int m = 0; m < 3; m++) {
            // policy channels (tiles 0, 1): MFMA row i of tile m is channel 8 (i >> 2) + 4 m + (i & 3), so that a lane ends up
            // with 8 consecutive channels of its pixel (one 16-byte store; the four lanes of a pixel write its whole 64-B row)
            const int row = m < 2 ? 8 * (r16 >> 2) + 4 * m + (r16 & 3) : m * 16 + r16;
            ha[m] = lds_ld128(row * 256 + (((ks * 4 + q) ^ ((row & 7) << 1)) << 4));
        }
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int nt = 0; nt < 6; nt++)
                hacc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[m], hb[nt], hacc[m][nt], 0, 0, 0);
    }
    stamp(60);
    uint8_t *Pb = reinterpret_cast<uint8_t *>(A.P) + (size_t)board * PIX * 64;
    uint8_t *Vb = reinterpret_cast<uint8_t *>(A.V) + (size_t)board * PIX * 16;
    {
        // policy head: channels 8 q .. 8 q + 7 of pixel p (tiles 0 and 1), bias, bf16, ReLU, 16 bytes; a store instruction
        // writes 16 pixels x 64 B = 1 KB contiguously
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(A.bh + 8 * q), b1 = *reinterpret_cast<const f32x4 *>(A.bh + 8 * q + 4);
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r16;
            if (board_ok && p < PIX) {
                const uint4 pk = make_uint4(relu_bf16x2(pack_bf16x2(hacc[0][nt][0] + b0[0], hacc[0][nt][1] + b0[1])),
                                            relu_bf16x2(pack_bf16x2(hacc[0][nt][2] + b0[2], hacc[0][nt][3] + b0[3])),
                                            relu_bf16x2(pack_bf16x2(hacc[1][nt][0] + b1[0], hacc[1][nt][1] + b1[1])),
                                            relu_bf16x2(pack_bf16x2(hacc[1][nt][2] + b1[2], hacc[1][nt][3] + b1[3])));
                *reinterpret_cast<uint4 *>(Pb + p * 64 + q * 16) = pk;
            }
        }
        // value head: channels 32 + 4 q .. (q < 2) of tile 2
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(A.bh + 32 + 4 * q);
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r16;
            if (board_ok && p < PIX && q < 2) {
                const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(hacc[2][nt][0] + b4[0], hacc[2][nt][1] + b4[1])),
                                            relu_bf16x2(pack_bf16x2(hacc[2][nt][2] + b4[2], hacc[2][nt][3] + b4[3])));
                *reinterpret_cast<uint2 *>(Vb + p * 16 + q * 8) = pk;
            }
        }
    }
    stamp(61);
    if (clk_sample) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            atomicAdd(A.stamps + 0, c1 - clk0);
            atomicAdd(A.stamps + 1, r1 - rt0);
            atomicAdd(A.stamps + 2, 1ull);
        }
    }
}

}  // namespace

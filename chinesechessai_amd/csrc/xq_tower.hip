// xq_tower.hip — the whole convolutional trunk of the policy/value network in ONE launch.
//
//   planes [G][10][9][16] bf16  ->  conv3x3(16->128)+ReLU  ->  nblocks x { conv+ReLU, conv+skip+ReLU }
//                               ->  1x1 heads (policy 32, value 8)+ReLU  ->  P [G][90*32], V [G][90*8]
//
// Every layer of this network is board-local (zero padding never crosses a board), so a workgroup
// can carry its boards through all layers without any grid-wide synchronisation: the activations
// stay in LDS from the input planes to the head outputs (the residual too: round 1 kept it in registers), and HBM
// sees 2.9 KB in + 7.2 KB out per board instead of 23 KB in/out (+23 KB residual) per layer.
// The per-layer kernels of xq_conv.hip spend 18 % of a workgroup's time in those global
// prologue / epilogue phases (tools/bench_conv.py stamps); here only the weight stream is left.
//
// Builds (xq_tower_set_variant):
//   k_tower1wa (60)      the default from 2,048 boards up (round 4, xq_tower1wa.hpp): ONE wave per SIMD, a wave = a board x all 128
//                        output channels (wave tile 128 x 96, accumulators on a[0:191]), 4 boards per 256-thread workgroup, ring
//                        of 4 weight stages; the residual tower is one generated, symbolically checked asm statement in which a
//                        layer's epilogue runs under the next layer's first tap.  Same bits as k_tower16b
// k_tower16b / k_tower: the tiling of k_conv3x3_b variant B: a wave = (board, output-channel half), wave tile 96 pixels x 64
// channels, weights streamed by LDS-DMA in K-slices [128 cout][64 cin] through a double buffer:
//   k_tower16b<NB = 4> (39)  round 3's default from 2,048 boards up: 4 boards per 512-thread workgroup, one
//                        workgroup per CU, ONE weight stream per 4 boards (half the L2 -> LDS traffic of the 2-board
//                        form); all 8 waves run in lock-step through the stage barriers.  Costs cycles (412 k against
//                        376 k per 4 boards) and wins them back in clock (2.06-2.10 against 1.85-1.90 GHz): 2-3 % faster
//                        on the MI355X boards of round 3, equal on round 2's
//   k_tower16b<NB = 2> (36)  2 boards per 256-thread workgroup, 80.1 KB LDS, 2 workgroups per CU: the default below
//                        2,048 boards (a 4-board workgroup per CU leaves CUs empty there)
//   k_tower (0)          v_mfma_f32_32x32x16_bf16, the first version: the comparison build
// k_tower16b: v_mfma_f32_16x16x32_bf16, fully unrolled issue stream; the output channels are dealt to the MFMA rows so
// that a lane owns 8 consecutive channels of a pixel and the epilogue stores 16 bytes per lane, conflict-free; one
// fragment read / DMA piece per MFMA gap.  Both NB forms accumulate every output element in the same order and agree to
// the bit (k_tower differs in the last bit of some elements).
// Experiments that lost and the timing probes live in xq_tower_probes.hpp (-DXQ_TOWER_PROBES=1 only): k_tower1w (one
// wave per SIMD, 128 x 96 wave tile), the ablation builds, bare MFMA / loop probes.  Deleted after measurement: the
// round-1 issue stream (k_tower16), 8-byte epilogue stores, the skip connection on the VALU, k_tower16s (two groups of 2
// boards two stage steps apart) - DESIGN.md section 5.
// Reference ops: neural_network.py:54-66,181-187 with eval-mode BatchNorm folded.
#include "../../include/xq_selfplay.h"
#include "../../include/xq_debug.h"
#include "xq_mfma.hpp"
#include <atomic>
#include <type_traits>
#include <utility>

#ifndef XQ_TOWER_PROBES
#define XQ_TOWER_PROBES 0      // 1: also compile the ablation / option builds behind xq_tower_debug_stamps (tools/bench_tower.py, tools/probe_tiles.py)
#endif

namespace {
using namespace xqm;

constexpr int PIX = 90, COUT = 128;
constexpr int ACT_BYTES = PIX * 256;          // one board, 128 channels bf16
constexpr int WBUF_BYTES = COUT * 128;        // one weight stage: [128 cout][64 cin] bf16
constexpr int LDS_BYTES = 2 * ACT_BYTES + 2 * WBUF_BYTES + 256 + 2 * 512;
constexpr int LDS_BYTES4 = 4 * ACT_BYTES + 2 * WBUF_BYTES + 256 + 2 * 512;      // k_tower16b<.., NB = 4>

struct TowerArgs {
    const uint16_t *planes;    // [G][90][16]
    const uint16_t *w1;        // [9][128][16]
    const uint16_t *wt;        // [2*nblocks][9][128][128]
    const float *bias;         // [1 + 2*nblocks][128]
    const uint16_t *wh;        // [64][128]  (rows 0..31 policy, 32..39 value, rest zero)
    const float *bh;           // [64]
    uint16_t *P;               // [G][90][32]
    uint16_t *V;               // [G][90][8]
    int G, nblocks;
    unsigned long long *stamps;   // diagnostic builds only: 64 u64 per workgroup
    // evaluator row compaction (xq_engine_set_row_compaction), both optional: board b reads the planes of row
    // row_src[b] and only boards below *n_rows exist (device values of the preceding k_assign_rows)
    const int32_t *row_src;
    const int32_t *n_rows;
};

template <bool STAMP>
__global__ __launch_bounds__(256, 2) void k_tower(TowerArgs A)
{
    // LDS: [weight stages 2 x 16 KB][activations 2 boards x 23,040 B][zero row 256 B][bias 2 x 512 B]
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int ACT0 = 2 * WBUF_BYTES, ZROW = ACT0 + 2 * ACT_BYTES, BIAS = ZROW + 256;
    uint8_t *wbuf = lds;
    float *lbias = reinterpret_cast<float *>(lds + BIAS);            // [2][128]

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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb_ = wave >> 1, hc = wave & 1;                         // board in workgroup, channel half
    int nrows = A.G;
    if (A.n_rows) { const int n = *A.n_rows; nrows = n < A.G ? n : A.G; }
    if ((int)blockIdx.x * 2 >= nrows) return;       // compaction: no row for this workgroup
    const int board = blockIdx.x * 2 + wb_;
    const bool board_ok = board < nrows;
    const int act_off = ACT0 + wb_ * ACT_BYTES;
    const int r32 = lane & 31, h = lane >> 5;

    int opix[3];
    uint32_t vmask[3];                    // bit t: tap t of this output pixel reads a real pixel
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
        const int o = nt * 32 + r32;
        opix[nt] = o;
        uint32_t m = 0;
        if (o < PIX) {
            const int yy = o / 9, xx = o % 9;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                if (yy + dy >= 0 && yy + dy < 10 && xx + dx >= 0 && xx + dx < 9) m |= 1u << t;
            }
        }
        vmask[nt] = m;
    }
    // staging address of (pixel, 16-B chunk c of the 256-B row): sbase ^ (c_local << 4)
    int sbase[3];
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
        const int p = opix[nt] < PIX ? opix[nt] : 0;
        sbase[nt] = (act_off + p * 256 + 8 * h + ((p & 15) << 4)) ^ (hc << 7);
    }

    f32x16 acc[2][3];
    uint2 xres[2][4][3];                  // block input (residual), packed bf16 in accumulator layout
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[mt][nt][i] = 0.f;
    };
    auto mfma6 = [&](const bf16x8 (&af)[2], const bf16x8 (&bf)[3]) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    };

    // ---------------------------------------------------------------- input conv (16 -> 128)
    // planes rows are 32 B (2 chunks); swizzle (row / 8) & 1; conv1 tap slices [128][16] = 4 KB
    if (tid < 16) reinterpret_cast<uint4 *>(lds + ZROW)[tid] = make_uint4(0, 0, 0, 0);
    if (wave == 1 && lane < 32) dma16(A.bias + lane * 4, lbias);
    auto stage_w1 = [&](int tap, int buf) {           // 256 chunks: one piece per wave
        const int q0 = wave * 64, q = q0 + lane, row = q >> 1, cp = q & 1;
        dma16(reinterpret_cast<const uint8_t *>(A.w1) + (size_t)tap * COUT * 32 + row * 32 + ((cp ^ ((row >> 3) & 1)) * 16),
              wbuf + buf * 4096 + q0 * 16);
    };
    stage_w1(0, 0);
    if (board_ok) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.planes) + (size_t)(A.row_src ? A.row_src[board] : board) * PIX * 32;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int q0 = (j * 2 + hc) * 64, q = q0 + lane, p = q >> 1, cp = q & 1;
            if (q < PIX * 2) dma16(src + p * 32 + ((cp ^ ((p >> 3) & 1)) * 16), lds + act_off + q0 * 16);
        }
    }
    zero_acc();
    barrier_dma();
    for (int tap = 0; tap < 9; tap++) {
        const int buf = tap & 1;
        if (tap + 1 < 9) stage_w1(tap + 1, buf ^ 1);
        const int off = (tap / 3 - 1) * 9 + (tap % 3 - 1);
        bf16x8 bf[3], af[2];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const bool ok = (vmask[nt] >> tap) & 1u;
            const int sp = opix[nt] + off;
            // (padding lanes read the zero region at the slot their own row would have used: no bank conflicts)
            const int a = (ok ? act_off + sp * 32 : ZROW + (sp & 7) * 32) + ((((sp >> 3) & 1) ^ h) << 4);
            bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + a);
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int row = hc * 64 + mt * 32 + r32;
            af[mt] = *reinterpret_cast<const bf16x8 *>(wbuf + buf * 4096 + row * 32 + ((h ^ ((row >> 3) & 1)) * 16));
        }
        mfma6(af, bf);
        barrier_dma();
    }
    stamp(1);

    // weight stream of the 128-channel layers, one global sequence of stages across all layers:
    // stage g = (layer, tap, K-slice) = [128 cout][64 cin] = 16 pieces of 1 KB, 4 per wave, into
    // buffer g & 1 (18 stages per layer: the parity is the K-slice)
    const int nlayers = 2 * A.nblocks, nstages = nlayers * 18;
    const rsrc_t wrsrc = make_rsrc(A.wt, nlayers * 9 * COUT * COUT * 2);
    int wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = (wave * 4 + j) * 64 + lane, row = q >> 3, cp = q & 7;
        wsrc[j] = row * 256 + ((cp ^ ((row >> 1) & 7)) * 16);
    }
    auto stage_weights = [&](int g, int buf) {
        const int soff = (g >> 1) * (COUT * COUT * 2) + (g & 1) * 128;
#pragma unroll
        for (int j = 0; j < 4; j++) dma16_buf(wrsrc, wsrc[j], soff, wbuf + buf * WBUF_BYTES + (wave * 4 + j) * 1024);
    };
    // A fragment of (mt, K-step kk) in a stage buffer: abase ^ (kk << 5), + mt * 4096
    const int abase = (hc * 64 + r32) * 128 + ((h ^ ((r32 >> 1) & 7)) << 4);
    auto load_a = [&](bf16x8 (&af)[2], int sl, int kk) {
        const int a = abase ^ (kk << 5);
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
            af[mt] = *reinterpret_cast<const bf16x8 *>(lds + a + sl * WBUF_BYTES + mt * 4096);
    };
    auto load_b = [&](bf16x8 (&bf)[3], const int (&a0)[3], int cconst) {
#pragma unroll
        for (int nt = 0; nt < 3; nt++) bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + (a0[nt] ^ cconst));
    };
    auto tap_addr = [&](int (&a0)[3], int tap) {
        const int off = (tap / 3 - 1) * 9 + (tap % 3 - 1);
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const bool ok = (vmask[nt] >> tap) & 1u;
            const int sp = opix[nt] + off;
            a0[nt] = (ok ? act_off + sp * 256 : ZROW) + (((sp & 15) ^ h) << 4);
        }
    };

    // epilogue: acc + bias [+ residual] -> ReLU -> bf16 -> LDS rows in place (+ keep as next residual).
    // The store addresses are derived here from one base per tile (opaque to the optimiser: hoisted
    // out of the layer loop they would occupy 24 registers the fragment double buffer needs).
    auto epilogue = [&](const float *lb, auto add_res, auto keep_res) {
        int sb[3];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            sb[nt] = sbase[nt];
            asm volatile("" : "+v"(sb[nt]));
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = hc * 64 + mt * 32 + 8 * q + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(lb + c0);
#pragma unroll
                for (int nt = 0; nt < 3; nt++) {
                    float v0 = acc[mt][nt][4 * q + 0] + b4[0], v1 = acc[mt][nt][4 * q + 1] + b4[1];
                    float v2 = acc[mt][nt][4 * q + 2] + b4[2], v3 = acc[mt][nt][4 * q + 3] + b4[3];
                    if constexpr (decltype(add_res)::value) {
                        const uint2 r = xres[mt][q][nt];
                        v0 += bf16_lo(r.x); v1 += bf16_hi(r.x); v2 += bf16_lo(r.y); v3 += bf16_hi(r.y);
                    }
                    const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(v0, v1)), relu_bf16x2(pack_bf16x2(v2, v3)));
                    if constexpr (decltype(keep_res)::value) xres[mt][q][nt] = pk;
                    if (nt < 2 || opix[nt] < PIX)
                        *reinterpret_cast<uint2 *>(lds + (sb[nt] ^ ((mt * 4 + q) << 4))) = pk;
                }
            }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    // conv1 epilogue (all taps were read before the last barrier: rows are rewritten in place);
    // the first two weight stages and the first tower bias land behind it
    if (nstages > 0) {
        stage_weights(0, 0);
        stage_weights(1, 1);
        if (wave == 1 && lane < 32) dma16(A.bias + 128 + lane * 4, lbias + 128);
    }
    epilogue(lbias, no{}, yes{});
    barrier_dma();
    stamp(2);

    // ---------------------------------------------------------------- residual tower
    for (int layer = 0; layer < nlayers; layer++) {
        zero_acc();
        if (wave == 1 && lane < 32 && layer + 1 < nlayers)          // next layer's bias, slot (layer + 2) & 1
            dma16(A.bias + (size_t)(layer + 2) * 128 + lane * 4, lbias + (layer & 1) * 128);
        int a0[3];
        tap_addr(a0, 0);
        bf16x8 fa[2][2], fb[2][3];                                  // fragment double buffer (K-step parity)
        load_a(fa[0], 0, 0);
        load_b(fb[0], a0, 0);
        for (int tap = 0; tap < 9; tap++) {
            int a0n[3];
            tap_addr(a0n, tap < 8 ? tap + 1 : 8);
#pragma unroll
            for (int u = 0; u < 8; u++) {                           // 2 K-slices x 4 K-steps of 16 channels
                const int sl = u >> 2, kk = u & 3, cur = u & 1;
                if (kk < 3) {
                    load_a(fa[cur ^ 1], sl, kk + 1);
                    load_b(fb[cur ^ 1], a0, (sl * 8 + (kk + 1) * 2) << 4);
                    mfma6(fa[cur], fb[cur]);
                    // issue order: one fragment read of the next K-step behind each MFMA of this one
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                } else {
                    // end of a stage: every wave has its last fragments of this buffer in registers,
                    // the next stage's pieces have landed -> fetch the next stage's first fragments and
                    // refill this buffer two stages ahead while this K-step's MFMAs run
                    barrier_dma();
                    const int g = layer * 18 + tap * 2 + sl;
                    if (sl == 0) {
                        load_a(fa[cur ^ 1], 1, 0);
                        load_b(fb[cur ^ 1], a0, 8 << 4);
                    } else {                                          // (tap 8: a dead prefetch of tap 8 again)
                        load_a(fa[cur ^ 1], 0, 0);
                        load_b(fb[cur ^ 1], a0n, 0);
                    }
                    // (the last two stages of the tower refetch the final stage into the dead buffer
                    // rather than branch: a branch here would split the MFMA block)
                    stage_weights(g + 2 < nstages ? g + 2 : nstages - 1, sl);
                    mfma6(fa[cur], fb[cur]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int j = 0; j < 4; j++) {                     // one LDS-DMA piece per MFMA gap
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < 3; nt++) a0[nt] = a0n[nt];
        }
        if (layer < 28) stamp(3 + 2 * layer);
        // bias slot of tower layer L is (L + 1) & 1 (slot 0 held conv1's); odd layers close a block
        if (layer & 1) epilogue(lbias + ((layer + 1) & 1) * 128, yes{}, yes{});
        else epilogue(lbias + ((layer + 1) & 1) * 128, no{}, no{});
        barrier_dma();
        if (layer < 28) stamp(4 + 2 * layer);
    }

    // ---------------------------------------------------------------- heads (1x1, 128 -> 32 + 8)
    {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.wh);     // [64][256 B], swizzle row & 15
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q0 = (wave * 4 + j) * 64, q = q0 + lane, row = q >> 4, cp = q & 15;
            dma16(src + row * 256 + ((cp ^ (row & 15)) * 16), wbuf + q0 * 16);
        }
    }
    f32x16 hacc[3];
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int i = 0; i < 16; i++) hacc[nt][i] = 0.f;
    barrier_dma();
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        const int c = kk * 2 + h;
        const int row = hc * 32 + r32;                             // wave hc: policy rows / value rows
        const bf16x8 af = *reinterpret_cast<const bf16x8 *>(wbuf + row * 256 + ((c ^ (row & 15)) * 16));
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const int p = opix[nt] < PIX ? opix[nt] : 0;
            const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(lds + act_off + p * 256 + ((c ^ (p & 15)) * 16));
            hacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, hacc[nt], 0, 0, 0);
        }
    }
    stamp(60);
    if (!board_ok) { stamp(61); return; }
    uint8_t *Pb = reinterpret_cast<uint8_t *>(A.P) + (size_t)board * PIX * 64;
    uint8_t *Vb = reinterpret_cast<uint8_t *>(A.V) + (size_t)board * PIX * 16;
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
        const int p = opix[nt];
        if (p < PIX) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (hc == 1 && q > 0) break;                       // value head: channels 32..39 only
                const int c0 = hc * 32 + 8 * q + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(A.bh + c0);
                const float v0 = hacc[nt][4 * q + 0] + b4[0], v1 = hacc[nt][4 * q + 1] + b4[1];
                const float v2 = hacc[nt][4 * q + 2] + b4[2], v3 = hacc[nt][4 * q + 3] + b4[3];
                const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(v0, v1)), relu_bf16x2(pack_bf16x2(v2, v3)));
                if (hc == 0) *reinterpret_cast<uint2 *>(Pb + p * 64 + c0 * 2) = pk;
                else *reinterpret_cast<uint2 *>(Vb + p * 16 + 4 * h * 2) = pk;
            }
        }
    }
    stamp(61);
}


// ------------------------------------------------------------------------------------------
// k_tower16b — the 16x16x32 trunk with its issue stream rebuilt around what k_tower16's ISA showed
// (round 2; k_tower16 stays selectable for A/B runs).  Same workgroup, tile, LDS image, weight
// stream and numerics contract; what changed, each aimed at a wave's speed when it has the SIMD
// to itself (k_tower16: 21.0 k cycles per layer against 13.8 k of MFMA issue):
//   * the tap loop of a layer is fully unrolled (36 K-steps), so every `s_waitcnt lgkmcnt` is an
//     exact count: the rolled loop drained the LDS queue at its back edge and again mid-K-step,
//     right behind reads it had just issued;
//   * activation fragments are double-buffered by K-step (all 6 of K-step k+1 are fetched during
//     K-step k: 24 MFMAs of distance instead of 8).  The 48 registers come from the skip
//     connection, which no longer lives in registers: the first convolution of a block re-reads
//     the block input x from the LDS rows its epilogue is about to overwrite and starts the second
//     convolution's accumulators at x + bias — through the matrix pipe: x + bias = S . X + bias with
//     a 0/1 selector as the A operand, X as 12 ordinary fragments, bias as the C operand (exact).  Every layer's bias
//     enters through the accumulator initialisation, so an epilogue is cvt / ReLU / store only;
//   * stage barriers are raw s_barrier + vmcnt(0) (no LDS drain), placed behind the first pixel
//     tile of a stage's second K-step: its 4 MFMAs consume all 4 weight fragments of the retiring
//     buffer, so each wave's reads of that buffer are complete when it arrives.  The last stage
//     barrier of a layer also waits lgkmcnt(0): every activation read of the layer has been
//     issued before it (none follow), so no wave can start its in-place epilogue while another
//     still reads the layer input — the window k_tower16 left open;
//   * tap addresses: row validity is kept as wave-uniform lane masks (SGPR pairs), the pixel-tile
//     offset rides in the ds_read immediate and the swizzle term is shared by the 6 tiles:
//     ~13 VALU per tap instead of ~54.
// ------------------------------------------------------------------------------------------
template <bool STAMP, int ABL = 0, int NB = 2>
__global__ __launch_bounds__(NB * 128, 2) void k_tower16b(TowerArgs A)
{
    constexpr int NWV = NB * 2, PPW = 16 / NWV;                       // waves per workgroup, weight pieces per wave and stage
    constexpr int ACT0 = 2 * WBUF_BYTES, ZROW = ACT0 + NB * ACT_BYTES, BIAS = ZROW + 256;   // bias: [2][128] f32

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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb_ = wave >> 1, hc = wave & 1;                         // board in workgroup, channel half
    // The output channels are dealt to the MFMA rows so that a lane's results of the weight tiles 2j and 2j + 1
    // are 8 consecutive channels = one 16-byte chunk of an activation row (row i of tile mt -> channel
    // hc * 64 + (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3)): the epilogue stores ds_write_b128 and the
    // activation rows are swizzled for it (chunk of K-step ks, lane quarter q at ((q & 1) << 3 | ks << 1 | q >> 1) ^
    // (pixel & 7)): conflict-free stores (8 consecutive pixels -> 8 chunks of one 128-byte half) and fragment reads.
    // (The first build stored ds_write_b64 at chunk ^ ((pixel & 7) << 1): 4-way conflicts on the 32 store banks, all of
    // the kernel's LDS conflicts.)
    auto chan_row = [&](int mt, int i) { return hc * 64 + (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3); };
    int nrows = A.G;
    if (A.n_rows) { const int n = *A.n_rows; nrows = n < A.G ? n : A.G; }
    if ((int)blockIdx.x * NB >= nrows) return;     // compaction: no row for this workgroup (uniform, before any barrier / DMA)
    const int board = blockIdx.x * NB + wb_;
    const bool board_ok = board < nrows;
    const int act_off = ACT0 + wb_ * ACT_BYTES;
    const int r16 = lane & 15, q = lane >> 4;

    f32x4 acc[4][6];

    // ---------------------------------------------------------------- input conv (16 -> 128)
    // as in k_tower16 (all 9 tap slices staged at once, planes in the tail of the activation region);
    // the accumulators start at this layer's bias
    const int pl_off = act_off + ACT_BYTES - PIX * 32;
    if (tid < 16) lds_st128(ZROW + tid * 16, make_uint4(0, 0, 0, 0));
    const int nlayers = 2 * A.nblocks, nstages = nlayers * 18;
    if (wave == 1 && lane < 32 && nstages > 0) dma16_abs(A.bias + 128 + lane * 4, BIAS + 512);   // bias[1] -> slot 1
#pragma unroll
    for (int j = 0; j < (36 + NWV - 1) / NWV; j++) {
        const int piece = j * NWV + wave;
        if (piece < 36) dma16_abs(reinterpret_cast<const uint8_t *>(A.w1) + piece * 1024 + lane * 16, piece * 1024);
    }
    if (board_ok) {
        const int srow = A.row_src ? A.row_src[board] : board;      // (wave-uniform: a scalar load)
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.planes) + (size_t)srow * PIX * 32;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int q0 = (j * 2 + hc) * 64, idx = q0 + lane;
            if (idx < PIX * 2) dma16_abs(src + idx * 16, pl_off + q0 * 16);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(A.bias + chan_row(mt, 4 * q));
#pragma unroll
        for (int nt = 0; nt < 6; nt++) acc[mt][nt] = b4;
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
            bf16x8 bf[6], af[4];
#pragma unroll
            for (int nt = 0; nt < 6; nt++) {
                const bool ok = tap_real && ((vm[nt / 3] >> ((nt % 3) * 9 + tp)) & 1u);
                const int sp = nt * 16 + r16 + off;
                bf[nt] = lds_ld128((ok ? pl_off + sp * 32 : ZROW + (sp & 7) * 32) + (q & 1) * 16);
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
                af[mt] = lds_ld128((tp * COUT + chan_row(mt, r16)) * 32 + (q & 1) * 16);
#pragma unroll
            for (int nt = 0; nt < 6; nt++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    barrier_dma();                                                 // every wave is done with planes and tap slices
    stamp(1);

    // weight stream of the 128-channel layers: as in k_tower16 (stage g -> buffer g & 1)
    const rsrc_t wrsrc = make_rsrc(A.wt, nlayers * 9 * COUT * COUT * 2);
    // LDS row rho = piece * 8 + (lane >> 3) of a stage image holds the weights of channel chan_row(rho):
    // piece P = channel half P >> 3, weight tile (P >> 1) & 3, tile rows (P & 1) * 8 .. -> a scalar row offset per piece
    // plus a lane part
    const int wch = (lane & 7) ^ (lane >> 4);
    const int wsrc_even = ((lane >> 5) * 8 + ((lane >> 3) & 3)) * 256 + (wch << 4), wsrc_odd = wsrc_even ^ 64;
    auto piece_off = [&](int P) { return ((P >> 3) * 64 + ((P >> 2) & 1) * 32 + (P & 1) * 16 + ((P >> 1) & 1) * 4) * 256; };
    auto stage_piece = [&](int g, int buf, int j) {                   // j < PPW
        const int soff = (g >> 1) * (COUT * COUT * 2) + (g & 1) * 128 + piece_off(wave * PPW + j);
        dma16_buf_abs(wrsrc, (j & 1) ? wsrc_odd : wsrc_even, soff, buf * WBUF_BYTES + (wave * PPW + j) * 1024);
    };
    // A fragment (weight tile mt, K-step kk of a stage): abase ^ (kk << 6), + mt * 2048
    const int abase = (hc * 64 + r16) * 128 + ((q ^ ((r16 >> 1) & 7)) << 4);
    auto load_a1 = [&](bf16x8 &af, int mt, int sl, int kk) {
        af = lds_ld128((abase ^ (kk << 6)) + sl * WBUF_BYTES + mt * 2048);
    };

    // row validity as lane masks (wave-uniform, SGPR pairs): pixel tile nt, lane -> pixel nt * 16 + r16.
    // Only tile 0 holds board row 0, only tile 5 holds board row 9 and the 6 slots past pixel 89.
    bool xl[6], xr[6];
#pragma unroll
    for (int nt = 0; nt < 6; nt++) {
        const int p = nt * 16 + r16, xx = p % 9;
        xl[nt] = xx != 0 && p < PIX;
        xr[nt] = xx != 8 && p < PIX;
    }
    const bool real5 = r16 < PIX - 80, yu0 = r16 >= 9, yd5 = r16 == 0;
    // row address of (pixel tile nt, tap), before the K-step term and without nt * 4096 (ds_read immediate):
    //   real pixel : act_off + sp * 256 + slot,  sp = r16 + tap offset (tile-independent: 16 | nt * 16)
    //   padding    : the zero row at the slot the lane's own row would have used (conflict-free groups)
    const int Rrow = act_off + r16 * 256, r5 = r16 << 4, q4 = (((q & 1) << 3) | (q >> 1)) << 4;
    auto tap_addrs = [&](int (&as)[6], int tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, off = dy * 9 + dx;
        // The addresses do not depend on the layer: left alone, the compiler computes all 54 ahead of the
        // layer loop and spills them (each reload then waits vmcnt(0) in the middle of the MFMA stream, a
        // full memory round trip that also drains the weight DMA).  An opaque copy of the row base per
        // call keeps the ~13 VALU instructions of a tap where they are written.
        int rrow = Rrow, r5o = r5;
        asm volatile("" : "+v"(rrow), "+v"(r5o));
        const int slot = ((r5o + off * 16) & 0x70) ^ q4;              // (((q & 1) << 3 | q >> 1) ^ (sp & 7)) << 4
        const int aok = rrow + off * 256 + slot;
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const bool sel = dx != 0 || (nt == 0 && dy < 0) || nt == 5;
            if (!sel) { as[nt] = aok; continue; }
            bool ok = dx < 0 ? xl[nt] : dx > 0 ? xr[nt] : real5;    // (dx == 0 needs a select for tile 5 and for tile 0 / row 0)
            if (dx == 0 && nt == 0) ok = yu0;
            else if (nt == 0 && dy < 0) ok = ok && yu0;
            if (nt == 5 && dy > 0) ok = ok && yd5;
            as[nt] = ok ? aok : slot + (ZROW - nt * 4096);
        }
    };
    auto load_b1 = [&](bf16x8 &bf, int a, int nt, int ks) { bf = lds_ld128((a ^ (ks << 5)) + nt * 4096); };

    // epilogue of one layer: acc (bias [+ skip] already inside) -> bf16 -> ReLU -> LDS rows in place;
    // READ_X (first convolution of a block): the block input x is read back from those rows first and
    // the next layer's accumulators start at x + bias; otherwise they start at the next layer's bias
    auto epilogue = [&](auto read_x, int lb_next) {
        int ln;                                                       // lane id, 2 VALU, not CSE-able (see k_tower16)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int r = ln & 15, qq = ln >> 4;
        int sb[6];                                                    // this lane's chunk of K-step hc * 2 (+ j: ^ (j << 5)) of pixel nt * 16 + r
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r < PIX ? nt * 16 + r : 0;
            sb[nt] = act_off + p * 256 + ((((((qq & 1) << 3) | (qq >> 1)) ^ (hc << 2)) ^ (p & 7)) << 4);
        }
        const int lbq = lb_next + (hc * 64 + qq * 8) * 4;              // bias of channel hc * 64 + j * 32 + qq * 8 + t * 4 ..
        // READ_X through the matrix pipe: x + bias = S . X + bias with a 0/1 selector S as the MFMA's A operand (row i of
        // tile 2j + t picks input channel (i >> 2) * 8 + t * 4 + (i & 3) of K-step hc * 2 + j), X = this wave's own 64
        // channels of the block input as 12 ordinary B fragments, bias as the C operand: one fp32 addition per element
        // (the MFMA adder's rounding), 24 MFMAs + 12 ds_read_b128 instead of 144 VALU + 24 ds_read_b64 - an epilogue runs
        // beside the partner wave's MFMA stream, where VALU issue slots are what is scarce (the VALU form: +1.3 % wall)
        bf16x8 xf[2][6], sel[2];
        if constexpr (decltype(read_x)::value) {
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                for (int nt = 0; nt < 6; nt++) xf[k2][nt] = lds_ld128(sb[nt] ^ (k2 << 5));     // the chunk this lane overwrites below
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    sel[o][j] = (qq == (r >> 2) && j == o * 4 + (r & 3)) ? (__bf16)1.0f : (__bf16)0.0f;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const f32x4 bn0 = lds_ldf4(lbq + j * 128), bn1 = lds_ldf4(lbq + j * 128 + 16);
#pragma unroll
            for (int nt = 0; nt < 6; nt++) {
                const f32x4 v0 = acc[2 * j][nt], v1 = acc[2 * j + 1][nt];
                const uint4 pk = make_uint4(relu_bf16x2(pack_bf16x2(v0[0], v0[1])), relu_bf16x2(pack_bf16x2(v0[2], v0[3])),
                                            relu_bf16x2(pack_bf16x2(v1[0], v1[1])), relu_bf16x2(pack_bf16x2(v1[2], v1[3])));
                if constexpr (decltype(read_x)::value) {
                    acc[2 * j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[0], xf[j][nt], bn0, 0, 0, 0);
                    acc[2 * j + 1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[1], xf[j][nt], bn1, 0, 0, 0);
                } else {
                    acc[2 * j][nt] = bn0;
                    acc[2 * j + 1][nt] = bn1;
                }
                if (nt < 5 || r < PIX - 80) lds_st128(sb[nt] ^ (j << 5), pk);
            }
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    if (nstages > 0) {
#pragma unroll
        for (int j = 0; j < PPW; j++) { stage_piece(0, 0, j); stage_piece(1, 1, j); }
    }
    epilogue(no{}, BIAS + 512);                                     // conv1 output; tower layer 0 starts at bias[1]
    barrier_dma();
    stamp(2);

    // ---------------------------------------------------------------- residual tower
    bf16x8 fa[2][4], fb[2][6];                                       // fragments, double-buffered by K-step parity
    if (nstages > 0) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) load_a1(fa[0][mt], mt, 0, 0);
    }
    for (int layer = 0; layer < nlayers; layer++) {
        if (wave == 1 && lane < 32 && layer + 1 < nlayers)          // bias of tower layer L + 1 (row L + 2) -> slot L & 1
            dma16_abs(A.bias + (size_t)(layer + 2) * 128 + lane * 4, BIAS + (layer & 1) * 512);
        int as[6];
        tap_addrs(as, 0);
#pragma unroll
        for (int nt = 0; nt < 6; nt++) load_b1(fb[0][nt], as[nt], nt, 0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            int asn[6];
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {                        // 4 K-steps of 32 channels = 2 weight stages
                const int sl = ks >> 1, kk = ks & 1, cur = ks & 1;
                const int g = layer * 18 + tap * 2 + sl;
                const bool last = tap == 8 && ks == 3;                // last K-step of the layer: no activation prefetch
                if (ks == 3 && !last) tap_addrs(asn, tap + 1);
#pragma unroll
                for (int n = 0; n < 6; n++) {                        // pixel tile n: 4 MFMAs
                    // the next K-step's activation fragment of this tile
                    if (!last) {
                        if (ks < 3) load_b1(fb[cur ^ 1][n], as[n], n, ks + 1);
                        else load_b1(fb[cur ^ 1][n], asn[n], n, 0);
                    }
                    // the next K-step's weight fragments (kk == 1: from the stage the barrier just published)
                    // and the refill of the retiring buffer two stages ahead
                    if (kk == 0 && n < 4) load_a1(fa[cur ^ 1][n], n, sl, 1);
                    if (kk == 1 && n >= 1 && n < 5) {
                        load_a1(fa[cur ^ 1][n - 1], n - 1, sl ^ 1, 0);
                        if (!(ABL & 1) && n - 1 < PPW) stage_piece(g + 2 < nstages ? g + 2 : nstages - 1, sl, n - 1);
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; mt++)
                        acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][mt], fb[cur][n], acc[mt][n], 0, 0, 0);
                    // issue order inside the tile: reads, first MFMA, DMA piece, the other MFMAs
                    const int nrd = (last ? 0 : 1) + ((kk == 0 && n < 4) || (kk == 1 && n >= 1 && n < 5) ? 1 : 0);
                    // one filler per MFMA gap: a 16-cycle MFMA leaves room for ~2 issue slots beside it, so two reads,
                    // the m0 write and a DMA piece in ONE gap stall the matrix pipe (clustered in front of the tile's
                    // MFMAs: +0.1..0.3 % at 16,384 boards, +4..5 % at 512)
                    if (nrd >= 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (kk == 1 && n >= 1 && n - 1 < PPW && !(ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (kk == 1 && n == 0 && !(ABL & 2)) {
                        // stage barrier: the next stage has landed (every wave drains its own pieces first) and,
                        // this tile's MFMAs having consumed all 4 weight fragments of buffer sl, nobody reads it any more
                        // (sched_barrier: the tile's MFMAs, and with them the waits for their operands, stay above)
                        __builtin_amdgcn_sched_barrier(0);
                        if (last) __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): + all activation reads done
                        else __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (tap < 8) {
#pragma unroll
                for (int nt = 0; nt < 6; nt++) as[nt] = asn[nt];
            }
        }
        if (layer < 28) stamp(3 + 2 * layer);
        if (layer & 1) epilogue(no{}, BIAS + (layer & 1) * 512);
        else epilogue(yes{}, BIAS + (layer & 1) * 512);
        barrier_dma();
        if (layer < 28) stamp(4 + 2 * layer);
    }

    // ---------------------------------------------------------------- heads (1x1, 128 -> 32 + 8): as in k_tower16
    {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.wh);     // [64][256 B], chunk ^ ((row & 7) << 1)
#pragma unroll
        for (int j = 0; j < PPW; j++) {
            const int q0 = (wave * PPW + j) * 64, idx = q0 + lane, row = idx >> 4, cp = idx & 15;
            dma16_abs(src + row * 256 + ((cp ^ ((row & 7) << 1)) * 16), q0 * 16);
        }
    }
    f32x4 hacc[2][6];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int nt = 0; nt < 6; nt++)
#pragma unroll
            for (int i = 0; i < 4; i++) hacc[m][nt][i] = 0.f;
    barrier_dma();
    const int nm = hc == 0 ? 2 : 1;                                  // policy: rows 0..31, value: rows 32..47
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        bf16x8 hb[6], ha[2];
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r16 < PIX ? nt * 16 + r16 : 0;
            hb[nt] = lds_ld128(act_off + p * 256 + (((((q & 1) << 3) | (ks << 1) | (q >> 1)) ^ (p & 7)) << 4));
        }
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int row = hc * 32 + m * 16 + r16;
            ha[m] = lds_ld128(row * 256 + (((ks * 4 + q) ^ ((row & 7) << 1)) << 4));
        }
#pragma unroll
        for (int m = 0; m < 2; m++)
            if (m < nm)
#pragma unroll
                for (int nt = 0; nt < 6; nt++)
                    hacc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[m], hb[nt], hacc[m][nt], 0, 0, 0);
    }
    stamp(60);
    if (!board_ok) { stamp(61); return; }
    uint8_t *Pb = reinterpret_cast<uint8_t *>(A.P) + (size_t)board * PIX * 64;
    uint8_t *Vb = reinterpret_cast<uint8_t *>(A.V) + (size_t)board * PIX * 16;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        if (m >= nm) break;
        const int c0 = hc * 32 + m * 16 + 4 * q;                     // head channel of element 0
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(A.bh + c0);
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r16;
            if (p < PIX && (hc == 0 || q < 2)) {                     // value head: channels 32..39 only
                const float v0 = hacc[m][nt][0] + b4[0], v1 = hacc[m][nt][1] + b4[1];
                const float v2 = hacc[m][nt][2] + b4[2], v3 = hacc[m][nt][3] + b4[3];
                const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(v0, v1)), relu_bf16x2(pack_bf16x2(v2, v3)));
                if (hc == 0) *reinterpret_cast<uint2 *>(Pb + p * 64 + c0 * 2) = pk;
                else *reinterpret_cast<uint2 *>(Vb + p * 16 + (c0 - 32) * 2) = pk;
            }
        }
    }
    stamp(61);
}

}  // namespace

static int g_tower_variant = -1;    // -1 = automatic: k_tower1wa from 2,048 boards up, k_tower16b<NB = 2> below;
                                    // 60 = k_tower1wa; 36 = k_tower16b, 2 boards per workgroup; 39 = 4 boards per workgroup; 0 = k_tower (32x32x16
                                    // comparison build); 30, 31, 41, 43, 50, 61..66 = timing probes and experiments (XQ_TOWER_PROBES builds)
// diagnostic switch (include/xq_debug.h): the builds compute the same function (36, 39 and 50 to the bit)
extern "C" void xq_tower_set_variant(int v) { g_tower_variant = v; }

// diagnostic (include/xq_debug.h): three device uint64 that k_tower1wa's product build adds (shader cycles, 100 MHz ticks,
// samples) to, one workgroup in 64; NULL switches it off.  The buffer belongs to the device that is current when it is
// set: launches on any other device do not get the pointer (a process may drive several GPUs, xq_config.device).
static std::atomic<void *> g_clock_sample{ nullptr };
static std::atomic<int> g_clock_sample_dev{ -1 };
extern "C" void xq_tower_set_clock_sample(void *dev_u64x3)
{
    int dev = -1;
    if (dev_u64x3 && hipGetDevice(&dev) != hipSuccess) dev = -1;
    g_clock_sample.store(nullptr, std::memory_order_release);
    g_clock_sample_dev.store(dev, std::memory_order_release);
    g_clock_sample.store(dev >= 0 ? dev_u64x3 : nullptr, std::memory_order_release);
}
static void *clock_sample_for_this_device()
{
    void *p = g_clock_sample.load(std::memory_order_acquire);
    if (!p) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_clock_sample_dev.load(std::memory_order_acquire)) return nullptr;
    return p;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remembered per (kernel, device
// ordinal) - a process may drive several GPUs (xq_config.device) - in one atomic bit mask per kernel
template <auto KERNEL>
static int tower_lds_opt_in(int bytes)
{
    static std::atomic<uint64_t> done{ 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return XQ_E_HIP;
    if (done.load(std::memory_order_acquire) >> dev & 1) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return XQ_E_HIP;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return 0;
}

#include "xq_tower1wa.hpp"
#if XQ_TOWER_PROBES
#include "xq_tower_probes.hpp"
#endif

template <bool STAMP>
static int launch_tower(void *stream, const void *planes, const void *w1, const void *wt, const void *bias, const void *wh,
                        const void *bh, void *policy_out, void *value_out, int n_boards, int n_blocks, void *stamps,
                        const void *row_src, const void *n_rows)
{
    if (!planes || !w1 || !bias || !wh || !bh || !policy_out || !value_out || n_boards <= 0 || n_blocks < 0 ||
        (n_blocks > 0 && !wt) || n_blocks > 64)
        return XQ_E_INVALID;
    TowerArgs a{ (const uint16_t *)planes, (const uint16_t *)w1, (const uint16_t *)wt, (const float *)bias,
                 (const uint16_t *)wh, (const float *)bh, (uint16_t *)policy_out, (uint16_t *)value_out, n_boards, n_blocks,
                 (unsigned long long *)(STAMP ? stamps : clock_sample_for_this_device()), (const int32_t *)row_src, (const int32_t *)n_rows };
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid2((n_boards + 1) / 2), grid4((n_boards + 3) / 4);
#define XQ_TOWER_LAUNCH(KERNEL, GRID, THREADS, LDS)                                                     \
    do {                                                                                                \
        if (int rc = tower_lds_opt_in<&KERNEL>(LDS)) return rc;                                         \
        hipLaunchKernelGGL((KERNEL), GRID, dim3(THREADS), LDS, st, a);                                  \
    } while (0)
    // 4 boards per workgroup share one weight stream (half the L2 -> LDS traffic per board): the chip answers with a
    // higher clock, worth 2-3 % from 2,048 boards up; below that one 512-thread workgroup per CU leaves CUs empty
    // (512 boards: 0.19 against 0.13 ms).  Same bits either way.  (n_boards is the launch's capacity: with a row map
    // the rows really present may be fewer, the choice is then not optimal, never wrong.)
    // Round 4: from 2,048 boards up the default is k_tower1wa (one wave per SIMD, the residual tower as one hand-written asm
    // statement): the same bits, 1.0-1.6 % faster than the 4-board k_tower16b on the three boxes it was timed on.
    int v = g_tower_variant;
    if (v < 0) v = n_boards >= 2048 ? 60 : 36;
    if (v == 36) XQ_TOWER_LAUNCH((k_tower16b<STAMP, 0, 2>), grid2, 256, LDS_BYTES);
    else if (v == 39) XQ_TOWER_LAUNCH((k_tower16b<STAMP, 0, 4>), grid4, 512, LDS_BYTES4);
    else if (v == 0) XQ_TOWER_LAUNCH((k_tower<STAMP>), grid2, 256, LDS_BYTES);
    else if (v == 60) XQ_TOWER_LAUNCH((k_tower1wa<STAMP>), grid4, 256, LDS_BYTES1WA);                  // one wave per SIMD, assembly layer body
#if XQ_TOWER_PROBES
    else if (v == 50) XQ_TOWER_LAUNCH((k_tower1w<STAMP>), grid4, 256, LDS_BYTES1W);                   // one wave per SIMD (results valid)
    else if (STAMP && v == 61) XQ_TOWER_LAUNCH((k_tower1wa<true, 1>), grid4, 256, LDS_BYTES1WA);     // k_tower1wa timing-only bodies (wrong results):
    else if (STAMP && v == 62) XQ_TOWER_LAUNCH((k_tower1wa<true, 2>), grid4, 256, LDS_BYTES1WA);     //   drains read VGPRs / no stores / no drain /
    else if (STAMP && v == 63) XQ_TOWER_LAUNCH((k_tower1wa<true, 3>), grid4, 256, LDS_BYTES1WA);     //   no weight DMA / no barriers / none of the three
    else if (STAMP && v == 64) XQ_TOWER_LAUNCH((k_tower1wa<true, 4>), grid4, 256, LDS_BYTES1WA);
    else if (STAMP && v == 65) XQ_TOWER_LAUNCH((k_tower1wa<true, 5>), grid4, 256, LDS_BYTES1WA);
    else if (STAMP && v == 66) XQ_TOWER_LAUNCH((k_tower1wa<true, 6>), grid4, 256, LDS_BYTES1WA);
    else if (STAMP && v == 67) XQ_TOWER_LAUNCH((k_tower1wa<true, 7>), grid4, 256, LDS_BYTES1WA);     // Winograd go / no-go probes: 2, 3, 4 extra VALU
    else if (STAMP && v == 68) XQ_TOWER_LAUNCH((k_tower1wa<true, 8>), grid4, 256, LDS_BYTES1WA);     //   per MFMA; 3 / 4 with 3.5 x the weight DMA
    else if (STAMP && v == 69) XQ_TOWER_LAUNCH((k_tower1wa<true, 9>), grid4, 256, LDS_BYTES1WA);
    else if (STAMP && v == 70) XQ_TOWER_LAUNCH((k_tower1wa<true, 10>), grid4, 256, LDS_BYTES1WA);
    else if (STAMP && v == 71) XQ_TOWER_LAUNCH((k_tower1wa<true, 11>), grid4, 256, LDS_BYTES1WA);
    else if (STAMP && v == 72) XQ_TOWER_LAUNCH((k_tower1wa<true, 12>), grid4, 256, LDS_BYTES1WA);    // work-removal probes: 46 / 44 of the 48 MFMAs
    else if (STAMP && v == 73) XQ_TOWER_LAUNCH((k_tower1wa<true, 13>), grid4, 256, LDS_BYTES1WA);    //   per K-step (pixel tile 5 without 2 / 4 channel tiles)
    else if (STAMP && v == 51) XQ_TOWER_LAUNCH((k_tower1w<true, 1>), grid4, 256, LDS_BYTES1W);       // ... no stage barriers (wrong results)
    else if (STAMP && v == 52) XQ_TOWER_LAUNCH((k_tower1w<true, 3>), grid4, 256, LDS_BYTES1W);       // ... no barriers, no vmcnt waits
    else if (STAMP && v == 53) XQ_TOWER_LAUNCH((k_tower1w<true, 7>), grid4, 256, LDS_BYTES1W);       // ... and no weight DMA
    else if (STAMP && (row_src || n_rows)) return XQ_E_INVALID;
    else if (STAMP && v == 30) XQ_TOWER_LAUNCH((k_tower16b<true, 1, 2>), grid2, 256, LDS_BYTES);     // no weight refills (wrong results)
    else if (STAMP && v == 31) XQ_TOWER_LAUNCH((k_tower16b<true, 2, 2>), grid2, 256, LDS_BYTES);     // no stage barriers
    else if (STAMP && v == 41) XQ_TOWER_LAUNCH((k_tower16b<true, 2, 4>), grid4, 512, LDS_BYTES4);    // 4 boards, no stage barriers
    else if (STAMP && v == 43) XQ_TOWER_LAUNCH((k_tower16b<true, 3, 4>), grid4, 512, LDS_BYTES4);    // 4 boards, no barriers, no refills
#endif
    else return XQ_E_INVALID;
#undef XQ_TOWER_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

extern "C" int xq_tower_nhwc_bf16(void *stream, const void *planes, const void *w1, const void *wt, const void *bias,
                                  const void *wh, const void *bh, void *policy_out, void *value_out, int n_boards,
                                  int n_blocks, const void *row_src_dev, const void *n_rows_dev)
{
    return launch_tower<false>(stream, planes, w1, wt, bias, wh, bh, policy_out, value_out, n_boards, n_blocks, nullptr,
                               row_src_dev, n_rows_dev);
}

// diagnostic (include/xq_debug.h): s_memtime phase stamps, 64 u64 per workgroup
// (0 start, 1 input conv done, 2 its epilogue, 3+2L / 4+2L main loop / epilogue of layer L, 60 heads
// MFMAs done, 61 end, 62/63 s_memrealtime at start / end)
extern "C" int xq_tower_debug_stamps(void *stream, const void *planes, const void *w1, const void *wt, const void *bias,
                                     const void *wh, const void *bh, void *policy_out, void *value_out, int n_boards,
                                     int n_blocks, void *stamps)
{
    if (!stamps) return XQ_E_INVALID;
    return launch_tower<true>(stream, planes, w1, wt, bias, wh, bh, policy_out, value_out, n_boards, n_blocks, stamps,
                              nullptr, nullptr);
}

#if !XQ_TOWER_PROBES
// (the MFMA / loop probes live in xq_tower_probes.hpp; the symbol stays so that tools fail with a message, not a link error)
extern "C" int xq_mfma_probe(void *, const void *, const void *, void *, int, int, int) { return XQ_E_INVALID; }
#endif

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
// Same tiling as k_conv3x3_b variant B: workgroup = 4 waves = 2 boards x 2 output-channel halves,
// wave tile 96 pixels x 64 channels, weights streamed by LDS-DMA in K-slices [128 cout][64 cin]
// through a double buffer, 80.1 KB LDS -> 2 workgroups per CU.  Builds of that dataflow (xq_tower_set_variant):
//   k_tower16b<PAIR> (36, 8) round 2, the default (36: one read / DMA piece per MFMA gap; 8: clustered in front of a tile's MFMAs): v_mfma_f32_16x16x32_bf16, fully unrolled issue stream; the output channels are
//                        dealt to the MFMA rows so that a lane owns 8 consecutive channels of a pixel and the epilogue
//                        stores 16 bytes per lane, conflict-free (the 8-byte stores of the first build, variant 2, were
//                        4-way conflicted on the 32 store banks: all of the kernel's LDS conflicts)
//   k_tower16s (10, 24)  experiment: 4 boards per 512-thread workgroup share one weight stream (ring of 4 stages, 160 KB
//                        LDS, one workgroup per CU); the two waves of a SIMD belong to two groups that run two stage steps
//                        apart, so that one group's epilogue runs beside the other's MFMA stream.  Half the weight
//                        traffic and a higher clock, but the shared stage barrier leaves the matrix pipe to one wave
//                        whenever the other arrives early: slower than k_tower16b<PAIR>
//   k_tower16 (1)        round 1: the same shape, hand-pipelined stage loop
//   k_tower (0)          v_mfma_f32_32x32x16_bf16, the first version
// The k_tower16b / k_tower16s builds accumulate every output element in the same order and agree to the bit (the builds
// with the skip connection on the VALU, 3 and 33, and the two older kernels differ in the last bit of some elements).
// plus diagnostic entry points (phase stamps, ablation builds, bare-MFMA power probes) used by
// tools/bench_tower.py.  Reference ops: neural_network.py:54-66,181-187 with eval-mode BatchNorm folded.
#include "../../include/xq_selfplay.h"
#include "xq_mfma.hpp"
#include <atomic>
#include <type_traits>

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
constexpr int LDS_BYTES4S = 4 * ACT_BYTES + 4 * WBUF_BYTES + 256 + 4 * 512;     // k_tower16s: ring of 4 stages, bias slots per group

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
template <bool STAMP, int ABL = 0, int NB = 2, bool PAIR = false>
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
    // PAIR: the output channels are dealt to the MFMA rows so that a lane's results of the weight tiles 2j and 2j + 1
    // are 8 consecutive channels = one 16-byte chunk of an activation row (row i of tile mt -> channel
    // hc * 64 + (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3)): the epilogue stores ds_write_b128 and the
    // activation rows are swizzled for it (chunk of K-step ks, lane quarter q at ((q & 1) << 3 | ks << 1 | q >> 1) ^
    // (pixel & 7)): conflict-free stores (8 consecutive pixels -> 8 chunks of one 128-byte half) and fragment reads.
    // Without PAIR the stores are ds_write_b64 at chunk ^ ((pixel & 7) << 1), 4-way conflicts on the 32 store banks.
    auto chan_row = [&](int mt, int i) { return PAIR ? hc * 64 + (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3) : hc * 64 + mt * 16 + i; };
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
    // LDS row rho = piece * 8 + (lane >> 3) of a stage image; with PAIR it holds the weights of channel chan_row(rho):
    // piece P = channel half P >> 3, weight tile (P >> 1) & 3, tile rows (P & 1) * 8 .. -> a scalar row offset per piece
    // plus a lane part
    const int wch = (lane & 7) ^ (lane >> 4);
    const int wsrc_even = (PAIR ? (lane >> 5) * 8 + ((lane >> 3) & 3) : lane >> 3) * 256 + (wch << 4), wsrc_odd = wsrc_even ^ 64;
    auto piece_off = [&](int P) { return PAIR ? ((P >> 3) * 64 + ((P >> 2) & 1) * 32 + (P & 1) * 16 + ((P >> 1) & 1) * 4) * 256 : P * 2048; };
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
    const int Rrow = act_off + r16 * 256, r5 = PAIR ? r16 << 4 : r16 << 5, q4 = PAIR ? (((q & 1) << 3) | (q >> 1)) << 4 : q << 4;
    auto tap_addrs = [&](int (&as)[6], int tap) {
        if (ABL & 4) tap = 4;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, off = dy * 9 + dx;
        // The addresses do not depend on the layer: left alone, the compiler computes all 54 ahead of the
        // layer loop and spills them (each reload then waits vmcnt(0) in the middle of the MFMA stream, a
        // full memory round trip that also drains the weight DMA).  An opaque copy of the row base per
        // call keeps the ~13 VALU instructions of a tap where they are written.
        int rrow = Rrow, r5o = r5;
        asm volatile("" : "+v"(rrow), "+v"(r5o));
        const int slot = PAIR ? ((r5o + off * 16) & 0x70) ^ q4        // (((q & 1) << 3 | q >> 1) ^ (sp & 7)) << 4
                              : ((r5o + off * 32) & 0xE0) ^ q4;       // ((q ^ ((sp & 7) << 1)) << 4)
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
    auto load_b1 = [&](bf16x8 &bf, int a, int nt, int ks) { bf = lds_ld128((a ^ (ks << (PAIR ? 5 : 6))) + nt * 4096); };

    // epilogue of one layer: acc (bias [+ skip] already inside) -> bf16 -> ReLU -> LDS rows in place;
    // READ_X (first convolution of a block): the block input x is read back from those rows first and
    // the next layer's accumulators start at x + bias; otherwise they start at the next layer's bias
    auto epilogue_b64 = [&](auto read_x, int lb_next, auto &&mid) {
        if (ABL & 16) __builtin_amdgcn_s_setprio(3);
        int ln;                                                       // lane id, 2 VALU, not CSE-able (see k_tower16)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int r = ln & 15, qq = ln >> 4;
        int sb[6];
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r < PIX ? nt * 16 + r : 0;
            sb[nt] = act_off + p * 256 + (qq & 1) * 8 + (((hc * 8 + (qq >> 1)) ^ ((p & 7) << 1)) << 4);
        }
        const int lbq = lb_next + (hc * 64 + 4 * qq) * 4;
        // READ_X through the matrix pipe (default; ABL & 64 selects the VALU form it replaced: +1.3 % wall):
        // x + bias = S . X + bias with a 0/1 selector S as the MFMA's A
        // operand (row i of weight tile mt picks input channel (mt & 1) * 16 + i of K-step hc * 2 + (mt >> 1)), X
        // = this wave's own 64 channels of the block input as 12 ordinary B fragments, bias as the C operand:
        // one fp32 addition per element (the MFMA adder's rounding, last-bit differences to v_add_f32), 24 MFMAs + 12 ds_read_b128 instead of 144 VALU + 24 ds_read_b64 —
        // an epilogue runs beside the partner wave's MFMA stream, where VALU issue slots are what is scarce.
        bf16x8 xf[2][6], sel[2];
        if constexpr (decltype(read_x)::value && (ABL & 64) == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                for (int nt = 0; nt < 6; nt++) {
                    const int p = nt * 16 + r < PIX ? nt * 16 + r : 0;
                    xf[k2][nt] = lds_ld128(act_off + p * 256 + ((((hc * 2 + k2) * 4 + qq) ^ ((p & 7) << 1)) << 4));
                }
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    sel[o][j] = (qq == 2 * o + (r >> 3) && j == (r & 7)) ? (__bf16)1.0f : (__bf16)0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const f32x4 bn = lds_ldf4(lbq + mt * 64);
#pragma unroll
            for (int nt = 0; nt < 6; nt++) {
                const f32x4 v = acc[mt][nt];
                const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(v[0], v[1])), relu_bf16x2(pack_bf16x2(v[2], v[3])));
                const int a = sb[nt] ^ (mt << 5);
                if constexpr (decltype(read_x)::value && (ABL & 64) == 0) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[mt & 1], xf[mt >> 1][nt], bn, 0, 0, 0);
                } else if constexpr (decltype(read_x)::value) {
                    const u32x2 x = lds_ld64(a);                                  // same lane, same address: ordered before the store
                    acc[mt][nt] = f32x4{ bf16_lo(x.x) + bn[0], bf16_hi(x.x) + bn[1], bf16_lo(x.y) + bn[2], bf16_hi(x.y) + bn[3] };
                } else acc[mt][nt] = bn;
                if (nt < 5 || r < PIX - 80) lds_st64(a, pk);
            }
        }
        if (ABL & 16) __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue_pair = [&](auto read_x, int lb_next, auto &&mid) {
        if (ABL & 16) __builtin_amdgcn_s_setprio(3);
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
        // x + bias on the matrix pipe as in k_tower16b; the selector follows the channel deal: row i of tile 2j + t
        // picks input channel (i >> 2) * 8 + t * 4 + (i & 3) of K-step hc * 2 + j
        bf16x8 xf[2][6], sel[2];
        if constexpr (decltype(read_x)::value && (ABL & 64) == 0) {
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
            if (j == 1) mid();                                        // second epilogue step
            const f32x4 bn0 = lds_ldf4(lbq + j * 128), bn1 = lds_ldf4(lbq + j * 128 + 16);
#pragma unroll
            for (int nt = 0; nt < 6; nt++) {
                const f32x4 v0 = acc[2 * j][nt], v1 = acc[2 * j + 1][nt];
                const uint4 pk = make_uint4(relu_bf16x2(pack_bf16x2(v0[0], v0[1])), relu_bf16x2(pack_bf16x2(v0[2], v0[3])),
                                            relu_bf16x2(pack_bf16x2(v1[0], v1[1])), relu_bf16x2(pack_bf16x2(v1[2], v1[3])));
                if constexpr (decltype(read_x)::value && (ABL & 64) == 0) {
                    acc[2 * j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[0], xf[j][nt], bn0, 0, 0, 0);
                    acc[2 * j + 1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[1], xf[j][nt], bn1, 0, 0, 0);
                } else if constexpr (decltype(read_x)::value) {       // ABL & 64: the skip connection on the VALU (the lane's own 16 bytes)
                    const u32x4 x = lds_ld128u(sb[nt] ^ (j << 5));    // same lane, same address: ordered before the store
                    acc[2 * j][nt] = f32x4{ bf16_lo(x.x) + bn0[0], bf16_hi(x.x) + bn0[1], bf16_lo(x.y) + bn0[2], bf16_hi(x.y) + bn0[3] };
                    acc[2 * j + 1][nt] = f32x4{ bf16_lo(x.z) + bn1[0], bf16_hi(x.z) + bn1[1], bf16_lo(x.w) + bn1[2], bf16_hi(x.w) + bn1[3] };
                } else {
                    acc[2 * j][nt] = bn0;
                    acc[2 * j + 1][nt] = bn1;
                }
                if (nt < 5 || r < PIX - 80) lds_st128(sb[nt] ^ (j << 5), pk);
                if constexpr (decltype(read_x)::value && (ABL & 64) != 0) {
                    if (nt & 1) __builtin_amdgcn_sched_barrier(0);    // (keeps the 12 x loads from being hoisted together: spills)
                }
            }
        }
        if (ABL & 16) __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue = [&](auto read_x, int lb_next) {
        if constexpr (PAIR) epilogue_pair(read_x, lb_next, [] {});
        else epilogue_b64(read_x, lb_next, [] {});
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
                    if (ABL & 32) {
                        // one filler per MFMA gap: a 16-cycle MFMA leaves room for ~2 issue slots beside it, so two
                        // reads, the m0 write and a DMA piece in ONE gap stall the matrix pipe
                        if (nrd >= 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (kk == 1 && n >= 1 && n - 1 < PPW && !(ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    } else {
                    if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else if (nrd == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (kk == 1 && n >= 1 && n - 1 < PPW && !(ABL & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    }
                    if (kk == 1 && n == 0 && !(ABL & 2)) {
                        // stage barrier: the next stage has landed (every wave drains its own pieces first) and,
                        // this tile's MFMAs having consumed all 4 weight fragments of buffer sl, nobody reads it any more
                        // (sched_barrier: the tile's MFMAs, and with them the waits for their operands, stay above)
                        __builtin_amdgcn_sched_barrier(0);
                        if (ABL & 8) { }                                   // ablation: pieces are issued but never waited for
                        else if (last) __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0) lgkmcnt(0): + all activation reads done
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
            hb[nt] = lds_ld128(act_off + p * 256 + ((PAIR ? (((q & 1) << 3) | (ks << 1) | (q >> 1)) ^ (p & 7) : (ks * 4 + q) ^ ((p & 7) << 1)) << 4));
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

template <bool STAMP, int ABLS = 0, bool PAIR = false>
__global__ __launch_bounds__(512, 2) void k_tower16s(TowerArgs A)
{
    constexpr int NB = 4, NWV = 8, PPW = 2, ABL = (ABLS & 2) ? 16 : 0;                  // boards, waves, weight pieces per wave and stage
    constexpr int ACT0 = 4 * WBUF_BYTES, ZROW = ACT0 + NB * ACT_BYTES, BIAS = ZROW + 256;   // ring of 4 stages; bias: [2][128] f32

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
    // fine stamps inside the epilogues of layers 2 (first convolution of a block) and 3: slots 30 + 6 * (layer - 2) + i
    // for the lead group (thread 0), + 12 for the lag group (thread 256)
    auto fstamp = [&](int layer, int i) {
        if constexpr (STAMP) {
            if (layer == 2 || layer == 3) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 255) == 0)
                    A.stamps[(size_t)blockIdx.x * 64 + 30 + 6 * (layer - 2) + i + 12 * (threadIdx.x >> 8)] = t;
            }
        }
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb_ = wave >> 1, hc = wave & 1;                         // board in workgroup, channel half
    const int grp = wave >> 2;                                        // 0: boards 0, 1 (leads); 1: boards 2, 3 (two stage steps behind)
    // PAIR: the output channels are dealt to the MFMA rows so that a lane's results of the weight tiles 2j and 2j + 1
    // are 8 consecutive channels = one 16-byte chunk of an activation row (row i of tile mt -> channel
    // hc * 64 + (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3)): the epilogue stores ds_write_b128 and the
    // activation rows are swizzled for it (chunk of K-step ks, lane quarter q at ((q & 1) << 3 | ks << 1 | q >> 1) ^
    // (pixel & 7)): conflict-free stores (8 consecutive pixels -> 8 chunks of one 128-byte half) and fragment reads.
    // Without PAIR the stores are ds_write_b64 at chunk ^ ((pixel & 7) << 1), 4-way conflicts on the 32 store banks.
    auto chan_row = [&](int mt, int i) { return PAIR ? hc * 64 + (mt >> 1) * 32 + (i >> 2) * 8 + (mt & 1) * 4 + (i & 3) : hc * 64 + mt * 16 + i; };
    const int board = blockIdx.x * NB + wb_;
    const bool board_ok = board < A.G;
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
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.planes) + (size_t)board * PIX * 32;
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
    // LDS row rho = piece * 8 + (lane >> 3) of a stage image; with PAIR it holds the weights of channel chan_row(rho):
    // piece P = channel half P >> 3, weight tile (P >> 1) & 3, tile rows (P & 1) * 8 .. -> a scalar row offset per piece
    // plus a lane part
    const int wch = (lane & 7) ^ (lane >> 4);
    const int wsrc_even = (PAIR ? (lane >> 5) * 8 + ((lane >> 3) & 3) : lane >> 3) * 256 + (wch << 4), wsrc_odd = wsrc_even ^ 64;
    auto piece_off = [&](int P) { return PAIR ? ((P >> 3) * 64 + ((P >> 2) & 1) * 32 + (P & 1) * 16 + ((P >> 1) & 1) * 4) * 256 : P * 2048; };
    auto stage_piece = [&](int g, int slot, int j) {                  // stage g -> ring slot; j < PPW
        const int soff = (g >> 1) * (COUT * COUT * 2) + (g & 1) * 128 + piece_off(wave * PPW + j);
        dma16_buf_abs(wrsrc, (j & 1) ? wsrc_odd : wsrc_even, soff, slot * WBUF_BYTES + (wave * PPW + j) * 1024);
    };
    // what the wave issues behind its own barrier #p of a layer (p 0..17 = stage barriers, 18 / 19 = the two epilogue
    // barriers): the lead group is at the lag group's barrier p + 2, so the same stage comes out of both formulas;
    // where a group has nothing to fetch (lead at 0, 1; lag at 18, 19) it re-fetches a resident stage into its own
    // slot (identical bytes), so that every wave issues PPW pieces behind every barrier and no branch is needed
    auto fill = [&](int base, int p, int j) {
        int g;
        if (p >= 2 && p <= 17) g = base + p + 2 + 2 * grp;
        else if (p < 2) g = base + p + 4 * grp;
        else g = base + p + 2 * (1 - grp);
        stage_piece(g < nstages ? g : nstages - 1, g & 3, j);
    };
    // A fragment (weight tile mt, K-step kk of a stage): abase ^ (kk << 6), + mt * 2048
    const int abase = (hc * 64 + r16) * 128 + ((q ^ ((r16 >> 1) & 7)) << 4);
    // ring slot of stage p of a layer = (layer * 18 + p) & 3 = (p & 3) ^ (2 * (layer & 1)): bit 0 rides in the immediate,
    // bit 1 in one of two per-layer base addresses
    int bsel[2] = { abase, abase + 2 * WBUF_BYTES };
    auto load_a1 = [&](bf16x8 &af, int mt, int p, int kk) {
        af = lds_ld128((bsel[(p >> 1) & 1] ^ (kk << 6)) + (p & 1) * WBUF_BYTES + mt * 2048);
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
    const int Rrow = act_off + r16 * 256, r5 = PAIR ? r16 << 4 : r16 << 5, q4 = PAIR ? (((q & 1) << 3) | (q >> 1)) << 4 : q << 4;
    auto tap_addrs = [&](int (&as)[6], int tap) {
        if (ABL & 4) tap = 4;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, off = dy * 9 + dx;
        // The addresses do not depend on the layer: left alone, the compiler computes all 54 ahead of the
        // layer loop and spills them (each reload then waits vmcnt(0) in the middle of the MFMA stream, a
        // full memory round trip that also drains the weight DMA).  An opaque copy of the row base per
        // call keeps the ~13 VALU instructions of a tap where they are written.
        int rrow = Rrow, r5o = r5;
        asm volatile("" : "+v"(rrow), "+v"(r5o));
        const int slot = PAIR ? ((r5o + off * 16) & 0x70) ^ q4        // (((q & 1) << 3 | q >> 1) ^ (sp & 7)) << 4
                              : ((r5o + off * 32) & 0xE0) ^ q4;       // ((q ^ ((sp & 7) << 1)) << 4)
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
    auto load_b1 = [&](bf16x8 &bf, int a, int nt, int ks) { bf = lds_ld128((a ^ (ks << (PAIR ? 5 : 6))) + nt * 4096); };

    // epilogue of one layer: acc (bias [+ skip] already inside) -> bf16 -> ReLU -> LDS rows in place;
    // READ_X (first convolution of a block): the block input x is read back from those rows first and
    // the next layer's accumulators start at x + bias; otherwise they start at the next layer's bias
    auto epilogue_b64 = [&](auto read_x, int lb_next, auto &&mid) {
        if (ABL & 16) __builtin_amdgcn_s_setprio(3);
        int ln;                                                       // lane id, 2 VALU, not CSE-able (see k_tower16)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int r = ln & 15, qq = ln >> 4;
        int sb[6];
#pragma unroll
        for (int nt = 0; nt < 6; nt++) {
            const int p = nt * 16 + r < PIX ? nt * 16 + r : 0;
            sb[nt] = act_off + p * 256 + (qq & 1) * 8 + (((hc * 8 + (qq >> 1)) ^ ((p & 7) << 1)) << 4);
        }
        const int lbq = lb_next + (hc * 64 + 4 * qq) * 4;
        // READ_X through the matrix pipe (default; ABL & 64 selects the VALU form it replaced: +1.3 % wall):
        // x + bias = S . X + bias with a 0/1 selector S as the MFMA's A
        // operand (row i of weight tile mt picks input channel (mt & 1) * 16 + i of K-step hc * 2 + (mt >> 1)), X
        // = this wave's own 64 channels of the block input as 12 ordinary B fragments, bias as the C operand:
        // one fp32 addition per element (the MFMA adder's rounding, last-bit differences to v_add_f32), 24 MFMAs + 12 ds_read_b128 instead of 144 VALU + 24 ds_read_b64 —
        // an epilogue runs beside the partner wave's MFMA stream, where VALU issue slots are what is scarce.
        bf16x8 xf[2][6], sel[2];
        if constexpr (decltype(read_x)::value && (ABL & 64) == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                for (int nt = 0; nt < 6; nt++) {
                    const int p = nt * 16 + r < PIX ? nt * 16 + r : 0;
                    xf[k2][nt] = lds_ld128(act_off + p * 256 + ((((hc * 2 + k2) * 4 + qq) ^ ((p & 7) << 1)) << 4));
                }
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    sel[o][j] = (qq == 2 * o + (r >> 3) && j == (r & 7)) ? (__bf16)1.0f : (__bf16)0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            if (mt == 2) mid();                                       // second epilogue step
            const f32x4 bn = lds_ldf4(lbq + mt * 64);
#pragma unroll
            for (int nt = 0; nt < 6; nt++) {
                const f32x4 v = acc[mt][nt];
                const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(v[0], v[1])), relu_bf16x2(pack_bf16x2(v[2], v[3])));
                const int a = sb[nt] ^ (mt << 5);
                if constexpr (decltype(read_x)::value && (ABL & 64) == 0) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[mt & 1], xf[mt >> 1][nt], bn, 0, 0, 0);
                } else if constexpr (decltype(read_x)::value) {
                    const u32x2 x = lds_ld64(a);                                  // same lane, same address: ordered before the store
                    acc[mt][nt] = f32x4{ bf16_lo(x.x) + bn[0], bf16_hi(x.x) + bn[1], bf16_lo(x.y) + bn[2], bf16_hi(x.y) + bn[3] };
                } else acc[mt][nt] = bn;
                if (nt < 5 || r < PIX - 80) lds_st64(a, pk);
            }
        }
        if (ABL & 16) __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue_pair = [&](auto read_x, int lb_next, auto &&mid) {
        if (ABL & 16) __builtin_amdgcn_s_setprio(3);
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
        // x + bias on the matrix pipe as in k_tower16b; the selector follows the channel deal: row i of tile 2j + t
        // picks input channel (i >> 2) * 8 + t * 4 + (i & 3) of K-step hc * 2 + j
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
            if (j == 1) mid();                                        // second epilogue step
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
        if (ABL & 16) __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue = [&](auto read_x, int lb_next, auto &&mid) {
        if constexpr (PAIR) epilogue_pair(read_x, lb_next, mid);
        else epilogue_b64(read_x, lb_next, mid);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    if (nstages > 0) {
#pragma unroll
        for (int j = 0; j < PPW; j++) { stage_piece(0, 0, j); stage_piece(1, 1, j); stage_piece(2, 2, j); stage_piece(3, 3, j); }
    }
    epilogue(no{}, BIAS + 512, [] {});                              // conv1 output; tower layer 0 starts at bias[1]
    barrier_dma();
    stamp(2);

    // ---------------------------------------------------------------- residual tower
    // Every wave executes 2 + 20 * nlayers workgroup barriers: the lag group 2 idle ones first, the lead group 2 idle
    // ones last.  Own barrier #p of a layer: p 0..17 in the middle of stage p (as in k_tower16b), 18 between the two
    // halves of the epilogue, 19 behind it.  So the lead group's epilogue runs beside the lag group's stages 16, 17
    // and the lag group's beside the lead group's stages 0, 1 of the next layer: a SIMD's two waves are never both
    // off the matrix pipe.
    auto idle_step = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 1) { idle_step(); idle_step(); }
    bf16x8 fa[2][4], fb[2][6];                                       // fragments, double-buffered by K-step parity
    for (int layer = 0; layer < nlayers; layer++) {
        const int base = layer * 18, par = layer & 1;
        bsel[0] = abase + par * 2 * WBUF_BYTES;
        bsel[1] = abase + (par ^ 1) * 2 * WBUF_BYTES;
        if (wave == 1 + 4 * grp && lane < 32 && layer + 1 < nlayers)   // bias of tower layer L + 1 (row L + 2) -> this group's slot L & 1
            dma16_abs(A.bias + (size_t)(layer + 2) * 128 + lane * 4, BIAS + grp * 1024 + (layer & 1) * 512);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) load_a1(fa[0][mt], mt, 0, 0);
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
                const int p = tap * 2 + sl;                           // stage of the layer
                const bool last = tap == 8 && ks == 3;                // last K-step of the layer: no prefetch of any kind
                if (ks == 3 && !last) tap_addrs(asn, tap + 1);
#pragma unroll
                for (int n = 0; n < 6; n++) {                        // pixel tile n: 4 MFMAs
                    if (!last) {
                        if (ks < 3) load_b1(fb[cur ^ 1][n], as[n], n, ks + 1);
                        else load_b1(fb[cur ^ 1][n], asn[n], n, 0);
                    }
                    if (kk == 0 && n < 4) load_a1(fa[cur ^ 1][n], n, p, 1);
                    if (kk == 1 && n >= 1 && n < 5 && !last) load_a1(fa[cur ^ 1][n - 1], n - 1, p + 1, 0);
                    if (kk == 1 && n >= 1 && n - 1 < PPW) fill(base, p, n - 1);
                    // ABLS & 1 (timing probe, wrong results): the last two waves leave out their sixth pixel tile — the
                    // MFMA count of 23 instead of 24 pixel tiles per 4 boards, with no change to the critical path
                    if (!((ABLS & 1) && n == 5 && wave >= 6))
#pragma unroll
                    for (int mt = 0; mt < 4; mt++)
                        acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][mt], fb[cur][n], acc[mt][n], 0, 0, 0);
                    const int nrd = (last ? 0 : 1) + ((kk == 0 && n < 4) || (kk == 1 && n >= 1 && n < 5 && !last) ? 1 : 0);
                    if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else if (nrd == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (kk == 1 && n >= 1 && n - 1 < PPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    if (kk == 1 && n == 0) {                         // own barrier #p
                        __builtin_amdgcn_sched_barrier(0);
                        if (last) __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): + all activation reads done
                        else __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
                        if constexpr (STAMP) {                             // arrival / release at stage barriers 8 and 9 of layer 2: slots 54.. (+ 3 for the lag group)
                            if (layer == 2 && (p == 8 || p == 9)) {
                                const unsigned long long t = __builtin_amdgcn_s_memtime();
                                if ((threadIdx.x & 255) == 0) A.stamps[(size_t)blockIdx.x * 64 + 54 + (p - 8) * 2 + 3 * (threadIdx.x >> 8)] = t;
                            }
                        }
                        __builtin_amdgcn_s_barrier();
                        if constexpr (STAMP) {
                            if (layer == 2 && p == 8) {
                                const unsigned long long t = __builtin_amdgcn_s_memtime();
                                if ((threadIdx.x & 255) == 0) A.stamps[(size_t)blockIdx.x * 64 + 55 + 3 * (threadIdx.x >> 8)] = t;
                            }
                        }
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
        fstamp(layer, 0);
        auto mid = [&]() {                                            // own barrier #18, between the epilogue's halves
            __builtin_amdgcn_sched_barrier(0);
            fstamp(layer, 5);                                         // first half issued
            if constexpr (STAMP) __builtin_amdgcn_s_waitcnt(0x0070);  // (the next stamp wants its LDS traffic done)
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            fstamp(layer, 1);
            __builtin_amdgcn_s_barrier();
            fstamp(layer, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (layer + 1 < nlayers) { fill(base, 18, 0); fill(base, 18, 1); }
        };
        if (layer & 1) epilogue(no{}, BIAS + grp * 1024 + (layer & 1) * 512, mid);
        else epilogue(yes{}, BIAS + grp * 1024 + (layer & 1) * 512, mid);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);                          // own barrier #19: vmcnt(0) lgkmcnt(0), the layer's output is complete
        fstamp(layer, 3);
        __builtin_amdgcn_s_barrier();
        fstamp(layer, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (layer + 1 < nlayers) { fill(base, 19, 0); fill(base, 19, 1); }
        if (layer < 28) stamp(4 + 2 * layer);
    }
    if (grp == 0) { idle_step(); idle_step(); }

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
            hb[nt] = lds_ld128(act_off + p * 256 + ((PAIR ? (((q & 1) << 3) | (ks << 1) | (q >> 1)) ^ (p & 7) : (ks * 4 + q) ^ ((p & 7) << 1)) << 4));
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

static int g_tower_variant = 36;    // 36 = k_tower16b<PAIR>, one filler per MFMA gap (default); 8 = the same with the fillers clustered; 0 = k_tower (32x32x16 comparison build); 24 / 29 = 4 boards per workgroup (k_tower16s staggered groups / lock-step); 11, 25, 30, 31 = timing probes (XQ_TOWER_PROBES builds, wrong results)
// diagnostic switch (not part of the public ABI): the builds compute the same function
extern "C" void xq_tower_set_variant(int v) { g_tower_variant = v; }

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
                 (unsigned long long *)stamps, (const int32_t *)row_src, (const int32_t *)n_rows };
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid2((n_boards + 1) / 2), grid4((n_boards + 3) / 4);
#define XQ_TOWER_LAUNCH(KERNEL, GRID, THREADS, LDS)                                                     \
    do {                                                                                                \
        if (int rc = tower_lds_opt_in<&KERNEL>(LDS)) return rc;                                         \
        hipLaunchKernelGGL((KERNEL), GRID, dim3(THREADS), LDS, st, a);                                  \
    } while (0)
    const int v = g_tower_variant;
    if ((row_src || n_rows) && v != 36 && v != 8 && v != 0) return XQ_E_INVALID;   // the experiments take no row map
    if (v == 36) XQ_TOWER_LAUNCH((k_tower16b<STAMP, 32, 2, true>), grid2, 256, LDS_BYTES);
    else if (v == 8) XQ_TOWER_LAUNCH((k_tower16b<STAMP, 0, 2, true>), grid2, 256, LDS_BYTES);
    else if (v == 0) XQ_TOWER_LAUNCH((k_tower<STAMP>), grid2, 256, LDS_BYTES);
    else if (v == 24) XQ_TOWER_LAUNCH((k_tower16s<STAMP, 0, true>), grid4, 512, LDS_BYTES4S);
    else if (v == 29) XQ_TOWER_LAUNCH((k_tower16b<STAMP, 0, 4, true>), grid4, 512, LDS_BYTES4);
#if XQ_TOWER_PROBES
    else if (STAMP && v == 30) XQ_TOWER_LAUNCH((k_tower16b<true, 1, 2, true>), grid2, 256, LDS_BYTES);   // no weight refills
    else if (STAMP && v == 31) XQ_TOWER_LAUNCH((k_tower16b<true, 2, 2, true>), grid2, 256, LDS_BYTES);   // no stage barriers
    else if (STAMP && v == 25) XQ_TOWER_LAUNCH((k_tower16s<true, 2, true>), grid4, 512, LDS_BYTES4S);    // s_setprio 3 in epilogues
    else if (STAMP && v == 11) XQ_TOWER_LAUNCH((k_tower16s<true, 1, true>), grid4, 512, LDS_BYTES4S);    // 23 of 24 pixel tiles (power probe)
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

// diagnostic only (not part of the public ABI): s_memtime phase stamps, 64 u64 per workgroup
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

// diagnostic only: what the matrix pipes sustain under this board's power cap — 2 waves per SIMD
// issuing v_mfma_f32_32x32x16_bf16 back to back (6 accumulators per wave like the conv tile).
//   mode 0: operands fixed in registers (no other activity)
//   mode 1: operands re-read from LDS for every K-step in the conv's 2 A + 3 B pattern (random bf16
//           data, half of the B values zero like post-ReLU activations), no barriers, no DMA
//   mode 2: mode 1 + the conv's weight stream (4 LDS-DMA pieces of 1 KB per 24 MFMAs from a 3.5 MB
//           L2-resident buffer), still no barriers
// tools/bench_tower.py prints these beside the kernel: the gap between them is what the data
// movement costs in clock under the power cap.
namespace {
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mfma_probe(const uint32_t *seed, const uint8_t *wsrc, float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t s = seed[lane] + blockIdx.x * 2654435761u + tid * 40503u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    if (MODE > 0) {
        // 32 KB "weights" (dense random) + 46 KB "activations" (half zeros), bf16 in [-1, 1)
        for (int i = tid; i < (32768 + 46080) / 4; i += 256) {
            const uint32_t r = rnd();
            const float a = ((int)((r >> 8) & 2047) - 1024) * (1.0f / 1024.0f), b = ((int)((r >> 20) & 2047) - 1024) * (1.0f / 1024.0f);
            uint32_t w = pack_bf16x2(a, b);
            if (i >= 8192) w = relu_bf16x2(w);
            reinterpret_cast<uint32_t *>(lds)[i] = w;
        }
        barrier_dma();
    }
    bf16x8 a[2], b[3];
    for (int i = 0; i < 8; i++) {
        a[0][i] = (__bf16)(((int)(rnd() >> 9) % 2048 - 1024) * (1.0f / 1024.0f));
        a[1][i] = (__bf16)(((int)(rnd() >> 9) % 2048 - 1024) * (1.0f / 1024.0f));
        for (int n = 0; n < 3; n++) {
            const float v = ((int)(rnd() >> 9) % 2048 - 1024) * (1.0f / 1024.0f);
            b[n][i] = (__bf16)(v > 0.f ? v : 0.f);
        }
    }
    f32x16 acc[2][3];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 3; n++) for (int i = 0; i < 16; i++) acc[m][n][i] = 0.f;
    const int r32 = lane & 31, h = lane >> 5, hc = wave & 1, bd = wave >> 1;
    const int abase = (hc * 64 + r32) * 128 + ((h ^ ((r32 >> 1) & 7)) << 4);
    rsrc_t wr = make_rsrc(wsrc, 216 * 16384);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                dma16_buf(wr, lane * 16, ((it % 216) * 16 + wave * 4 + j) * 1024, lds + 78848 + (wave * 4 + j) * 1024 * 0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE > 0) {
                const int sl = it & 1;
                const int px = (it * 7 + u * 3) % 58;                       // sliding pixel window like the taps
#pragma unroll
                for (int m = 0; m < 2; m++)
                    a[m] = *reinterpret_cast<const bf16x8 *>(lds + (abase ^ (u << 5)) + sl * 16384 + m * 4096);
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    const int sp = (px + n * 32 + r32) % 90;
                    b[n] = *reinterpret_cast<const bf16x8 *>(lds + 32768 + bd * 23040 + sp * 256 +
                                                               ((((sp & 15) ^ h) << 4) ^ ((sl * 8 + u * 2) << 4)));
                }
            }
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 3; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int m = 0; m < 2; m++) for (int n = 0; n < 3; n++) for (int i = 0; i < 16; i++) t += acc[m][n][i];
    if (t == 123.456f) out[0] = t;
    if (blockIdx.x == 0 && tid == 0) {       // core cycles and 100 MHz ticks of the loop -> clock under load
        out[1] = (float)(__builtin_amdgcn_s_memtime() - c0);
        out[2] = (float)(__builtin_amdgcn_s_memrealtime() - r0);
    }
}
}  // namespace

// the same bare loop on v_mfma_f32_16x16x32_bf16 (24 accumulators of 4 registers: the same 96 x 64
// output tile per wave, the same FLOPs per iteration): which shape the chip clocks higher on
namespace {
template <int ORDER>        // 0: pixel tile outer, weight tile inner (the kernels' order: B operand constant over 4 MFMAs); 1: weight tile outer (A constant over 6)
__global__ __launch_bounds__(256, 2) void k_mfma_probe16(const uint32_t *seed, float *out, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t s = seed[lane] + blockIdx.x * 2654435761u + tid * 40503u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    bf16x8 a[4], b[6];
    for (int i = 0; i < 8; i++) {
        for (int m = 0; m < 4; m++) a[m][i] = (__bf16)(((int)(rnd() >> 9) % 2048 - 1024) * (1.0f / 1024.0f));
        for (int n = 0; n < 6; n++) {
            const float v = ((int)(rnd() >> 9) % 2048 - 1024) * (1.0f / 1024.0f);
            b[n][i] = (__bf16)(v > 0.f ? v : 0.f);
        }
    }
    f32x4 acc[4][6];
    for (int m = 0; m < 4; m++) for (int n = 0; n < 6; n++) for (int i = 0; i < 4; i++) acc[m][n][i] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (ORDER == 0) {
#pragma unroll
                for (int n = 0; n < 6; n++)
#pragma unroll
                    for (int m = 0; m < 4; m++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int n = 0; n < 6; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
    }
    float t = 0.f;
    for (int m = 0; m < 4; m++) for (int n = 0; n < 6; n++) for (int i = 0; i < 4; i++) t += acc[m][n][i];
    if (t == 123.456f) out[0] = t;
    if (blockIdx.x == 0 && tid == 0) {
        out[1] = (float)(__builtin_amdgcn_s_memtime() - c0);
        out[2] = (float)(__builtin_amdgcn_s_memrealtime() - r0);
    }
}
}  // namespace

extern "C" int xq_mfma_probe(void *stream, const void *seed64_dev, const void *weights_dev, void *out_dev, int n_workgroups,
                             int iters, int mode)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mode == 16) {       // 48 MFMAs of 16x16x32 per iteration = the FLOPs of 24 of 32x32x16
        hipLaunchKernelGGL(k_mfma_probe16<0>, dim3(n_workgroups), dim3(256), 0, st, (const uint32_t *)seed64_dev, (float *)out_dev, iters);
        return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
    }
    if (mode == 17) {       // the same with the weight tile in the outer loop
        hipLaunchKernelGGL(k_mfma_probe16<1>, dim3(n_workgroups), dim3(256), 0, st, (const uint32_t *)seed64_dev, (float *)out_dev, iters);
        return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
    }
    const int lds_bytes = mode ? 80128 : 0;
    if (mode == 0) hipLaunchKernelGGL(k_mfma_probe<0>, dim3(n_workgroups), dim3(256), 0, st, (const uint32_t *)seed64_dev, (const uint8_t *)weights_dev, (float *)out_dev, iters);
    else {
        const void *f = mode == 1 ? reinterpret_cast<const void *>(&k_mfma_probe<1>) : reinterpret_cast<const void *>(&k_mfma_probe<2>);
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return XQ_E_HIP;
        if (mode == 1) hipLaunchKernelGGL(k_mfma_probe<1>, dim3(n_workgroups), dim3(256), lds_bytes, st, (const uint32_t *)seed64_dev, (const uint8_t *)weights_dev, (float *)out_dev, iters);
        else hipLaunchKernelGGL(k_mfma_probe<2>, dim3(n_workgroups), dim3(256), lds_bytes, st, (const uint32_t *)seed64_dev, (const uint8_t *)weights_dev, (float *)out_dev, iters);
    }
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

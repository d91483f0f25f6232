// xq_policy.hip — the policy head's fully-connected layer (neural_network.py:39,64: Linear 2880 -> 8100)
// as a hand-written bf16 MFMA GEMM for gfx950, replacing the library (hipBLASLt stream-K) kernel:
//
//   logits[m][n] = bias[n] + sum_k act[m][k] * w[n][k]      act [M][K] bf16, w [N][K] bf16 (both K-contiguous),
//                                                          bias f32, logits [M][N] bf16, fp32 accumulation
//
// Why hand-written: (1) a fixed accumulation order — the library's stream-K split made self-play with the
// real network differ from run to run; every output element here is one fp32 chain
// over k = 0 .. K-1 in 32-wide MFMA steps, identical for every launch, tile position and batch size;
// (2) the caller restricts N to the columns a legal move can index (2,294 of 8,100, padded to 2,304:
// neural_network.reachable_policy_columns) — the search only ever gathers legal moves
// (neural_network.py:148-169), so the other 5,806 logits are dead outputs.
//
// Tiling: workgroup = 8 waves = 256 (M) x 192 (N) outputs, K-stage 64; wave (wm, wn) owns 64 x 96 = 4 x 6
// MFMA tiles (96 accumulator registers: the trunk kernel's wave tile, same 10 fragment reads per 24 MFMAs).
// 2,304 = 12 x 192 and 16,384 = 64 x 256: 768 workgroups = exactly 3 per CU.  Both operands are streamed by
// LDS-DMA into a double buffer of [256 + 192 rows][128 B] with the trunk kernel's XOR chunk swizzle (applied
// on the source address), 7 pieces of 1 KB per wave and stage; stage barriers are raw s_barrier + vmcnt(0)
// behind the first tile of a stage's second K-step, fragments are double-buffered by K-step (as in
// k_tower16b).  The MFMA's A operand is the WEIGHT tile, so a lane ends up with 4 consecutive columns of one
// row: 8-byte stores.  blockIdx -> tile is XCD-aware: the 12 column tiles of a row block run on one XCD,
// so the activations are fetched into that L2 once.
#include "../../include/xq_selfplay.h"
#include "../../include/xq_debug.h"
#include "xq_mfma.hpp"
#include <atomic>
#include <type_traits>

#ifndef XQ_TOWER_PROBES
#define XQ_TOWER_PROBES 0
#endif

namespace {
using namespace xqm;

constexpr int BM = 256, BN = 192;
constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;     // 57,344
constexpr int FC_LDS_BYTES = 2 * STAGE_BYTES;                                              // 114,688

struct FcArgs {
    const uint16_t *act;      // [M][K]
    const uint16_t *w;        // [N][K]
    const float *bias;        // [N]
    uint16_t *out;            // [M][N]
    int M, N, K, ncol;
    const int32_t *n_rows;    // optional device value: only rows below *n_rows exist (evaluator row compaction)
    unsigned long long *stamps;   // (stamped probe bodies of k_policy_fc1w only: 8 uint64 per wave, 32 per workgroup)
};

// ABL (timing only, wrong results; -DXQ_TOWER_PROBES=1 libraries through xq_policy_fc_debug): 1 = no operand DMA behind the
// first two stages (what the MFMA stream + fragment reads + barriers take), 2 = no MFMAs (what the operand delivery takes)
template <int ABL = 0>
__global__ __launch_bounds__(512, 2) void k_policy_fc(FcArgs P)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;

    // XCD-aware, bijective blockIdx -> tile (cdna_hip_programming.md T1): blocks b, b + 8, ... share an XCD
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
    const int t = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
    const int rb = t / P.ncol, cb = t - rb * P.ncol;
    const int m0 = rb * BM, n0 = cb * BN;
    const int K2 = P.K * 2, nst = P.K >> 6;
    int M = P.M;
    if (P.n_rows) { const int n = *P.n_rows; M = n < M ? n : M; }
    if (m0 >= M) return;                                              // compaction: no row in this tile (uniform, before any barrier)
    const int rows = M - m0 < BM ? M - m0 : BM;                       // rows past M read as zeros (buffer bounds)
    const rsrc_t ra = make_rsrc(P.act + (size_t)m0 * P.K, rows * K2);
    const rsrc_t rw = make_rsrc(P.w + (size_t)n0 * P.K, BN * K2);

    // piece p (8 rows x 128 B) of an operand tile: rows p * 8 + (lane >> 3); its source chunk is swizzled by
    // (row >> 1) & 7 = (lane >> 4) + 4 * (p & 1); pieces are dealt p = j * 8 + wave, so p & 1 = wave & 1
    const int voff = (lane >> 3) * K2 + ((((lane & 7) ^ (lane >> 4)) << 4) ^ ((wave & 1) << 6));
    auto stage_piece = [&](int s, int buf, int j) {                   // j 0..3: activations, 4..6: weights
        if (ABL == 1 && s >= 2) return;
        if (j < 4) {
            const int p = j * 8 + wave;
            dma16_buf_abs(ra, voff, p * 8 * K2 + s * 128, buf * STAGE_BYTES + p * 1024);
        } else {
            const int p = (j - 4) * 8 + wave;
            dma16_buf_abs(rw, voff, p * 8 * K2 + s * 128, buf * STAGE_BYTES + A_BYTES + p * 1024);
        }
    };
    const int swz = (q ^ ((r16 >> 1) & 7)) << 4;
    const int xbase = (wm * 64 + r16) * 128 + swz, wbase = A_BYTES + (wn * 96 + r16) * 128 + swz;
    auto load_x = [&](bf16x8 &f, int mt, int buf, int kk) { f = lds_ld128((xbase ^ (kk << 6)) + buf * STAGE_BYTES + mt * 2048); };
    auto load_w = [&](bf16x8 &f, int nt, int buf, int kk) { f = lds_ld128((wbase ^ (kk << 6)) + buf * STAGE_BYTES + nt * 2048); };

    f32x4 acc[6][4];
#pragma unroll
    for (int nt = 0; nt < 6; nt++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[nt][mt] = f32x4{ 0.f, 0.f, 0.f, 0.f };

#pragma unroll
    for (int j = 0; j < 7; j++) stage_piece(0, 0, j);
#pragma unroll
    for (int j = 0; j < 7; j++) stage_piece(nst > 1 ? 1 : 0, 1, j);
    bf16x8 fw[2][6], fx[2][4];
    barrier_dma();
#pragma unroll
    for (int nt = 0; nt < 6; nt++) load_w(fw[0][nt], nt, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; mt++) load_x(fx[0][mt], mt, 0, 0);

    // one stage = 2 K-steps of 32; buffer = stage parity (compile-time inside the pair)
    auto stage = [&](int s, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int cur = kk;
#pragma unroll
            for (int n = 0; n < 6; n++) {                            // weight tile n: 4 MFMAs
                // next K-step's fragments: kk == 0 from this stage; kk == 1 from the stage the barrier (behind tile 0)
                // has just published, together with the refill of this buffer two stages ahead
                if (kk == 0) {
                    load_w(fw[1][n], n, buf, 1);
                    if (n < 4) load_x(fx[1][n], n, buf, 1);
                } else if (n >= 1) {
                    load_w(fw[0][n - 1], n - 1, buf ^ 1, 0);
                    if (n < 5) load_x(fx[0][n - 1], n - 1, buf ^ 1, 0);
                    const int s2 = s + 2 < nst ? s + 2 : nst - 1;     // (the last two stages refetch the final one: no branch)
                    stage_piece(s2, buf, n - 1);
                    if (n >= 4) stage_piece(s2, buf, n + 1);          // pieces 5, 6 ride with tiles 4, 5
                }
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    if (ABL == 2) asm volatile("" : : "v"(fw[cur][n]), "v"(fx[cur][mt]));          // (keeps the fragment reads alive)
                    else acc[n][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[cur][n], fx[cur][mt], acc[n][mt], 0, 0, 0);
                }
                const int nrd = kk == 0 ? (n < 4 ? 2 : 1) : (n == 0 ? 0 : n < 5 ? 2 : 1);
                if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (nrd == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kk == 1 && n >= 4) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                else if (kk == 1 && n >= 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                if (kk == 1 && n == 0) {
                    // tile 0's MFMAs have consumed... (weight fragment 0 only; the other fragments of this buffer are
                    // in registers too: every read of the K-step was issued during the previous one and LDS returns
                    // in order) -> wait for them explicitly, then retire the buffer
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0)
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kk == 1) {                                            // last weight fragment of the next stage
                load_w(fw[0][5], 5, buf ^ 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    using b0 = std::integral_constant<int, 0>;
    using b1 = std::integral_constant<int, 1>;
    int s = 0;
    for (; s + 1 < nst; s += 2) {
        stage(s, b0{});
        stage(s + 1, b1{});
    }
    if (s < nst) stage(s, b0{});
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // no LDS-DMA may outlive the workgroup's LDS

    // epilogue: + bias, bf16, 8-byte stores (lane: row m, 4 consecutive columns)
#pragma unroll
    for (int nt = 0; nt < 6; nt++) {
        const int n = n0 + wn * 96 + nt * 16 + 4 * q;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(P.bias + n);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const int m = m0 + wm * 64 + mt * 16 + r16;
            const f32x4 v = acc[nt][mt];
            const uint2 pk = make_uint2(pack_bf16x2(v[0] + b4[0], v[1] + b4[1]), pack_bf16x2(v[2] + b4[2], v[3] + b4[3]));
            if (m < M) *reinterpret_cast<uint2 *>(P.out + (size_t)m * P.N + n) = pk;
        }
    }
}


// ------------------------------------------------------------------------------------------
// k_policy_fc1w (round 4): the same GEMM with ONE wave per SIMD and its whole body - operand DMA, K loop, epilogue - as one
// asm statement generated and checked by tools/gen_policy_fc1w.py (the design and its reasons are in that file's header and in
// DESIGN.md section 5): 256-thread workgroup, wave tile 96 columns x 128 rows (6 x 8 MFMA tiles, accumulators on a[0:191]),
// K-stages of 64 as whole 128-B cache lines through two ring slots, one barrier per 96 MFMAs, fragments double-buffered in
// registers, two instructions per DMA piece, 16-byte output stores.  Same fp32 chain per output element as k_policy_fc (K
// ascending in 32-wide MFMA steps, weights as the A operand, bias added last): the two kernels agree to the bit, for every K
// that is a multiple of 64.
// ------------------------------------------------------------------------------------------
#include "xq_policy_fc1w_body.inc"
#if XQ_TOWER_PROBES
#include "xq_policy_fc1w_abl.inc"      // timing-only bodies (wrong results): python tools/gen_policy_fc1w.py --ablations
#endif
#define XQ_FV8(b) "v" #b "0", "v" #b "1", "v" #b "2", "v" #b "3", "v" #b "4", "v" #b "5", "v" #b "6", "v" #b "7", "v" #b "8", "v" #b "9"
#define XQ_FA8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define XQ_FS8(b) "s" #b "0", "s" #b "1", "s" #b "2", "s" #b "3", "s" #b "4", "s" #b "5", "s" #b "6", "s" #b "7", "s" #b "8", "s" #b "9"
static_assert(XQ_FC1W_V_LAST == 177 && XQ_FC1W_S_FIRST == 36 && XQ_FC1W_S_LAST == 97, "clobber list of the FC body");

// ABL (probes builds): 2 / 5 = other placements of a stage's DMA pieces (same results), 3 = no operand DMA behind the prologue,
// 4 = no MFMAs, 6 = no MFMAs and no fragment reads (wrong results), 7 / 8 = the product / body 3 with phase stamps
#define XQ_FC1W_ASM(BODY)                                                                                                          \
    asm volatile(BODY                                                                                                              \
                 :                                                                                                                 \
                 : "s"(act_t), "s"(act_bytes), "s"(w_t), "s"(w_bytes), "s"(out_t), "s"(wave), "s"(K2), "s"(nst), "s"(N2), "s"(rows), \
                   "s"(bias_t), "v"(vfa), "v"(vfb), "v"(vdma), "v"(vstore), "v"(vrow), "v"(vbias), "s"(stamp_p), "v"(vdmaw)        \
                 : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", XQ_FV8(1), XQ_FV8(2), XQ_FV8(3), XQ_FV8(4), XQ_FV8(5), XQ_FV8(6), \
                   XQ_FV8(7), XQ_FV8(8), XQ_FV8(9), XQ_FV8(10), XQ_FV8(11), XQ_FV8(12), XQ_FV8(13), XQ_FV8(14), XQ_FV8(15), XQ_FV8(16), \
                   "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177",                                                 \
                   "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", XQ_FA8(1), XQ_FA8(2), XQ_FA8(3), XQ_FA8(4), XQ_FA8(5), XQ_FA8(6), \
                   XQ_FA8(7), XQ_FA8(8), XQ_FA8(9), XQ_FA8(10), XQ_FA8(11), XQ_FA8(12), XQ_FA8(13), XQ_FA8(14), XQ_FA8(15), XQ_FA8(16), \
                   XQ_FA8(17), XQ_FA8(18), "a190", "a191",                                                                         \
                   "s36", "s37", "s38", "s39", XQ_FS8(4), XQ_FS8(5), XQ_FS8(6), XQ_FS8(7), XQ_FS8(8), "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "vcc", "scc", "memory")

template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_policy_fc1w(FcArgs P)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    // XCD-aware, bijective blockIdx -> tile, as in k_policy_fc
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
    const int t = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
    const int rb = t / P.ncol, cb = t - rb * P.ncol;
    const int m0 = rb * BM, n0 = cb * BN;
    int M = P.M;
    if (P.n_rows) { const int n = *P.n_rows; M = n < M ? n : M; }
    if (m0 >= M) return;                                              // compaction: no row in this tile
    const int rows = M - m0 < BM ? M - m0 : BM;                       // rows past M read as zeros (buffer bounds), are not stored
    const int K2 = P.K * 2, N2 = P.N * 2, nst = P.K >> 6;
    const uint16_t *act_t = P.act + (size_t)m0 * P.K;
    const uint16_t *w_t = P.w + (size_t)n0 * P.K;
    uint16_t *out_t = P.out + (size_t)m0 * P.N + n0;
    const float *bias_t = P.bias + n0;
    const int act_bytes = rows * K2, w_bytes = BN * K2;
    // LDS image of a stage: 256 activation rows, then (at 32,768) 192 weight rows, 128 B (64 bf16) each; chunk c of a row sits
    // at chunk c ^ g.  Activations: g = (row >> 1) & 7 - the 16 lanes one ds_read_b128 cycle serves (lanes {0-3, 12-15, 20-27},
    // ...) then cover all 16 (row & 1, chunk) positions of the 256-B bank row.  Weights: MFMA row i of tile at reads weight row
    // 32 (at >> 1) + 4 (at & 1) + 8 (i >> 2) + (i & 3) (a lane then holds 8 consecutive output columns of a tile pair) and g is
    // the same function of i - for weight row n that is 2 ((n >> 3) & 3) + ((n >> 1) & 1).  The DMA lands lane l at byte 16 l of
    // its piece (8 rows x 128 B, piece p = wave + 4 j), so the swizzle is in what each lane fetches.
    const int gr = (r16 >> 1) & 7;
    const int vfa = (wn * 96 + 8 * (r16 >> 2) + (r16 & 3)) * 128 + ((q ^ gr) << 4);   // weight fragment (K half 0, ring slot 0; + 32,768 in the body)
    const int vfb = (wm * 128 + r16) * 128 + ((q ^ gr) << 4);                                    // activation fragment
    const int ga = 4 * (wave & 1) + ((lane >> 4) & 3), gw = 2 * wave + ((lane >> 4) & 1);
    const int vdma = (lane >> 3) * K2 + (((lane & 7) ^ ga) << 4);
    const int vdmaw = (lane >> 3) * K2 + (((lane & 7) ^ gw) << 4);
    const int vstore = ((wm * 128 + r16) * P.N + wn * 96 + 8 * q) * 2;
    const int vrow = wm * 128 + r16;
    const int vbias = (wn * 96 + 8 * q) * 4;
    unsigned long long *stamp_p = ABL >= 7 ? P.stamps + (size_t)blockIdx.x * 32 : nullptr;
#if XQ_TOWER_PROBES
    if constexpr (ABL == 7) { XQ_FC1W_ASM(XQ_FC1W_BODY_STAMPED); return; }
    if constexpr (ABL == 8) { XQ_FC1W_ASM(XQ_FC1W_BODY_NODMA_STAMPED); return; }
    if constexpr (ABL == 2) { XQ_FC1W_ASM(XQ_FC1W_BODY_FRONT); return; }
    if constexpr (ABL == 3) { XQ_FC1W_ASM(XQ_FC1W_BODY_NODMA); return; }
    if constexpr (ABL == 4) { XQ_FC1W_ASM(XQ_FC1W_BODY_NOMFMA); return; }
    if constexpr (ABL == 5) { XQ_FC1W_ASM(XQ_FC1W_BODY_EVEN); return; }
    if constexpr (ABL == 6) { XQ_FC1W_ASM(XQ_FC1W_BODY_NOMFMA_NOLDS); return; }
#endif
    XQ_FC1W_ASM(XQ_FC1W_BODY);
}

// ------------------------------------------------------------------------------------------
// Value head behind the 1x1 convolution (neural_network.py:43-45,66-69): v = tanh(fc2(relu(fc1(hv)))).
// 184 KB of fc1 weights, 92 kMAC per row: one wave per 16 rows, fragments straight from global memory
// (L2-resident weights, each activation byte read once), 8 x 23 MFMAs, the 128 hidden units stay in the
// accumulators (fp32, no bf16 round trip), fc2 is a 32-term dot product per lane + 2 shuffles.
// K = 720 is padded to 736 = 23 x 32 in the WEIGHTS (zero columns); the activation operand of those 16
// columns is whatever follows the row (finite bf16: the next row, or the caller's 32-byte slack after the
// last row) and meets a zero weight.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_value_head(const uint16_t *__restrict__ hv, const uint16_t *__restrict__ w1,
                                                    const float *__restrict__ b1, const float *__restrict__ w2,
                                                    const float *__restrict__ b2, uint16_t *__restrict__ out, int M,
                                                    const int32_t *__restrict__ n_rows)
{
    constexpr int K = 720, KP = 736;
    if (n_rows) { const int n = *n_rows; M = n < M ? n : M; }        // evaluator row compaction
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (row0 >= M) return;
    const int row = row0 + r16 < M ? row0 + r16 : M - 1;
    const uint16_t *xr = hv + (size_t)row * K + q * 8;
    const uint16_t *wr = w1 + (size_t)r16 * KP + q * 8;
    f32x4 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; nt++) acc[nt] = *reinterpret_cast<const f32x4 *>(b1 + nt * 16 + 4 * q);
    // (one K-step at a time on purpose: unrolled - hipcc turns `#pragma unroll 4` into a rolling pipeline of ~10 loads in
    // flight over all 23 steps - the kernel takes 73 us instead of 20 for 16,384 rows, measured in round 3)
#pragma unroll 1
    for (int ks = 0; ks < KP / 32; ks++) {
        const bf16x8 x = *reinterpret_cast<const bf16x8 *>(xr + ks * 32);
#pragma unroll
        for (int nt = 0; nt < 8; nt++) {
            const bf16x8 w = *reinterpret_cast<const bf16x8 *>(wr + (size_t)nt * 16 * KP + ks * 32);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc[nt], 0, 0, 0);     // D[hidden unit][row]
        }
    }
    float part = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; nt++) {
        const f32x4 c = *reinterpret_cast<const f32x4 *>(w2 + nt * 16 + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; i++) part += fmaxf(acc[nt][i], 0.f) * c[i];
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (q == 0 && row0 + r16 < M) {
        const float v = tanhf(part + b2[0]);
        out[row0 + r16] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
    }
}

}  // namespace

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: remembered per device ordinal (a process may drive
// several GPUs, xq_config.device)
// -1 / 1 = k_policy_fc1w (one wave per SIMD, generated asm body: the product since round 5), 0 = k_policy_fc (8 waves, HIP).
// Same bits.  k_policy_fc1w takes 152-156 us where k_policy_fc takes 175-200 (16,384 rows, alternating in one process).  Round 4
// kept the slower kernel as the default: on one box the self-play step was 0.6 % LONGER with the faster one (the trunk kernel
// around it held a lower clock), on another +0.6 / -1.0 %.  Round 5 repeated the alternation (5 pairs of 5 steps) on a fast box
// (trunk at 2.16 GHz) and on a slow one (2.06 GHz): +0.95 % and +0.85 % for k_policy_fc1w, every pair, spread +- 0.1 %
// (profiles/r05_ab_policy_fc.txt, r05c_ab_policy_fc.txt) - the kernel's own saving arrives in full.
static int g_fc_variant = -1;
// diagnostic switch (include/xq_debug.h): both kernels compute the same bits
extern "C" void xq_policy_fc_set_variant(int v) { g_fc_variant = v; }

static int fc_lds_opt_in()
{
    static std::atomic<uint64_t> done{ 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return XQ_E_HIP;
    if (done.load(std::memory_order_acquire) >> dev & 1) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            FC_LDS_BYTES) != hipSuccess)
        return XQ_E_HIP;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            XQ_FC1W_LDS_BYTES) != hipSuccess)
        return XQ_E_HIP;
#if XQ_TOWER_PROBES
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<2>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<3>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<4>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<5>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<6>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<7>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc1w<8>), hipFuncAttributeMaxDynamicSharedMemorySize, XQ_FC1W_LDS_BYTES) != hipSuccess)
        return XQ_E_HIP;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc<1>), hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_fc<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS_BYTES) != hipSuccess)
        return XQ_E_HIP;
#endif
    done.fetch_or(1ull << dev, std::memory_order_release);
    return 0;
}

/* logits = act . w^T + bias on the engine's stream.  K % 64 == 0, N % 192 == 0 (pad the weight rows).
 * n_rows_dev (optional): device int32, only rows below it are computed (evaluator row compaction). */
extern "C" int xq_policy_fc_bf16(void *stream, const void *act, const void *w, const void *bias, void *out, int M, int N, int K,
                                 const void *n_rows_dev)
{
    if (!act || !w || !bias || !out || M <= 0 || N <= 0 || K < 64 || (K & 63) || N % BN) return XQ_E_INVALID;
    if (int rc = fc_lds_opt_in()) return rc;
    FcArgs a{ (const uint16_t *)act, (const uint16_t *)w, (const float *)bias, (uint16_t *)out, M, N, K, N / BN,
              (const int32_t *)n_rows_dev, nullptr };
    const int ntiles = ((M + BM - 1) / BM) * (N / BN);
#if XQ_TOWER_PROBES
    if (g_fc_variant >= 2 && g_fc_variant <= 6) {
        const hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
        if (g_fc_variant == 2) hipLaunchKernelGGL(k_policy_fc1w<2>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
        if (g_fc_variant == 3) hipLaunchKernelGGL(k_policy_fc1w<3>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
        if (g_fc_variant == 4) hipLaunchKernelGGL(k_policy_fc1w<4>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
        if (g_fc_variant == 5) hipLaunchKernelGGL(k_policy_fc1w<5>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
        if (g_fc_variant == 6) hipLaunchKernelGGL(k_policy_fc1w<6>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
        return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
    }
#endif
    if (g_fc_variant > 1) return XQ_E_INVALID;
    if (g_fc_variant != 0)
        hipLaunchKernelGGL(k_policy_fc1w<0>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(k_policy_fc<0>, dim3(ntiles), dim3(512), FC_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

// diagnostic (include/xq_debug.h): the same launch with a timing-only body (wrong results); XQ_E_INVALID in a product library
extern "C" int xq_policy_fc_debug(int ablate, void *stream, const void *act, const void *w, const void *bias, void *out, int M, int N,
                                  int K)
{
#if XQ_TOWER_PROBES
    if (!act || !w || !bias || !out || M <= 0 || N <= 0 || K < 64 || (K & 63) || N % BN || ablate < 1 || ablate > 2) return XQ_E_INVALID;
    if (int rc = fc_lds_opt_in()) return rc;
    FcArgs a{ (const uint16_t *)act, (const uint16_t *)w, (const float *)bias, (uint16_t *)out, M, N, K, N / BN, nullptr, nullptr };
    const int ntiles = ((M + BM - 1) / BM) * (N / BN);
    if (ablate == 1) hipLaunchKernelGGL(k_policy_fc<1>, dim3(ntiles), dim3(512), FC_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(k_policy_fc<2>, dim3(ntiles), dim3(512), FC_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
#else
    return XQ_E_INVALID;
#endif
}

/* k_policy_fc1w with phase stamps (probes library; include/xq_debug.h) */
extern "C" int xq_policy_fc_debug_stamps(int nodma, void *stream, const void *act, const void *w, const void *bias, void *out, int M, int N,
                                         int K, void *stamps_dev)
{
#if XQ_TOWER_PROBES
    if (!act || !w || !bias || !out || !stamps_dev || M <= 0 || N <= 0 || K < 64 || (K & 63) || N % BN) return XQ_E_INVALID;
    if (int rc = fc_lds_opt_in()) return rc;
    FcArgs a{ (const uint16_t *)act, (const uint16_t *)w, (const float *)bias, (uint16_t *)out, M, N, K, N / BN, nullptr,
              (unsigned long long *)stamps_dev };
    const int ntiles = ((M + BM - 1) / BM) * (N / BN);
    const hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    if (nodma) hipLaunchKernelGGL(k_policy_fc1w<8>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
    else hipLaunchKernelGGL(k_policy_fc1w<7>, dim3(ntiles), dim3(256), XQ_FC1W_LDS_BYTES, hs, a);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
#else
    return XQ_E_INVALID;
#endif
}

/* values[m] = tanh(w2 . relu(w1 . hv[m] + b1) + b2): hv [n_rows][720] bf16 with >= 32 readable bytes behind the last
 * row, w1 [128][736] bf16 (columns 720.. zero), b1 / w2 float32[128], b2 float32[1], values bf16[n_rows].
 * n_rows_dev (optional): device int32, only rows below it are computed. */
extern "C" int xq_value_head_bf16(void *stream, const void *hv, const void *w1, const void *b1, const void *w2, const void *b2,
                                  void *values, int n_rows, const void *n_rows_dev)
{
    if (!hv || !w1 || !b1 || !w2 || !b2 || !values || n_rows <= 0) return XQ_E_INVALID;
    hipLaunchKernelGGL(k_value_head, dim3((n_rows + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const uint16_t *)hv, (const uint16_t *)w1, (const float *)b1, (const float *)w2, (const float *)b2,
                       (uint16_t *)values, n_rows, (const int32_t *)n_rows_dev);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

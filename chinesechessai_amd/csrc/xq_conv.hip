// xq_conv.hip — fused 3x3 convolution for the policy/value ResNet on gfx950 (CDNA4 MFMA).
//
//   y = relu( conv3x3(x, w) + bias [+ residual] )        x, y, residual: NHWC bf16 [G][10][9][C]
//
// replaces, per residual-tower layer, MIOpen's igemm + a bias kernel + a ReLU kernel + a residual
// add kernel + a layout copy (38.8 % of the round-1 step was those elementwise passes:
// profiles/r01_bench_c3_kernel_stats.csv).  Reference op: neural_network.py:54,181-187 with
// eval-mode BatchNorm folded into (w, bias).
//
// Mapping (implicit GEMM, D = W · Xᵀ so that the accumulator holds 4 consecutive output channels
// per lane and the epilogue packs them without cross-lane traffic): see k_conv3x3_b below (2 boards x
// 2 output-channel halves per 256-thread workgroup, 2 workgroups per CU).  Since csrc/xq_tower.hip runs
// the whole trunk in one launch these per-layer kernels are the selectable fallback and the pinned
// comparison of the trunk kernel.
#include <atomic>
#include "../../include/xq_selfplay.h"
#include "../../include/xq_debug.h"
#include "xq_mfma.hpp"

namespace {
using namespace xqm;

constexpr int PIX = 90;
constexpr int COUT = 128;

// ------------------------------------------------------------------------------------------
// Variant B: 2 boards per workgroup, 2 workgroups per CU (8 waves, two per SIMD) so that one
// workgroup's load / store phases and barriers hide behind the other's MFMA work.
//   wave (b, hc): board b, output-channel half hc -> 3 pixel tiles x 2 channel tiles (96 acc regs)
//   weights streamed in K-slices of KSL input channels: [128 cout][KSL] double-buffered (32 KB)
//   LDS per workgroup: 2 x 23,040 (boards / output staging) + 32,768 + 256 = 79,104 B
// ------------------------------------------------------------------------------------------
// ABLATE (diagnostic builds only, results are wrong): 1 = no stage barriers / weight DMA in the main loop
// NB = boards per workgroup (2 waves per board), KSLP = input channels per weight stage.
//   variant B: NB 2, KSLP 64  -> 256 threads, 79.6 KB LDS, 2 workgroups per CU, 18 stages
//   variant C: NB 4, KSLP 128 -> 512 threads, 158.5 KB LDS, 1 workgroup per CU, 9 stages (half the
//              weight re-streaming and half the barriers / DMA pieces of B)

template <int CIN, int NB, int KSLP, bool STAMP = false, int ABLATE = 0>
__global__ __launch_bounds__(NB * 128, 2) void k_conv3x3_b(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w,
                                                      const float *__restrict__ bias, const uint16_t *__restrict__ res,
                                                      uint16_t *__restrict__ y, int G, int relu,
                                                      unsigned long long *stamps = nullptr)
{
    constexpr int NT = NB * 128;                         // threads per workgroup
    constexpr int ROWB = CIN * 2, NCH = CIN / 8;
    constexpr int ARP = 256 / ROWB >= 1 ? (ROWB >= 256 ? 1 : 256 / ROWB) : 1;     // act rows per 256 B
    constexpr int KSL = CIN >= KSLP ? KSLP : CIN;        // input channels per weight stage
    constexpr int SPT = CIN / KSL;                       // stages per tap
    constexpr int NSTAGE = 9 * SPT;
    constexpr int WROWB = KSL * 2, WNCH = KSL / 8;       // weight-slice row bytes / chunks
    constexpr int WRP = 256 / WROWB;                     // weight rows per 256 B
    constexpr int KSTEPS = KSL / 16;
    constexpr int ACT_BYTES = PIX * 256;
    constexpr int WBUF_BYTES = COUT * WROWB;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *act = lds;
    uint8_t *wbuf = lds + NB * ACT_BYTES;
    uint8_t *zrow = wbuf + 2 * WBUF_BYTES;
    float *lbias = reinterpret_cast<float *>(zrow + 256);     // 128 floats

    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 32 + slot] = t;
            if (slot == 0 || slot == 23) {
                unsigned long long rt = __builtin_amdgcn_s_memrealtime();
                if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 32 + (slot == 0 ? 24 : 25)] = rt;
            }
        }
    };
    stamp(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases stay in SGPRs
    const int wb_ = wave >> 1, hc = wave & 1;            // board within the workgroup, channel half
    const int board = blockIdx.x * NB + wb_;
    const bool board_ok = board < G;
    uint8_t *my_act = act + wb_ * ACT_BYTES;

    auto aswz = [&](int row) { return (row / ARP) & (NCH - 1); };
    auto wswz = [&](int row) { return (row / WRP) & (WNCH - 1); };

    constexpr int WINSTR = (COUT * WNCH + NT - 1) / NT;  // DMA instructions per wave per stage
    constexpr bool WEXACT = (WINSTR * NT == COUT * WNCH);
    // per-lane source offsets of the weight DMA pieces are loop-invariant: a piece is then
    // "uniform stage base + 32-bit lane offset", no 64-bit address arithmetic per piece
    int wsrc[WINSTR];
#pragma unroll
    for (int j = 0; j < WINSTR; j++) {
        const int q = (wave * WINSTR + j) * 64 + lane, row = q / WNCH, cp = q % WNCH;
        wsrc[j] = row * ROWB + ((cp ^ wswz(row)) * 16);
    }
    const rsrc_t wrsrc = make_rsrc(w, 9 * COUT * CIN * 2);
    auto stage_weights = [&](int st, int buf) {
        const int tap = st / SPT, kb = (st % SPT) * KSL;
        const int soff = (tap * COUT * CIN + kb) * 2;                    // scalar stage base
#pragma unroll
        for (int j = 0; j < WINSTR; j++) {
            const int q0 = (wave * WINSTR + j) * 64;
            if (WEXACT || q0 + lane < COUT * WNCH)
                dma16_buf(wrsrc, wsrc[j], soff, wbuf + buf * WBUF_BYTES + q0 * 16);
        }
    };
    if (tid < 16) reinterpret_cast<uint4 *>(zrow)[tid] = make_uint4(0, 0, 0, 0);
    if (tid >= 64 && tid < 96) reinterpret_cast<f32x4 *>(lbias)[tid - 64] = reinterpret_cast<const f32x4 *>(bias)[tid - 64];
    stage_weights(0, 0);
    if (board_ok) {
        // the two waves of a board each fetch half of it
        const uint8_t *src = reinterpret_cast<const uint8_t *>(x) + (size_t)board * PIX * ROWB;
        constexpr int NCHUNK = PIX * NCH;
        constexpr int NI = (NCHUNK + 127) / 128;
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int q0 = (j * 2 + hc) * 64, q = q0 + lane, p = q / NCH, cp = q % NCH;
            if (q < NCHUNK) dma16(src + p * ROWB + ((cp ^ aswz(p)) * 16), my_act + q0 * 16);
        }
    }

    const int r32 = lane & 31, h = lane >> 5;
    int opix[3];
    uint32_t vmask[3];
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

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mt][nt][i] = 0.f;

    barrier_dma();
    stamp(1);

    // residual rows for the epilogue: requested a few stages before the end of the main loop so
    // that their HBM latency is covered by MFMA work
    constexpr int NO = (PIX * 16 + 127) / 128;              // 12 row-chunks per lane
    uint4 rres[NO];
    const uint4 *rsrc = (res && board_ok) ? reinterpret_cast<const uint4 *>(res + (size_t)board * PIX * COUT) : nullptr;
    constexpr int RES_STAGE = NSTAGE > 4 ? NSTAGE - 4 : 0;

    // per-lane byte offsets of the weight fragments inside a stage buffer: fixed for the whole kernel
    int aoff[2][KSTEPS];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; kk++) {
            const int row = hc * 64 + mt * 32 + r32, cw = kk * 2 + h;
            aoff[mt][kk] = row * WROWB + ((cw ^ wswz(row)) * 16);
        }
    const int act_off = (int)(my_act - lds), zrow_off = (int)(zrow - lds), wbuf_off = (int)(wbuf - lds);

    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, off = dy * 9 + dx;
        // row address with the swizzle folded in: chunk c of the row lives at a0 ^ (c << 4)
        // (rows are ROWB-aligned, so the XOR only touches the chunk bits); padding -> zero row
        int a0[3];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const bool ok = (vmask[nt] >> tap) & 1u;
            const int sp = opix[nt] + off;
            a0[nt] = ok ? act_off + sp * ROWB + ((aswz(sp) ^ h) << 4) : zrow_off + (h << 4);
        }
#pragma unroll
        for (int sl = 0; sl < SPT; sl++) {
            const int st = tap * SPT + sl;
            const int buf = (SPT % 2 == 0) ? (sl & 1) : (st & 1);   // compile-time when SPT is even
            // next stage's weight slice by LDS-DMA into the other buffer.  (Register staging — plain
            // dwordx4 loads + ds_write_b128 at the end of the stage — was measured 1.55x SLOWER here:
            // the wave stalls on the load latency before the barrier; tools/bench_conv.py.)
            if (ABLATE != 1 && ABLATE != 2 && st + 1 < NSTAGE) stage_weights(st + 1, buf ^ 1);
            if (st == RES_STAGE && rsrc) {
#pragma unroll
                for (int j = 0; j < NO; j++) {
                    const int i = (j * 2 + hc) * 64 + lane;
                    rres[j] = (i < PIX * 16) ? rsrc[i] : make_uint4(0, 0, 0, 0);
                }
            }
            const int wb_off = wbuf_off + buf * WBUF_BYTES;
            auto load_frags = [&](int kk, bf16x8 (&bf)[3], bf16x8 (&af)[2]) {
                const int cconst = (sl * (KSL / 8) + kk * 2) << 4;      // compile-time
#pragma unroll
                for (int nt = 0; nt < 3; nt++)
                    bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + (a0[nt] ^ cconst));
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
                    af[mt] = *reinterpret_cast<const bf16x8 *>(lds + wb_off + aoff[mt][kk]);
            };
            bf16x8 bfr[2][3], afr[2][2];
            load_frags(0, bfr[0], afr[0]);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; kk++) {
                const int cur = kk & 1;
                if (kk + 1 < KSTEPS) load_frags(kk + 1, bfr[cur ^ 1], afr[cur ^ 1]);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int nt = 0; nt < 3; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[cur][mt], bfr[cur][nt], acc[mt][nt], 0, 0, 0);
            }
            if (ABLATE != 1 && ABLATE != 3) barrier_dma();
            if constexpr (STAMP) { if (st < 18) stamp(2 + st); }
        }
    }

    stamp(20);
    // ---- epilogue: +bias -> bf16 -> the board's region [pixel][128 cout] (both channel halves) ----
    const bool early_relu = relu && !res;       // without a residual the ReLU is applied here and
                                                // the way out is a plain copy
    // staging address of pixel p, channel chunk c: row base | ((c ^ (p & 15)) << 4) | 8h, i.e.
    // sbase ^ (c_const << 4) with the per-lane part folded into sbase
    int sbase[3];
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
        const int p = opix[nt] < PIX ? opix[nt] : 0;
        sbase[nt] = (act_off + p * 256 + 8 * h + ((p & 15) << 4)) ^ (hc << 7);
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c0 = hc * 64 + mt * 32 + 8 * q + 4 * h;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(lbias + c0);
#pragma unroll
            for (int nt = 0; nt < 3; nt++) {
                if (nt < 2 || opix[nt] < PIX) {                 // tiles 0 and 1 are always real pixels
                    const float v0 = acc[mt][nt][4 * q + 0] + b4[0], v1 = acc[mt][nt][4 * q + 1] + b4[1];
                    const float v2 = acc[mt][nt][4 * q + 2] + b4[2], v3 = acc[mt][nt][4 * q + 3] + b4[3];
                    uint32_t w0 = pack_bf16x2(v0, v1), w1 = pack_bf16x2(v2, v3);
                    if (early_relu) { w0 = relu_bf16x2(w0); w1 = relu_bf16x2(w1); }
                    *reinterpret_cast<uint2 *>(lds + (sbase[nt] ^ ((mt * 4 + q) << 4))) = make_uint2(w0, w1);
                }
            }
        }
    }
    stamp(21);
    barrier_dma();
    stamp(22);
    if (board_ok) {
        // each wave streams half of its board's rows out (coalesced 16-B chunks), residual + ReLU
        uint4 *dst = reinterpret_cast<uint4 *>(y + (size_t)board * PIX * COUT);
#pragma unroll
        for (int j = 0; j < NO; j++) {
            const int i = (j * 2 + hc) * 64 + lane;
            if (i < PIX * 16) {
                const int p = i >> 4, c = i & 15;
                uint4 v = *reinterpret_cast<const uint4 *>(my_act + p * 256 + ((c ^ (p & 15)) * 16));
                uint32_t wv[4] = { v.x, v.y, v.z, v.w };
                uint32_t rv[4] = { 0, 0, 0, 0 };
                if (rsrc) { rv[0] = rres[j].x; rv[1] = rres[j].y; rv[2] = rres[j].z; rv[3] = rres[j].w; }
                if (rsrc) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float lo = bf16_lo(wv[k]) + bf16_lo(rv[k]), hi = bf16_hi(wv[k]) + bf16_hi(rv[k]);
                        wv[k] = pack_bf16x2(lo, hi);
                        if (relu) wv[k] = relu_bf16x2(wv[k]);
                    }
                }
                dst[i] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
        }
    }
    stamp(23);
}


// ------------------------------------------------------------------------------------------
// Heads: the policy (128->32) and value (128->8) 1x1 convolutions + folded BN + ReLU
// (neural_network.py:61,66) as ONE pass over the tower output, written directly in the layouts the
// two fully-connected layers consume: P [G][90*32] and V [G][90*8] (h, w, c order; the FC weights
// are permuted once on the host).  Memory-bound: 23 KB in, 7.2 KB out per board.
//   workgroup = 4 waves = 4 boards; a wave DMA's its board into LDS (swizzled) and runs
//   3 pixel tiles x 2 channel tiles x 8 k-steps of v_mfma_f32_32x32x16_bf16 (D = W . X^T);
//   weights [64 rows (40 used)][128] are staged once per workgroup.
// ------------------------------------------------------------------------------------------
template <int HNB>   // boards (= waves) per workgroup
__global__ __launch_bounds__(HNB * 64, 2) void k_heads(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w,
                                                  const float *__restrict__ bias, uint16_t *__restrict__ P,
                                                  uint16_t *__restrict__ V, int G)
{
    constexpr int ACT_BYTES = PIX * 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *wl = lds + HNB * ACT_BYTES;                                  // [64][256 B], swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board = blockIdx.x * HNB + wave;
    const bool board_ok = board < G;
    uint8_t *my_act = lds + wave * ACT_BYTES;
    {   // weights: 64 rows x 16 chunks = 1024 chunks = 16 DMA pieces per workgroup
        const uint8_t *src = reinterpret_cast<const uint8_t *>(w);
#pragma unroll
        for (int j = 0; j < 16 / HNB; j++) {
            const int q0 = (wave * (16 / HNB) + j) * 64, q = q0 + lane, row = q >> 4, cp = q & 15;
            dma16(src + row * 256 + ((cp ^ (row & 15)) * 16), wl + q0 * 16);
        }
    }
    if (board_ok) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(x) + (size_t)board * PIX * 256;
#pragma unroll
        for (int j = 0; j < 23; j++) {
            const int q = j * 64 + lane, p = q >> 4, cp = q & 15;
            if (q < PIX * 16) dma16(src + p * 256 + ((cp ^ (p & 15)) * 16), my_act + j * 1024);
        }
    }
    const int r32 = lane & 31, h = lane >> 5;
    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mt][nt][i] = 0.f;
    barrier_dma();
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        const int c = kk * 2 + h;
        bf16x8 bf[3], af[2];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const int p = nt * 32 + r32;                 // rows >= 90 read neighbouring LDS: discarded
            bf[nt] = *reinterpret_cast<const bf16x8 *>(my_act + p * 256 + ((c ^ (p & 15)) * 16));
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int row = mt * 32 + r32;
            af[mt] = *reinterpret_cast<const bf16x8 *>(wl + row * 256 + ((c ^ (row & 15)) * 16));
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
    if (!board_ok) return;
    // lane holds pixel p and channels mt*32 + 8q + 4h + (0..3): policy = channels 0..31, value = 32..39
    uint8_t *Pb = reinterpret_cast<uint8_t *>(P) + (size_t)board * PIX * 64;
    uint8_t *Vb = reinterpret_cast<uint8_t *>(V) + (size_t)board * PIX * 16;
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
        const int p = nt * 32 + r32;
        if (p < PIX) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = 8 * q + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + c0);
                float v0 = acc[0][nt][4 * q + 0] + b4[0], v1 = acc[0][nt][4 * q + 1] + b4[1];
                float v2 = acc[0][nt][4 * q + 2] + b4[2], v3 = acc[0][nt][4 * q + 3] + b4[3];
                v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
                *reinterpret_cast<uint2 *>(Pb + p * 64 + c0 * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
            {
                const int c0 = 32 + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + c0);
                float v0 = acc[1][nt][0] + b4[0], v1 = acc[1][nt][1] + b4[1];
                float v2 = acc[1][nt][2] + b4[2], v3 = acc[1][nt][3] + b4[3];
                v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
                *reinterpret_cast<uint2 *>(Vb + p * 16 + 4 * h * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
        }
    }
}

}  // namespace

static int g_conv_variant = 1;      // 1 = variant B (2 boards / WG, 2 WG / CU; default: 2 % faster in situ), 2 = variant C (4 boards / WG)

// diagnostic (include/xq_debug.h)
extern "C" void xq_conv3x3_set_variant(int v) { g_conv_variant = v; }

template <int CIN, int NB, int KSLP, bool STAMP, int ABLATE>
static int launch_t(hipStream_t s, const void *x, const void *w, const void *bias, const void *residual, void *y,
                    int n_boards, int relu, void *stamps)
{
    constexpr int KSL = CIN >= KSLP ? KSLP : CIN;
    constexpr int LDS = NB * PIX * 256 + 2 * COUT * KSL * 2 + 256 + 512;
    // (the opt-in is a per-device property: one bit per device ordinal, per kernel instantiation)
    static std::atomic<uint64_t> attr_set{ 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return XQ_E_HIP;
    if (!(attr_set.load(std::memory_order_acquire) >> dev & 1)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_b<CIN, NB, KSLP, STAMP, ABLATE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return XQ_E_HIP;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_conv3x3_b<CIN, NB, KSLP, STAMP, ABLATE>), dim3((n_boards + NB - 1) / NB), dim3(NB * 128), LDS, s,
                       (const uint16_t *)x, (const uint16_t *)w, (const float *)bias, (const uint16_t *)residual,
                       (uint16_t *)y, n_boards, relu, (unsigned long long *)stamps);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

extern "C" int xq_conv3x3_nhwc_bf16(void *stream, const void *x, const void *w, const void *bias, const void *residual,
                                    void *y, int n_boards, int c_in, int relu)
{
    if (!x || !w || !bias || !y || n_boards <= 0 || (c_in != 16 && c_in != 128)) return XQ_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (c_in == 16) return launch_t<16, 2, 64, false, 0>(s, x, w, bias, residual, y, n_boards, relu, nullptr);
    if (g_conv_variant == 1) return launch_t<128, 2, 64, false, 0>(s, x, w, bias, residual, y, n_boards, relu, nullptr);
    return launch_t<128, 4, 128, false, 0>(s, x, w, bias, residual, y, n_boards, relu, nullptr);
}

// diagnostic (include/xq_debug.h): s_memtime phase stamps, 32 u64 per workgroup;
// ablate 1 = no stage barriers / weight DMA, 2 = barriers only, 3 = DMA only (results wrong)
extern "C" int xq_conv3x3_debug_stamps(int variant, int ablate, void *stream, const void *x, const void *w, const void *bias,
                                       const void *residual, void *y, int n_boards, int relu, void *stamps)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (variant == 1) {
        switch (ablate) {
        case 0: return launch_t<128, 2, 64, true, 0>(s, x, w, bias, residual, y, n_boards, relu, stamps);
        case 1: return launch_t<128, 2, 64, true, 1>(s, x, w, bias, residual, y, n_boards, relu, stamps);
        case 2: return launch_t<128, 2, 64, true, 2>(s, x, w, bias, residual, y, n_boards, relu, stamps);
        default: return launch_t<128, 2, 64, true, 3>(s, x, w, bias, residual, y, n_boards, relu, stamps);
        }
    }
    switch (ablate) {
    case 0: return launch_t<128, 4, 128, true, 0>(s, x, w, bias, residual, y, n_boards, relu, stamps);
    case 1: return launch_t<128, 4, 128, true, 1>(s, x, w, bias, residual, y, n_boards, relu, stamps);
    case 2: return launch_t<128, 4, 128, true, 2>(s, x, w, bias, residual, y, n_boards, relu, stamps);
    default: return launch_t<128, 4, 128, true, 3>(s, x, w, bias, residual, y, n_boards, relu, stamps);
    }
}

extern "C" int xq_heads_nhwc_bf16(void *stream, const void *x, const void *w, const void *bias, void *policy_out,
                                  void *value_out, int n_boards)
{
    if (!x || !w || !bias || !policy_out || !value_out || n_boards <= 0) return XQ_E_INVALID;
    constexpr int HNB = 2;                                   // 62 KB LDS -> 2 workgroups per CU
    constexpr int LDS = HNB * PIX * 256 + 64 * 256;
    static std::atomic<uint64_t> attr_set{ 0 };                // one bit per device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return XQ_E_HIP;
    if (!(attr_set.load(std::memory_order_acquire) >> dev & 1)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_heads<HNB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS) != hipSuccess)
            return XQ_E_HIP;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_heads<HNB>, dim3((n_boards + HNB - 1) / HNB), dim3(HNB * 64), LDS, reinterpret_cast<hipStream_t>(stream),
                       (const uint16_t *)x, (const uint16_t *)w, (const float *)bias, (uint16_t *)policy_out,
                       (uint16_t *)value_out, n_boards);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

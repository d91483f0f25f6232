// xq_tower.hip — the whole convolutional trunk of the policy/value network in ONE launch.
//
//   planes [G][10][9][16] bf16  ->  conv3x3(16->128)+ReLU  ->  nblocks x { conv+ReLU, conv+skip+ReLU }
//                               ->  1x1 heads (policy 32, value 8)+ReLU  ->  P [G][90*32], V [G][90*8]
//
// Every layer of this network is board-local (zero padding never crosses a board), so a workgroup
// can carry its boards through all layers without any grid-wide synchronisation: the activations
// stay in LDS from the input planes to the head outputs, the residual stays in registers, and HBM
// sees 2.9 KB in + 7.2 KB out per board instead of 23 KB in/out (+23 KB residual) per layer.
// The per-layer kernels of xq_conv.hip spend 18 % of a workgroup's time in those global
// prologue / epilogue phases (tools/bench_conv.py stamps); here only the weight stream is left.
//
// Same tiling as k_conv3x3_b variant B: workgroup = 4 waves = 2 boards x 2 output-channel halves,
// wave tile 96 pixels x 64 channels (6 accumulators), weights streamed by LDS-DMA in K-slices
// [128 cout][64 cin] through a double buffer, 80.1 KB LDS -> 2 workgroups per CU.
// Reference ops: neural_network.py:54-66,181-187 with eval-mode BatchNorm folded.
#include "../../include/xq_selfplay.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

namespace {

constexpr int PIX = 90, COUT = 128;
constexpr int ACT_BYTES = PIX * 256;          // one board, 128 channels bf16
constexpr int WBUF_BYTES = COUT * 128;        // one weight stage: [128 cout][64 cin] bf16
constexpr int LDS_BYTES = 2 * ACT_BYTES + 2 * WBUF_BYTES + 256 + 2 * 512;

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = { (__bf16)a, (__bf16)b };
    return *reinterpret_cast<uint32_t *>(&v);
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w)      // max(int16, 0) per half
{
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    s16x2 v = *reinterpret_cast<s16x2 *>(&w);
    s16x2 z = { 0, 0 };
    v = __builtin_elementwise_max(v, z);
    return *reinterpret_cast<uint32_t *>(&v);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16_buf(rsrc_t rsrc, int voffset, int soffset, void *lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset,
                                             soffset, 0, 0);
}

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
};

__global__ __launch_bounds__(256, 2) void k_tower(TowerArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *wbuf = lds + 2 * ACT_BYTES;
    uint8_t *zrow = wbuf + 2 * WBUF_BYTES;
    float *lbias = reinterpret_cast<float *>(zrow + 256);            // [2][128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb_ = wave >> 1, hc = wave & 1;                         // board in workgroup, channel half
    const int board = blockIdx.x * 2 + wb_;
    const bool board_ok = board < A.G;
    const int act_off = wb_ * ACT_BYTES, wbuf_off = 2 * ACT_BYTES, zrow_off = wbuf_off + 2 * WBUF_BYTES;
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

    // ---------------------------------------------------------------- input conv (16 -> 128)
    // planes rows are 32 B (2 chunks); swizzle (row / 8) & 1; conv1 tap slices [128][16] = 4 KB
    if (tid < 16) reinterpret_cast<uint4 *>(zrow)[tid] = make_uint4(0, 0, 0, 0);
    if (tid >= 64 && tid < 96) reinterpret_cast<f32x4 *>(lbias)[tid - 64] = reinterpret_cast<const f32x4 *>(A.bias)[tid - 64];
    auto stage_w1 = [&](int tap, int buf) {           // 256 chunks: one piece per wave
        const int q0 = wave * 64, q = q0 + lane, row = q >> 1, cp = q & 1;
        dma16(reinterpret_cast<const uint8_t *>(A.w1) + (size_t)tap * COUT * 32 + row * 32 + ((cp ^ ((row >> 3) & 1)) * 16),
              wbuf + buf * 4096 + q0 * 16);
    };
    stage_w1(0, 0);
    if (board_ok) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(A.planes) + (size_t)board * PIX * 32;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int q0 = (j * 2 + hc) * 64, q = q0 + lane, p = q >> 1, cp = q & 1;
            if (q < PIX * 2) dma16(src + p * 32 + ((cp ^ ((p >> 3) & 1)) * 16), lds + act_off + q0 * 16);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mt][nt][i] = 0.f;
    __syncthreads();
    for (int tap = 0; tap < 9; tap++) {
        const int buf = tap & 1;
        if (tap + 1 < 9) stage_w1(tap + 1, buf ^ 1);
        const int off = (tap / 3 - 1) * 9 + (tap % 3 - 1);
        bf16x8 bf[3], af[2];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const bool ok = (vmask[nt] >> tap) & 1u;
            const int sp = opix[nt] + off;
            const int a = ok ? act_off + sp * 32 + ((((sp >> 3) & 1) ^ h) << 4) : zrow_off + (h << 4);
            bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + a);
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int row = hc * 64 + mt * 32 + r32;
            af[mt] = *reinterpret_cast<const bf16x8 *>(wbuf + buf * 4096 + row * 32 + ((h ^ ((row >> 3) & 1)) * 16));
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        __syncthreads();
    }

    // weight stream of the 128-channel layers: stage = [128 cout][64 cin], 16 pieces per stage
    const int nlayers = 2 * A.nblocks;
    const rsrc_t wrsrc = make_rsrc(A.wt, nlayers * 9 * COUT * COUT * 2);
    int wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = (wave * 4 + j) * 64 + lane, row = q >> 3, cp = q & 7;
        wsrc[j] = row * 256 + ((cp ^ ((row >> 1) & 7)) * 16);
    }
    auto stage_weights = [&](int layer, int st, int buf) {
        const int soff = ((layer * 9 + (st >> 1)) * COUT * COUT + (st & 1) * 64) * 2;
#pragma unroll
        for (int j = 0; j < 4; j++) dma16_buf(wrsrc, wsrc[j], soff, wbuf + buf * WBUF_BYTES + (wave * 4 + j) * 1024);
    };
    int aoff[2][4];                               // weight fragment offsets inside a stage buffer
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int row = hc * 64 + mt * 32 + r32, cw = kk * 2 + h;
            aoff[mt][kk] = row * 128 + ((cw ^ ((row >> 1) & 7)) * 16);
        }

    // epilogue: acc + bias [+ residual] -> ReLU -> bf16 -> LDS rows in place (+ keep as next residual)
    auto epilogue = [&](const float *lb, bool add_res, bool keep_res) {
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
                    if (add_res) {
                        const uint2 r = xres[mt][q][nt];
                        v0 += bf16_lo(r.x); v1 += bf16_hi(r.x); v2 += bf16_lo(r.y); v3 += bf16_hi(r.y);
                    }
                    const uint2 pk = make_uint2(relu_bf16x2(pack_bf16x2(v0, v1)), relu_bf16x2(pack_bf16x2(v2, v3)));
                    if (keep_res) xres[mt][q][nt] = pk;
                    if (nt < 2 || opix[nt] < PIX)
                        *reinterpret_cast<uint2 *>(lds + (sbase[nt] ^ ((mt * 4 + q) << 4))) = pk;
                }
            }
    };

    // conv1 epilogue (all taps were read before the last barrier: rows are rewritten in place)
    if (nlayers > 0) stage_weights(0, 0, 0);
    if (tid >= 64 && tid < 96 && nlayers > 0)
        reinterpret_cast<f32x4 *>(lbias + 128)[tid - 64] = reinterpret_cast<const f32x4 *>(A.bias + 128)[tid - 64];
    epilogue(lbias, false, true);
    __syncthreads();

    // ---------------------------------------------------------------- residual tower
    for (int layer = 0; layer < nlayers; layer++) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[mt][nt][i] = 0.f;
        for (int tap = 0; tap < 9; tap++) {
            const int off = (tap / 3 - 1) * 9 + (tap % 3 - 1);
            int a0[3];
#pragma unroll
            for (int nt = 0; nt < 3; nt++) {
                const bool ok = (vmask[nt] >> tap) & 1u;
                const int sp = opix[nt] + off;
                a0[nt] = ok ? act_off + sp * 256 + (((sp & 15) ^ h) << 4) : zrow_off + (h << 4);
            }
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const int st = tap * 2 + sl;                       // buffer parity == sl (18 stages per layer)
                if (st + 1 < 18) stage_weights(layer, st + 1, sl ^ 1);
                else if (layer + 1 < nlayers) {
                    stage_weights(layer + 1, 0, 0);                // next layer's first slice + its bias
                    if (tid >= 64 && tid < 96)
                        reinterpret_cast<f32x4 *>(lbias + (layer & 1) * 128)[tid - 64] =
                            reinterpret_cast<const f32x4 *>(A.bias + (size_t)(layer + 2) * 128)[tid - 64];
                }
                const int wb_off = wbuf_off + sl * WBUF_BYTES;
                bf16x8 bfr[2][3], afr[2][2];
                auto load_frags = [&](int kk, bf16x8 (&bf)[3], bf16x8 (&af)[2]) {
                    const int cconst = (sl * 8 + kk * 2) << 4;
#pragma unroll
                    for (int nt = 0; nt < 3; nt++) bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + (a0[nt] ^ cconst));
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) af[mt] = *reinterpret_cast<const bf16x8 *>(lds + wb_off + aoff[mt][kk]);
                };
                load_frags(0, bfr[0], afr[0]);
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int cur = kk & 1;
                    if (kk + 1 < 4) load_frags(kk + 1, bfr[cur ^ 1], afr[cur ^ 1]);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int nt = 0; nt < 3; nt++)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[cur][mt], bfr[cur][nt], acc[mt][nt], 0, 0, 0);
                }
                __syncthreads();
            }
        }
        // bias slot of tower layer L is (L + 1) & 1 (slot 0 held conv1's)
        const bool second = layer & 1;                             // second conv of a block: + skip
        epilogue(lbias + ((layer + 1) & 1) * 128, second, second);
        __syncthreads();
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
    __syncthreads();
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
    if (!board_ok) return;
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
}

}  // namespace

extern "C" int xq_tower_nhwc_bf16(void *stream, const void *planes, const void *w1, const void *wt, const void *bias,
                                  const void *wh, const void *bh, void *policy_out, void *value_out, int n_boards,
                                  int n_blocks)
{
    if (!planes || !w1 || !bias || !wh || !bh || !policy_out || !value_out || n_boards <= 0 || n_blocks < 0 ||
        (n_blocks > 0 && !wt) || n_blocks > 64)
        return XQ_E_INVALID;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess)
            return XQ_E_HIP;
        attr_set = true;
    }
    TowerArgs a{ (const uint16_t *)planes, (const uint16_t *)w1, (const uint16_t *)wt, (const float *)bias,
                 (const uint16_t *)wh, (const float *)bh, (uint16_t *)policy_out, (uint16_t *)value_out, n_boards, n_blocks };
    hipLaunchKernelGGL(k_tower, dim3((n_boards + 1) / 2), dim3(256), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? 0 : XQ_E_HIP;
}

// xq_mfma.hpp — what the hand-written MFMA kernels (xq_conv.hip, xq_tower.hip) share: vector types,
// bf16 packing, LDS-DMA (global -> LDS without a VGPR round trip) and the barrier that publishes it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

namespace xqm {

// two floats -> packed bf16 (a in the low half), round-to-nearest-even, NaN stays NaN
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b)
{
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ReLU on two packed bf16: a negative bf16 has its sign bit set, i.e. is a negative int16, so
// max(int16, 0) per half zeroes exactly the negative values (-0.0 becomes +0.0, NaN payloads with
// the sign bit set become 0 — the fp32 path would keep them; activations are finite here).
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w)
{
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    s16x2 v = *reinterpret_cast<s16x2 *>(&w);
    s16x2 z = { 0, 0 };
    v = __builtin_elementwise_max(v, z);
    return *reinterpret_cast<uint32_t *>(&v);
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 16-byte global -> LDS copy (LDS-DMA): the LDS destination is wave-uniform base + lane * 16, the
// global source is per lane, so an XOR swizzle goes on the SOURCE address
// (cdna_hip_programming.md §5.4 rule 21).
__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer resource: address = SGPR resource + per-lane 32-bit voffset + scalar
// soffset, i.e. no per-piece 64-bit VALU address arithmetic.
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), /*stride*/ 0, bytes, /*flags*/ 0x00020000);
}
__device__ __forceinline__ void dma16_buf(rsrc_t rsrc, int voffset, int soffset, void *lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset,
                                             soffset, 0, 0);
}

// A workgroup barrier that also publishes LDS-DMA data: each wave first drains ITS OWN pieces
// (s_waitcnt vmcnt(0)), then the barrier makes every wave's pieces visible.  The wait is explicit:
// the compiler's fence lowering for __syncthreads() does not promise a vmcnt wait at workgroup scope
// (it was missing at one of the stage barriers of k_tower16: a race that showed on a cold device).
__device__ __forceinline__ void barrier_dma()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();
}

// LDS by absolute byte offset (for kernels without static __shared__, whose dynamic allocation starts
// at 0): spares the per-access add of the relocatable base that `extern __shared__` arrays cost
#define XQ_AS3 __attribute__((address_space(3)))
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ bf16x8 lds_ld128(int off) { return *(const XQ_AS3 bf16x8 *)(uint32_t)off; }
__device__ __forceinline__ f32x4 lds_ldf4(int off) { return *(const XQ_AS3 f32x4 *)(uint32_t)off; }
__device__ __forceinline__ u32x2 lds_ld64(int off) { return *(const XQ_AS3 u32x2 *)(uint32_t)off; }
__device__ __forceinline__ u32x4 lds_ld128u(int off) { return *(const XQ_AS3 u32x4 *)(uint32_t)off; }
__device__ __forceinline__ void lds_st64(int off, uint2 v) { *(XQ_AS3 u32x2 *)(uint32_t)off = u32x2{ v.x, v.y }; }
__device__ __forceinline__ void lds_st128(int off, uint4 v) { *(XQ_AS3 u32x4 *)(uint32_t)off = u32x4{ v.x, v.y, v.z, v.w }; }
__device__ __forceinline__ void dma16_abs(const void *gsrc, int lds_off)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (XQ_AS3 void *)(uint32_t)lds_off, 16, 0, 0);
}
__device__ __forceinline__ void dma16_buf_abs(rsrc_t rsrc, int voffset, int soffset, int lds_off)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (XQ_AS3 void *)(uint32_t)lds_off, 16, voffset, soffset, 0, 0);
}
#pragma clang diagnostic pop

}  // namespace xqm

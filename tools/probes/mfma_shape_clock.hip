// Which MFMA shape holds the higher clock under the power limit?  The trunk (k_tower1wa) issues v_mfma_f32_16x16x32_bf16
// and runs at 2.06-2.19 of 2.4 GHz; v_mfma_f32_32x32x16_bf16 does the same flops per cycle with half the instructions and
// half the operand-register reads per flop.  This probe runs the trunk's macro tile (96 x 128 accumulators per wave, one
// wave per SIMD, one workgroup per CU) as a bare issue loop in both shapes, with and without the trunk's LDS operand reads
// (14 x ds_read_b128 per K = 32), for ~0.3 s each, alternating, and prints TFLOP/s and the shader clock held.  At full
// MFMA duty the chip cannot hold its clock: the first line of each pair is the sustained MFMA rate under the power cap.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_clock tools/probes/mfma_shape_clock.hip && /tmp/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#include "mfma_shape_clock_body.inc"

// V: 0 = R16, 1 = R32 (operands stay in registers), 2 = L16, 3 = L32 (the other operand set re-read from LDS under the MFMAs:
// two K = 32 steps per loop pass); 4..8 = L16 with 1, 2, 3, 4, 6 x 16 idle cycles behind every 12 MFMAs (the duty sweep: how the
// sustained rate moves with the matrix pipe's share of the cycles).  The loops are asm statements with fixed registers
// (gen_mfma_shape_clock.py).
template <int V>
__global__ __launch_bounds__(256) void k_probe(const uint4 *src, float *sink, unsigned long long *clk, int passes)
{
    extern __shared__ uint4 lds[];                               // 2 banks x 14 fragments x 256 lanes x 16 B = 114,688 B: one workgroup per CU
    const int t = threadIdx.x;
    for (int i = t; i < 2 * 14 * 256; i += 256) lds[i] = src[i];
    __syncthreads();
    const bool stamp = (blockIdx.x & 63) == 0 && t == 0;
    unsigned long long c0 = 0, r0 = 0;
    if (stamp) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const int a0 = t * 16, a1 = t * 16 + 14 * 4096;
    float o;
    if (V == 0) asm volatile(PROBE_BODY_R16 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 1) asm volatile(PROBE_BODY_R32 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 2) asm volatile(PROBE_BODY_L16 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 3) asm volatile(PROBE_BODY_L32 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 4) asm volatile(PROBE_BODY_L16_I1 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 5) asm volatile(PROBE_BODY_L16_I2 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 6) asm volatile(PROBE_BODY_L16_I3 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 7) asm volatile(PROBE_BODY_L16_I4 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (V == 8) asm volatile(PROBE_BODY_L16_I6 : [o] "=v"(o) : [a0] "v"(a0), [a1] "v"(a1), [n] "s"(passes) : PROBE_CLOBBERS);
    if (stamp) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&clk[0], c1 - c0);
        atomicAdd(&clk[1], r1 - r0);
    }
    sink[blockIdx.x * 256 + t] = o;
}

static uint16_t bf16_of(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); }

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 0.3;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const int n16 = 2 * 14 * 256;
    std::vector<uint16_t> h(n16 * 8);
    uint32_t x = 12345u;
    for (auto &v : h) {                                           // roughly N(0, 1/16): sums of K = 32 stay O(1), nothing overflows
        float s = 0.f;
        for (int k = 0; k < 4; k++) { x = x * 1664525u + 1013904223u; s += (float)(x >> 8) / 16777216.f - 0.5f; }
        v = bf16_of(s * 0.43f);
    }
    uint4 *src; float *sink; unsigned long long *clk;
    const int WG = 256;
    CHECK(hipMalloc(&src, n16 * 16)); CHECK(hipMalloc(&sink, WG * 256 * 4)); CHECK(hipMalloc(&clk, 16));
    CHECK(hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice));
    const size_t ldsb = n16 * 16;
    const int NV = 9;
    void (*ks[NV])(const uint4 *, float *, unsigned long long *, int) = {k_probe<0>, k_probe<1>, k_probe<2>, k_probe<3>, k_probe<4>, k_probe<5>,
                                                                        k_probe<6>, k_probe<7>, k_probe<8>};
    const char *names[NV] = {"16x16x32, registers only", "32x32x16, registers only", "16x16x32 + 14 ds_read_b128 per K=32", "32x32x16 + 14 ds_read_b128 per K=32",
                             "16x16x32 + LDS, 16 idle / 12 MFMAs", "16x16x32 + LDS, 32 idle / 12 MFMAs", "16x16x32 + LDS, 48 idle / 12 MFMAs",
                             "16x16x32 + LDS, 64 idle / 12 MFMAs", "16x16x32 + LDS, 96 idle / 12 MFMAs"};
    for (int c = 0; c < NV; c++) CHECK(hipFuncSetAttribute((const void *)ks[c], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const double flop_it = 1024.0 * 48 * 16384;                  // per loop iteration over the chip: 1,024 waves x 48 MFMAs' worth
    const int iters = (int)(seconds * 2.2e15 / flop_it) & ~1;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("iterations per launch %d (%.1f TFLOP per launch)\n", iters, iters * flop_it / 1e12);
    for (int r = 0; r < reps + 1; r++)
        for (int c = 0; c < NV; c++) {
            CHECK(hipMemset(clk, 0, 16));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(ks[c], dim3(WG), dim3(256), ldsb, 0, src, sink, clk, c >= 2 ? iters / 2 : iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hc[2]; CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
            if (r == 0) continue;                                 // (first pass: warm-up, lets the power controller settle)
            printf("%-40s %8.2f ms  %7.1f TFLOP/s  %.3f GHz  MFMA-pipe share of cycles %.3f\n", names[c], ms, iters * flop_it / ms / 1e9,
                   (double)hc[0] / (double)hc[1] * 0.1, iters * 48.0 * 16.0 / ((double)hc[0] / 4.0));
            fflush(stdout);
        }
    return 0;
}

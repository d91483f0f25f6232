// Is a relaxed agent-scope 64-bit CAS unique across the XCDs of an MI355X?  16,384 one-wave workgroups (dealt round-robin over
// the 8 XCDs) race for the same 4,096 words; exactly one may win each.  Variants of how the expected value is obtained:
// 0 = constant 0 (no load), 1 = plain load first, 2 = agent-scope atomic load first.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/cas_unique tools/probes/cas_unique.hip && /tmp/cas_unique
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(64) void k_race(unsigned long long *tab, int n, int *wins, int variant, unsigned long long tag)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const int pos = (g * 2654435761u >> 7) % n;
    unsigned long long st = 0;
    if (variant == 1) st = tab[pos];
    else if (variant == 2) st = __hip_atomic_load(&tab[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((st >> 32) == tag) return;                       // taken in this round already
    unsigned long long old = 0;
    if (lane == 0) {
        old = atomicCAS(&tab[pos], st, (tag << 32) | (unsigned)g);
        if (old == st) atomicAdd(&wins[pos], 1);
    }
}

int main()
{
    const int n = 4096, G = 16384;
    unsigned long long *tab; int *wins;
    hipMalloc(&tab, n * 8); hipMalloc(&wins, n * 4);
    hipMemset(tab, 0, n * 8);
    std::vector<int> h(n);
    for (int variant = 0; variant < 3; variant++) {
        long bad = 0, total = 0;
        for (int round = 1; round <= 200; round++) {
            hipMemset(wins, 0, n * 4);
            if (variant == 0) hipMemset(tab, 0, n * 8);
            hipLaunchKernelGGL(k_race, dim3(G), dim3(64), 0, 0, tab, n, wins, variant, (unsigned long long)(variant * 1000 + round));
            hipMemcpy(h.data(), wins, n * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; i++) { if (h[i] > 1) bad++; total += h[i]; }
        }
        printf("variant %d: %ld words with more than one winner, %ld wins in 200 rounds\n", variant, bad, total);
    }
    return 0;
}

// the engine's reservation pattern, verbatim: all lanes load, lane 0 CASes, readfirstlane of the result
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ int probe(unsigned long long *tab, unsigned pos, unsigned epoch, unsigned seq)
{
    const int lane = threadIdx.x & 63;
    for (int p = 0; p < 4; p++, pos = (pos + 1) & 4095) {
        const unsigned long long raw = atomicOr(&tab[pos], 0ull);
        const unsigned long long st = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(raw >> 32)) << 32) |
                                      __builtin_amdgcn_readfirstlane((uint32_t)raw);
        const unsigned ep = (unsigned)(st >> 32), filled = (unsigned)st;
        if (epoch - ep > 1u) {
            unsigned long long old = 0;
            if (lane == 0) old = atomicCAS(&tab[pos], st, ((unsigned long long)epoch << 32) | 0xffffffffull);
            old = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(old >> 32)) << 32) |
                  __builtin_amdgcn_readfirstlane((uint32_t)old);
            return old == st ? -(int)pos - 2 : -1;
        }
        if (filled >= seq) return -1;
    }
    return -1;
}
__global__ __launch_bounds__(64) void k(unsigned long long *tab, int *codes, unsigned epoch, unsigned seq)
{
    const int g = blockIdx.x;
    const int ec = probe(tab, 1234u, epoch, seq);
    if ((threadIdx.x & 63) == 0) codes[g] = ec;
}
int main()
{
    const int G = 2048;
    unsigned long long *tab; int *codes;
    hipMalloc(&tab, 4096 * 8); hipMalloc(&codes, G * 4);
    hipMemset(tab, 0, 4096 * 8);
    std::vector<int> h(G);
    for (int round = 0; round < 5; round++) {
        hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, tab, codes, 20u + 2 * round, 1u + round);
        hipMemcpy(h.data(), codes, G * 4, hipMemcpyDeviceToHost);
        int res = 0, none = 0;
        for (int i = 0; i < G; i++) { if (h[i] <= -2) res++; else if (h[i] == -1) none++; }
        printf("round %d: %d waves hold a reservation, %d none\n", round, res, none);
    }
    return 0;
}

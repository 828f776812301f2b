// Streaming-read micro-benchmark: what does a CU sustain with VGPR-destination loads?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
template <int UNROLL, bool NT>
__global__ __launch_bounds__(512) void rd(const u4* __restrict__ p, size_t n_per_block, unsigned* out) {
    const u4* q = p + (size_t)blockIdx.x * n_per_block;
    u4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < n_per_block; i += 512 * UNROLL) {
        u4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + u * 512) : q[i + u * 512];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
int main() {
    const size_t bytes = 2048ull << 20;
    u4* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt)
    for (int blocks : {64, 128, 176, 256, 512, 1024, 2048}) {
        for (size_t mb_per_launch : {16ull, 64ull, 512ull}) {
            size_t per_block = (mb_per_launch << 20) / 16 / blocks / (512 * 8) * (512 * 8);
            if (per_block == 0) continue;
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                // rotate through the 2 GB buffer so data comes from HBM
                size_t off = ((size_t)rep * (mb_per_launch << 20) / 16) % (bytes / 16 - per_block * blocks);
                hipEventRecord(a);
                if (nt) hipLaunchKernelGGL((rd<8, true>), dim3(blocks), dim3(512), 0, 0, d + off, per_block, o);
                else hipLaunchKernelGGL((rd<8, false>), dim3(blocks), dim3(512), 0, 0, d + off, per_block, o);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            double gb = (double)per_block * blocks * 16 / 1e9;
            printf("nt=%d blocks %5d  %4zu MB/launch: %7.1f us  %6.2f TB/s  per-block-CU %6.1f GB/s\n", nt, blocks, mb_per_launch,
                   best * 1e3, gb / best, gb / best * 1e3 / (blocks < 256 ? blocks : 256));
        }
    }
    return 0;
}

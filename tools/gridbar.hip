// Microbenchmark: cost of an in-kernel grid barrier on gfx950 (all workgroups co-resident).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/gridbar tools/gridbar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned SPIN_LIMIT = 1u << 22;

// flat: one counter, monotonically increasing; barrier k completes when counter >= k * nblocks
__device__ __forceinline__ bool bar_flat(unsigned* ctr, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++n > SPIN_LIMIT) { *err = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return true;
}

// hierarchical: 8 per-XCD counters (block b is on XCD b%8), XCD-last arriver bumps the global counter
__device__ __forceinline__ void bar_hier(unsigned* xcd_ctr, unsigned* glob, unsigned k, unsigned per_xcd, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned x = blockIdx.x & 7;
        unsigned old = __hip_atomic_fetch_add(xcd_ctr + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == k * per_xcd) __hip_atomic_fetch_add(glob, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n = 0;
        while (__hip_atomic_load(glob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k * 8) {
            __builtin_amdgcn_s_sleep(1);
            if (++n > SPIN_LIMIT) { *err = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void bar_kernel(unsigned* ctr, int iters, int* err, float* data, int work) {
    unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int k = 1; k <= iters; ++k) {
        if (work) {   // a little dependent traffic: write one value, read a neighbour's after the barrier
            data[(size_t)((k & 1) * nb + blockIdx.x) * 64 + (threadIdx.x & 63)] = acc + k;
        }
        if (MODE == 0) bar_flat(ctr, k * nb, err);
        else bar_hier(ctr + 64, ctr, k, nb / 8, err);
        if (work) acc += data[(size_t)((k & 1) * nb + (blockIdx.x + 17) % nb) * 64 + (threadIdx.x & 63)];
    }
    if (acc == 12345.f) data[0] = acc;
}

int main(int argc, char** argv) {
    int iters = 2000;
    unsigned* ctr; int* err; float* data;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&data, 2 * 1024 * 64 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode)
        for (int work = 0; work < 2; ++work)
            for (int nb : {256, 512}) for (int nt : {256, 512}) {
                CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(err, 0, 4));
                CK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(bar_kernel<0>, nb, nt, 0, 0, ctr, iters, err, data, work);
                else hipLaunchKernelGGL(bar_kernel<1>, nb, nt, 0, 0, ctr, iters, err, data, work);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                printf("mode %s work %d blocks %d threads %d: %.3f us/barrier err %d\n", mode ? "hier" : "flat", work, nb, nt, ms * 1e3 / iters, h);
                fflush(stdout);
            }
    return 0;
}

// glds_probe.hip -- does global_load_lds_dwordx4 reach LDS addresses beyond 64 KB on gfx950 (M0 width), with per-lane
// source addresses (a gather of 16-byte chunks) and a wave-uniform destination?  And what do issue and landing cost?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/glds_probe tools/micro/glds_probe.hip && tools/micro/bin/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256, 1) void k_probe(const uint8_t* __restrict__ src, size_t stride, const int* __restrict__ rows,
                                                 const int* __restrict__ dsts, int ndst, int* __restrict__ bad, unsigned long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint32_t base = lds_addr(smem);
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (wave == 1) {
        t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long c0 = __builtin_readcyclecounter();
        for (int d = 0; d < ndst; d++) {
            // lane l: chunk (l & 31) ^ 5 of row rows[2 d + (l >> 5)] (a swizzled gather) -> LDS dsts[d] + 16 l
            const int row = rows[2 * d + (lane >> 5)];
            const uint8_t* g = src + (size_t)row * stride + 16 * (size_t)((lane & 31) ^ 5);
            glds16(g, base + (uint32_t)__builtin_amdgcn_readfirstlane(dsts[d]));
        }
        unsigned long long c1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long c2 = __builtin_readcyclecounter();
        if (lane == 0) { ticks[0] = c1 - c0; ticks[1] = c2 - c0; }
        (void)t0; (void)t1; (void)t2;
    }
    __syncthreads();
    int nb = 0;
    for (int d = 0; d < ndst; d++) {
        for (int l = tid; l < 64; l += 256) {
            const int row = rows[2 * d + (l >> 5)];
            const uint8_t* g = src + (size_t)row * stride + 16 * (size_t)((l & 31) ^ 5);
            const uint8_t* s = reinterpret_cast<const uint8_t*>(smem) + dsts[d] + 16 * l;
            for (int b = 0; b < 16; b++) nb += s[b] != g[b];
        }
    }
    if (nb) atomicAdd(bad, nb);
}

int main() {
    const size_t stride = 125056;
    const int nrows = 4096;
    std::vector<uint8_t> h(stride * nrows);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((i * 2654435761u) >> 13);
    uint8_t* d; hipMalloc(&d, h.size()); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    const int ndst = 16;
    std::vector<int> dsts(ndst), rows(2 * ndst);
    for (int i = 0; i < ndst; i++) { dsts[i] = i * 10240 + 1024; rows[2 * i] = (i * 977) % nrows; rows[2 * i + 1] = (i * 3301 + 17) % nrows; }
    int *dd, *dr, *dbad; unsigned long long* dt;
    hipMalloc(&dd, ndst * 4); hipMalloc(&dr, 2 * ndst * 4); hipMalloc(&dbad, 4); hipMalloc(&dt, 16);
    hipMemcpy(dd, dsts.data(), ndst * 4, hipMemcpyHostToDevice); hipMemcpy(dr, rows.data(), 2 * ndst * 4, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 160 * 1024, 0, d, stride, dr, dd, ndst, dbad, dt);
        hipError_t e = hipDeviceSynchronize();
        int bad = -1; unsigned long long t[2];
        hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost); hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
        printf("rep %d: %s, mismatching bytes %d (destinations up to %d), issue %llu cycles for %d loads, issue+landed %llu cycles\n",
               rep, hipGetErrorString(e), bad, dsts[ndst - 1] + 1024, t[0], ndst, t[1]);
    }
    return 0;
}

// Micro-benchmark of the sweep kernel's phase A building blocks (one workgroup of 256 threads,
// R = 2 bytes/thread, group of 16 markers, fast layout).  Prints shader cycles per group.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I gmrm_amd/csrc tools/micro/phase_a.hip -o /tmp/phase_a && /tmp/phase_a
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gm_common.h"
using namespace gm;

__device__ __forceinline__ unsigned lo32(double x) { return (unsigned)(unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ unsigned hi32(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ double mk64(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
__device__ __forceinline__ void swap32(double& a, double& b) {
    const auto l = __builtin_amdgcn_permlane32_swap(lo32(a), lo32(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi32(a), hi32(b), false, false);
    a = mk64(l[0], h[0]); b = mk64(l[1], h[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
    const auto l = __builtin_amdgcn_permlane16_swap(lo32(a), lo32(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi32(a), hi32(b), false, false);
    a = mk64(l[0], h[0]); b = mk64(l[1], h[1]);
}
template <int CTRL> __device__ __forceinline__ double dpp64(double x) {
    const int l = __builtin_amdgcn_update_dpp(0, (int)lo32(x), CTRL, 0xf, 0xf, false);
    const int h = __builtin_amdgcn_update_dpp(0, (int)hi32(x), CTRL, 0xf, 0xf, false);
    return mk64((unsigned)l, (unsigned)h);
}
__device__ __forceinline__ double reduce32(double (&acc)[32], int lane) {
#pragma unroll
    for (int i = 0; i < 16; i++) { swap32(acc[i], acc[i + 16]); acc[i] = acc[i] + acc[i + 16]; }
#pragma unroll
    for (int i = 0; i < 8; i++) { swap16(acc[i], acc[i + 8]); acc[i] = acc[i] + acc[i + 8]; }
    { const bool up = (lane & 8) != 0;
#pragma unroll
      for (int i = 0; i < 4; i++) { const double s = up ? acc[i] : acc[i + 4], k = up ? acc[i + 4] : acc[i]; acc[i] = k + dpp64<0x140>(s); } }
    { const bool up = (lane & 4) != 0;
#pragma unroll
      for (int i = 0; i < 2; i++) { const double s = up ? acc[i] : acc[i + 2], k = up ? acc[i + 2] : acc[i]; acc[i] = k + dpp64<0x141>(s); } }
    { const bool up = (lane & 2) != 0; const double s = up ? acc[0] : acc[1], k = up ? acc[1] : acc[0]; acc[0] = k + dpp64<0x1B>(s); }
    return acc[0] + dpp64<0xB1>(acc[0]);
}
__device__ __forceinline__ double code_a_bits(uint32_t w, uint32_t nw, int i) {
    const uint32_t h = (w >> (2 * i + 1)) & 1u;
    const uint32_t nl = (uint32_t)((int)(nw << (31 - 2 * i)) >> 31);
    return mk64(0u, (0x40000000u - (h << 20)) & nl);
}
__device__ __forceinline__ uint32_t blut_row(uint32_t e) { return e ^ ((e >> 3) & 7u) ^ ((e >> 6) & 3u); }

template <int V>
__global__ __launch_bounds__(256, 1) void k(const double* eps, const uint16_t* ringsrc, double* out, long long* cyc, int iters) {
    __shared__ uint16_t ring[64 * 256];
    __shared__ double blut[1024];
    __shared__ double wsum[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 256; i += 256) ring[i] = ringsrc[i];
    for (int i = tid; i < 1024; i += 256) blut[blut_row((uint32_t)i >> 2) * 4 + (i & 3)] = code_a(((i >> 2) >> (2 * (i & 3))) & 3);
    double q1[8], q2[8];
    for (int i = 0; i < 8; i++) split2(eps[tid * 8 + i], q1[i], q2[i]);
    __syncthreads();
    double sink = 0.0;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        double acc[32];
        const int base = (it & 3) * 16;
#pragma unroll
        for (int gm = 0; gm < 16; gm++) {
            const uint32_t wd = ring[(base + gm) * 256 + tid];
            double sa1 = 0.0, sa2 = 0.0;
            if (V == 0) {
                const uint32_t nw = ~wd;
#pragma unroll
                for (int i = 0; i < 8; i++) { const double av = code_a_bits(wd, nw, i); sa1 = fma_(av, q1[i], sa1); sa2 = fma_(av, q2[i], sa2); }
            } else if (V == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const double* row = blut + blut_row((wd >> (8 * kk)) & 0xFFu) * 4;
                    const double2 a01 = *reinterpret_cast<const double2*>(row), a23 = *reinterpret_cast<const double2*>(row + 2);
                    sa1 = fma_(a01.x, q1[4 * kk], sa1); sa2 = fma_(a01.x, q2[4 * kk], sa2);
                    sa1 = fma_(a01.y, q1[4 * kk + 1], sa1); sa2 = fma_(a01.y, q2[4 * kk + 1], sa2);
                    sa1 = fma_(a23.x, q1[4 * kk + 2], sa1); sa2 = fma_(a23.x, q2[4 * kk + 2], sa2);
                    sa1 = fma_(a23.y, q1[4 * kk + 3], sa1); sa2 = fma_(a23.y, q2[4 * kk + 3], sa2);
                }
            } else if (V == 2) {                       // FMAs only (multiplier from the word, no decode)
                const double av = mk64(0u, 0x3FF00000u + (wd & 1u));
#pragma unroll
                for (int i = 0; i < 8; i++) { sa1 = fma_(av, q1[i], sa1); sa2 = fma_(av, q2[i], sa2); }
            } else {                                   // V == 3: no FMAs, reduction only
                sa1 = q1[gm & 7] + (double)wd; sa2 = q2[gm & 7];
            }
            acc[gm * 2] = sa1; acc[gm * 2 + 1] = sa2;
        }
        const double r = reduce32(acc, lane);
        if ((lane & 1) == 0) wsum[wave * 64 + (lane >> 1)] = r;
        sink += r;
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + tid] = sink + wsum[tid & 63];
}

template <int V> void run(const char* name, int blocks) {
    std::vector<double> eps(2048);
    for (int i = 0; i < 2048; i++) eps[i] = 0.37 * ((i * 2654435761u) % 1000) / 500.0 - 0.3;
    std::vector<uint16_t> ring(64 * 256);
    for (size_t i = 0; i < ring.size(); i++) { uint32_t x = (uint32_t)i * 2246822519u; x ^= x >> 13; ring[i] = (uint16_t)(x | 0xAAAA) & 0xFBEF; }
    double *de, *dout; uint16_t* dr; long long* dc;
    hipMalloc(&de, eps.size() * 8); hipMalloc(&dr, ring.size() * 2); hipMalloc(&dout, blocks * 256 * 8); hipMalloc(&dc, blocks * 8);
    hipMemcpy(de, eps.data(), eps.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dr, ring.data(), ring.size() * 2, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, de, dr, dout, dc, iters);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, de, dr, dout, dc, iters);
    hipDeviceSynchronize();
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : c) avg += v; avg /= blocks;
    printf("%-28s blocks %3d: %8.1f s_memtime ticks per group of 16 markers (%.1f per marker)\n", name, blocks, avg / iters, avg / iters / 16);
    hipFree(de); hipFree(dr); hipFree(dout); hipFree(dc);
}
int main() {
    for (int blocks : {1, 245}) {
        run<0>("V0 arithmetic decode", blocks);
        run<1>("V1 byte LUT (LDS)", blocks);
        run<2>("V2 FMAs only", blocks);
        run<3>("V3 reduction only", blocks);
    }
    return 0;
}

// Micro-benchmark of a marker-per-lane phase A: the residual's two exact parts are kept in LDS as
// byte planes of 31-bit integers (Q1 = q1 * 2^22, Q2 = q2 * 2^53), every lane owns one marker of
// the batch and walks its slice bytes: one 4-byte LUT read turns a genotype byte into four 8-bit
// a-values, v_dot4 accumulates them against the eight planes.  No cross-lane reduction, no f64.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/phase_a2.hip -o tools/micro/bin/phase_a2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

constexpr int SB = 512;            // slice bytes per workgroup (R = 2)
constexpr int RING = 256;          // ring positions

// variant: MPL markers per lane, nbp = 64 * MPL / SPLIT... kept simple:
//   V=0: 64 markers, 1 per lane, 4 sub-slices (one per wave)
//   V=1: 128 markers, 2 per lane, 4 sub-slices
//   V=2: 16 markers, 1 per lane, 16 sub-slices
//   V=3: 128 markers, 1 per lane, 2 sub-slices
template <int V>
__global__ __launch_bounds__(256, 1) void k(const uint8_t* ringsrc, const int* q1src, const int* q2src,
                                            long long* out, long long* cyc, int iters) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* ring = smem;                                   // RING * SB, 16-B chunks swizzled by position
    uint32_t* planes = reinterpret_cast<uint32_t*>(smem + RING * SB);        // SB groups * 8 dwords
    uint32_t* lut = planes + SB * 8;                        // 256
    unsigned long long* sall = reinterpret_cast<unsigned long long*>(lut + 256);   // 128 * 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < RING * SB / 16; i += 256) {
        const int pos = i / (SB / 16), c = i % (SB / 16);
        reinterpret_cast<uint4*>(ring)[pos * (SB / 16) + (c ^ (pos & (SB / 16 - 1)))] = reinterpret_cast<const uint4*>(ringsrc)[i];
    }
    for (int g = tid; g < SB; g += 256) {
        uint32_t w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
        for (int j = 0; j < 4; j++) {
            int a = q1src[g * 4 + j], b = q2src[g * 4 + j];
            for (int pl = 0; pl < 4; pl++) {                // signed base-256 digits
                const int da = (int)(int8_t)(a & 0xff), db = (int)(int8_t)(b & 0xff);
                w1[pl] |= ((uint32_t)da & 0xffu) << (8 * j); w2[pl] |= ((uint32_t)db & 0xffu) << (8 * j);
                a = (a - da) >> 8; b = (b - db) >> 8;
            }
        }
        for (int pl = 0; pl < 4; pl++) { planes[g * 8 + pl] = w1[pl]; planes[g * 8 + 4 + pl] = w2[pl]; }
    }
    {
        uint32_t w = 0;
        for (int j = 0; j < 4; j++) { const int c = (tid >> (2 * j)) & 3; w |= (uint32_t)(c == 0 ? 2 : (c == 2 ? 1 : 0)) << (8 * j); }
        lut[tid] = w;
    }
    if (tid < 256) sall[tid] = 0;
    __syncthreads();

    constexpr int NB = V == 0 ? 64 : (V == 2 ? 16 : 128);
    constexpr int MPL = V == 1 ? 2 : 1;
    constexpr int NSUB = 256 * MPL / NB;                    // sub-slices
    constexpr int CH = SB / 16 / NSUB;                      // 16-byte chunks per thread
    const int m0 = tid % (NB / MPL), sub = tid / (NB / MPL);
    long long t0 = clock64(); const long long w0 = wall_clock64();
    long long sink = 0;
    for (int it = 0; it < iters; it++) {
        int acc[MPL][8];
#pragma unroll
        for (int mm = 0; mm < MPL; mm++)
#pragma unroll
            for (int q = 0; q < 8; q++) acc[mm][q] = 0;
        const int pbase = (it * 7) & (RING - 1);
#pragma unroll 2
        for (int c = 0; c < CH; c++) {
            const int chunk = sub * CH + c;
            uint4 w[MPL];
#pragma unroll
            for (int mm = 0; mm < MPL; mm++) {
                const int pos = (pbase + m0 + mm * 64) & (RING - 1);
                w[mm] = *reinterpret_cast<const uint4*>(ring + pos * SB + 16 * (chunk ^ (pos & (SB / 16 - 1))));
            }
            const uint4* pl = reinterpret_cast<const uint4*>(planes + ((size_t)chunk * 16 + (lane & 15)) * 8);
            const uint4 pa = pl[0], pb = pl[1];             // this lane's byte-step of the chunk; broadcast by DPP below
            uint32_t a4[MPL][16];
#pragma unroll
            for (int j = 0; j < 16; j++)
#pragma unroll
                for (int mm = 0; mm < MPL; mm++) {
                    const uint32_t ww = j < 4 ? w[mm].x : (j < 8 ? w[mm].y : (j < 12 ? w[mm].z : w[mm].w));
#ifdef NO_LUT
                    a4[mm][j] = (ww >> (8 * (j & 3))) & 0x03030303u;
#else
                    a4[mm][j] = lut[(ww >> (8 * (j & 3))) & 0xffu];
#endif
                }
            asm volatile("s_nop 1");
#ifdef NO_DPP
#define DPPS(J) ""
#define OPC "v_dot4c_i32_i8_e32"
#else
#define DPPS(J) " row_newbcast:" #J " row_mask:0xf bank_mask:0xf"
#define OPC "v_dot4c_i32_i8_dpp"
#endif
#define STEP(J) _Pragma("unroll") for (int mm = 0; mm < MPL; mm++) \
            asm volatile(OPC " %0, %8, %16" DPPS(J) "\n" \
                         OPC " %1, %9, %16" DPPS(J) "\n" \
                         OPC " %2, %10, %16" DPPS(J) "\n" \
                         OPC " %3, %11, %16" DPPS(J) "\n" \
                         OPC " %4, %12, %16" DPPS(J) "\n" \
                         OPC " %5, %13, %16" DPPS(J) "\n" \
                         OPC " %6, %14, %16" DPPS(J) "\n" \
                         OPC " %7, %15, %16" DPPS(J) \
                         : "+v"(acc[mm][0]), "+v"(acc[mm][1]), "+v"(acc[mm][2]), "+v"(acc[mm][3]), \
                           "+v"(acc[mm][4]), "+v"(acc[mm][5]), "+v"(acc[mm][6]), "+v"(acc[mm][7]) \
                         : "v"(pa.x), "v"(pa.y), "v"(pa.z), "v"(pa.w), "v"(pb.x), "v"(pb.y), "v"(pb.z), "v"(pb.w), "v"(a4[mm][J]));
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
            STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
        }
#pragma unroll
        for (int mm = 0; mm < MPL; mm++) {
            const long long s1 = (long long)acc[mm][0] + ((long long)acc[mm][1] << 8) + ((long long)acc[mm][2] << 16) + ((long long)acc[mm][3] << 24);
            const long long s2 = (long long)acc[mm][4] + ((long long)acc[mm][5] << 8) + ((long long)acc[mm][6] << 16) + ((long long)acc[mm][7] << 24);
            atomicAdd(&sall[(m0 + mm * 64) * 2], (unsigned long long)s1);
            atomicAdd(&sall[(m0 + mm * 64) * 2 + 1], (unsigned long long)s2);
        }
        __syncthreads();
        if (it == 0 && tid < NB * 2) out[(size_t)blockIdx.x * 256 + tid] = (long long)sall[tid];
        if (tid < NB * 2) { sink += (long long)sall[tid]; sall[tid] = 0; }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x + blockIdx.x] = wall_clock64() - w0; if (sink == 0x1234567) out[0] = sink; }
}

template <int V> static void run(const char* name, int nbv) {
    const int blocks = 245, iters = 200;
    std::vector<uint8_t> ring((size_t)RING * SB);
    std::vector<int> q1(SB * 4), q2(SB * 4);
    srand(7);
    for (auto& b : ring) b = (uint8_t)(rand() & 0xff);
    for (auto& x : q1) x = (int)((((long long)rand() << 16) ^ rand()) % (1ll << 30)) * ((rand() & 1) ? 1 : -1);
    for (auto& x : q2) x = (int)((((long long)rand() << 16) ^ rand()) % (1ll << 30)) * ((rand() & 1) ? 1 : -1);
    uint8_t* dr; int *d1, *d2; long long *dout, *dc;
    hipMalloc(&dr, ring.size()); hipMalloc(&d1, q1.size() * 4); hipMalloc(&d2, q2.size() * 4);
    hipMalloc(&dout, (size_t)blocks * 256 * 8); hipMalloc(&dc, blocks * 16);
    hipMemcpy(dr, ring.data(), ring.size(), hipMemcpyHostToDevice);
    hipMemcpy(d1, q1.data(), q1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d2, q2.data(), q2.size() * 4, hipMemcpyHostToDevice);
    const size_t sm = (size_t)RING * SB + SB * 32 + 1024 + 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), sm, 0, dr, d1, d2, dout, dc, iters);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), sm, 0, dr, d1, d2, dout, dc, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> out((size_t)blocks * 256), cyc(blocks * 2);
    hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cyc.data(), dc, cyc.size() * 8, hipMemcpyDeviceToHost);
    // check iteration 0 (pbase = 0): marker m = ring position m
    int bad = 0;
    for (int m = 0; m < nbv; m++) {
        long long s1 = 0, s2 = 0;
        for (int g = 0; g < SB; g++) {
            const uint8_t b = ring[(size_t)m * SB + g];
            for (int j = 0; j < 4; j++) {
                const int c = (b >> (2 * j)) & 3; const int a = c == 0 ? 2 : (c == 2 ? 1 : 0);
                s1 += (long long)a * q1[g * 4 + j]; s2 += (long long)a * q2[g * 4 + j];
            }
        }
        if (out[m * 2] != s1 || out[m * 2 + 1] != s2) bad++;
    }
    double avg = 0, wl = 0; for (int i = 0; i < blocks; i++) { avg += (double)cyc[i]; wl += (double)cyc[blocks + i]; } avg /= blocks; wl /= blocks;
    printf("%-44s markers %3d: %8.0f clk/batch  %6.1f clk/marker  %6.2f us/batch (%s)\n", name, nbv, avg / iters, avg / iters / nbv, wl / iters * 0.01, bad ? "MISMATCH" : "exact");
    hipFree(dr); hipFree(d1); hipFree(d2); hipFree(dout); hipFree(dc);
}

int main() {
    run<0>("64 markers, 1/lane, 4 sub-slices", 64);
    run<1>("128 markers, 2/lane, 4 sub-slices", 128);
    run<2>("16 markers, 1/lane, 16 sub-slices", 16);
    run<3>("128 markers, 1/lane, 2 sub-slices", 128);
    return 0;
}

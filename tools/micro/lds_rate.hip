// LDS read throughput of one CU as phase A uses it: 4 wavefronts, each issuing ds_read_b128 / b64 / b32 back to back from
// conflict-free addresses (lane-contiguous), a counted wait every 8 reads.  Cycles per wave-instruction = what the LDS
// pipeline needs for it when all four SIMDs ask at once.  Also with 9 of every 16 lanes masked off (the unused operand-B columns).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_rate.hip -o tools/micro/bin/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <int W, bool MASK>
__global__ __launch_bounds__(256, 1) void k(long long* cyc, int* sink, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(lane * W + (tid >> 6) * 4096);
    const unsigned long long m = MASK ? 0x007F007F007F007Full : ~0ull;
    v4i a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    int acc = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        unsigned long long sv;
        if (W == 16)
            asm volatile("s_and_saveexec_b64 %4, %6\n\tds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:1024\n\tds_read_b128 %2, %5 offset:2048\n\tds_read_b128 %3, %5 offset:3072\n\t"
                         "ds_read_b128 %0, %5 offset:64\n\tds_read_b128 %1, %5 offset:1088\n\tds_read_b128 %2, %5 offset:2112\n\tds_read_b128 %3, %5 offset:3136\n\ts_mov_b64 exec, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&s"(sv) : "v"(addr), "s"(m) : "memory", "scc");
        else if (W == 8) {
            v2i b0, b1, b2, b3;
            asm volatile("s_and_saveexec_b64 %4, %6\n\tds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:1024\n\tds_read_b64 %2, %5 offset:2048\n\tds_read_b64 %3, %5 offset:3072\n\t"
                         "ds_read_b64 %0, %5 offset:512\n\tds_read_b64 %1, %5 offset:1536\n\tds_read_b64 %2, %5 offset:2560\n\tds_read_b64 %3, %5 offset:3584\n\ts_mov_b64 exec, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&s"(sv) : "v"(addr), "s"(m) : "memory", "scc");
            a0.x += b0.x + b1.x + b2.x + b3.x;
        } else {
            int b0, b1, b2, b3;
            asm volatile("s_and_saveexec_b64 %4, %6\n\tds_read_b32 %0, %5\n\tds_read_b32 %1, %5 offset:1024\n\tds_read_b32 %2, %5 offset:2048\n\tds_read_b32 %3, %5 offset:3072\n\t"
                         "ds_read_b32 %0, %5 offset:256\n\tds_read_b32 %1, %5 offset:1280\n\tds_read_b32 %2, %5 offset:2304\n\tds_read_b32 %3, %5 offset:3328\n\ts_mov_b64 exec, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&s"(sv) : "v"(addr), "s"(m) : "memory", "scc");
            a0.x += b0 + b1 + b2 + b3;
        }
        acc += a0.x + a1.y + a2.z + a3.w;
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    sink[tid] = acc;
}
template <int W, bool MASK> static void run(const char* name) {
    long long* dc; int* ds;
    hipMalloc(&dc, 8); hipMalloc(&ds, 1024);
    const int iters = 20000;
    hipLaunchKernelGGL((k<W, MASK>), dim3(1), dim3(256), 32768, 0, dc, ds, iters);
    hipLaunchKernelGGL((k<W, MASK>), dim3(1), dim3(256), 32768, 0, dc, ds, iters);
    long long c = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    // 4 wavefronts x 8 reads per iteration share the CU's LDS
    printf("%-40s %6.1f cycles per wave-instruction (4 wavefronts asking)  %6.1f bytes/clk/CU\n", name, (double)c / iters / 32.0, (MASK ? 28.0 : 64.0) * W * 32.0 * iters / (double)c);
    hipFree(dc); hipFree(ds);
}
int main() {
    run<16, false>("ds_read_b128, all lanes");
    run<16, true>("ds_read_b128, 7 of 16 lanes");
    run<8, false>("ds_read_b64, all lanes");
    run<8, true>("ds_read_b64, 7 of 16 lanes");
    run<4, false>("ds_read_b32, all lanes");
    return 0;
}

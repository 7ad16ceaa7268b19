// Micro-benchmark: one-way hand-off latency between two workgroups on the same XCD vs on
// different XCDs, for the store/load flavours the sweep kernel could use.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handoff.hip -o tools/micro/bin/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// slots: reg[xcc*64 + k] = blockIdx of the k-th workgroup registered on that XCD; cnt[xcc] = count
// mode 0: same XCD, plain store + sc1 load.  mode 1: same XCD, sc1 store + sc1 load.  mode 2: cross XCD sc1/sc1.
__global__ void k(unsigned* cnt, unsigned* reg, unsigned long long* ball, long long* out, int mode, int iters, unsigned* done) {
    __shared__ unsigned s_x, s_rank;
    if (threadIdx.x == 0) {
        s_x = xcc_id();
        s_rank = atomicAdd(&cnt[s_x], 1u);
        if (s_rank < 64) reg[s_x * 64 + s_rank] = blockIdx.x;
        __threadfence();
        atomicAdd(done, 1u);
        while (__hip_atomic_load(done, RLX) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const unsigned x = s_x, r = s_rank;
    // players: A = (xcc 0, rank 0); B = (xcc 0, rank 1) for modes 0/1, (xcc 1, rank 0) for mode 2
    const bool isA = (x == 0 && r == 0);
    const bool isB = (mode == 2) ? (x == 1 && r == 0) : (x == 0 && r == 1);
    if (!(isA || isB) || threadIdx.x != 0) return;
    unsigned long long* mine = ball + (isA ? 0 : 16);      // separate 128-B lines
    unsigned long long* theirs = ball + (isA ? 16 : 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; i++) {
        if (isA) {
            if (mode == 0) *(volatile unsigned long long*)theirs = (unsigned long long)i; else __hip_atomic_store((gu64*)theirs, (unsigned long long)i, RLX);
            while (__hip_atomic_load((gu64*)mine, RLX) != (unsigned long long)i) {}
        } else {
            while (__hip_atomic_load((gu64*)mine, RLX) != (unsigned long long)i) {}
            if (mode == 0) *(volatile unsigned long long*)theirs = (unsigned long long)i; else __hip_atomic_store((gu64*)theirs, (unsigned long long)i, RLX);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (isA) out[0] = (long long)(t1 - t0);
}
int main() {
    unsigned *cnt, *reg, *done; unsigned long long* ball; long long* out;
    hipMalloc(&cnt, 64); hipMalloc(&reg, 16 * 64 * 4); hipMalloc(&ball, 4096); hipMalloc(&out, 64); hipMalloc(&done, 64);
    const char* names[3] = {"same XCD, plain store + sc1 load", "same XCD, sc1 store + sc1 load", "cross XCD, sc1 store + sc1 load"};
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(cnt, 0, 64); hipMemset(ball, 0, 4096); hipMemset(out, 0, 64); hipMemset(done, 0, 64);
            const int iters = 20000;
            hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, cnt, reg, ball, out, mode, iters, done);
            hipDeviceSynchronize();
            long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            std::vector<unsigned> c(16); hipMemcpy(c.data(), cnt, 64, hipMemcpyDeviceToHost);
            if (rep == 1) printf("%-36s: %.3f us per one-way hop  (workgroups per XCD: %u %u %u %u %u %u %u %u)\n", names[mode],
                                 t * 0.01 / iters / 2.0, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
        }
    }
    return 0;
}

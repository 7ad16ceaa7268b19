// Which XCD does workgroup i run on?  Prints HW_REG_XCC_ID for a 256-workgroup, one-per-CU launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcc_id.hip -o tools/micro/bin/xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 1) void k(unsigned* out) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = v;
    }
}
int main() {
    const int W = 256;
    unsigned* d;
    hipMalloc(&d, W * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k, dim3(W), dim3(256), 100 * 1024, 0, d);
    std::vector<unsigned> h(W);
    hipMemcpy(h.data(), d, W * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; i++) printf("wg %3d: raw 0x%08x xcc %u\n", i, h[i], h[i] & 0xf);
    int cnt[16] = {0};
    for (int i = 0; i < W; i++) cnt[h[i] & 0xf]++;
    printf("workgroups per XCC id:");
    for (int i = 0; i < 16; i++) printf(" %d", cnt[i]);
    printf("\n");
    return 0;
}

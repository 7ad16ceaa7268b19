// Micro-test for the next step of phase A: the batch's dot products on the block-scaled matrix
// instruction v_mfma_scale_f32_16x16x128_f8f6f4 with
//   operand A = genotype values as FP4 (E2M1): the ring's 2-bit code c' in the low half of a nibble IS the FP4 code of
//               c'/2 (0 -> 0, 1 -> 0.5, 2 -> 1.0, 3 -> 1.5), so `dword & 0x33333333` and `(dword >> 2) & 0x33333333`
//               turn a ring dword (16 individuals) into two operand registers (8 individuals each) -- ONE v_and per register,
//   operand B = signed base-16 digits (-8..7) of the residual's grid integer as FP6 (E3M2: every integer up to 8 is exact),
//               13 planes for the 52 bits + 2 stop planes = 15 of the 16 columns (the int8 form uses 7 of 16),
//   K = 128 individuals per instruction against 64 for v_mfma_i32_16x16x64_i8.
// Checks (1) the operand lane / element maps with exact data, (2) that the f32 accumulation of these products is exact
// (multiples of 0.5 far below 2^24), (3) cycles per instruction, both forms, back to back on one wavefront with four
// independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fp4_fp6.hip -o tools/micro/bin/mfma_fp4_fp6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint32_t e3m2(int d) {             // integer -8..8 -> FP6 E3M2 code (bias 3)
    const uint32_t s = d < 0 ? 32u : 0u;
    const int a = d < 0 ? -d : d;
    static const uint8_t code[9] = {0, (3 << 2) | 0, (4 << 2) | 0, (4 << 2) | 2, (5 << 2) | 0, (5 << 2) | 1, (5 << 2) | 2, (5 << 2) | 3, (6 << 2) | 0};
    return s | code[a];
}

// ring: [16 markers][8 dwords] 2-bit codes of 128 individuals (individual i of marker m: field i & 15 of dword i >> 4)
// digits: [128 individuals][16 planes] int8 in -8..7
__global__ void k_check(const uint32_t* ring, const int8_t* digits, float* out) {
    const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
    // A: K elements 32 kg + j, j = 0..31 <- the lane's two ring dwords (2 kg, 2 kg + 1): register r = 2 q + h holds
    // fields h, h + 2, ..., h + 14 of dword q, i.e. individuals 32 kg + 16 q + 2 e + h for nibble e
    const uint32_t x0 = ring[m * 8 + 2 * kg], x1 = ring[m * 8 + 2 * kg + 1];
    v8i a = {(int)(x0 & 0x33333333u), (int)((x0 >> 2) & 0x33333333u), (int)(x1 & 0x33333333u), (int)((x1 >> 2) & 0x33333333u), 0, 0, 0, 0};
    // B: column n = lane & 15, the same individual order: element j = 8 r + e <-> individual 32 kg + 16 (r >> 1) + 2 e + (r & 1)
    unsigned long long bits[3] = {0, 0, 0};
    for (int j = 0; j < 32; j++) {
        const int r = j >> 3, e = j & 7;
        const int ind = 32 * kg + 16 * (r >> 1) + 2 * e + (r & 1);
        const unsigned long long c = e3m2(digits[ind * 16 + m]);
        const int at = 6 * j;
        bits[at >> 6] |= c << (at & 63);
        if ((at & 63) > 58) bits[(at >> 6) + 1] |= c >> (64 - (at & 63));
    }
    v8i b = {(int)(uint32_t)bits[0], (int)(uint32_t)(bits[0] >> 32), (int)(uint32_t)bits[1], (int)(uint32_t)(bits[1] >> 32),
             (int)(uint32_t)bits[2], (int)(uint32_t)(bits[2] >> 32), 0, 0};
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 3, 0, 127, 0, 127);    // A: FP4 E2M1, B: FP6 E3M2, scales 2^0
    for (int r = 0; r < 4; r++) out[(4 * kg + r) * 16 + m] = c[r];                          // C: row 4 kg + r (marker), column lane & 15 (plane)
}

template <int FORM>
__global__ void k_rate(long long* cyc, float* sink, int iters) {
    v8i a = {(int)threadIdx.x * 0x11111, 0x22220000, 0x02020202, 0x20202020, 0, 0, 0, 0}, b = {0x12345678, 0x0fedcba9, 0x11111111, 0x22222222, 0x01010101, 0x10101010, 0, 0};
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    v4i i0 = {0, 0, 0, 0}, i1 = i0, i2 = i0, i3 = i0;
    v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (FORM == 0) {
            i0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, i0, 0, 0, 0);
            i1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, i1, 0, 0, 0);
            i2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, i2, 0, 0, 0);
            i3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, i3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 4, 3, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 4, 3, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 4, 3, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 4, 3, 0, 127, 0, 127);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[FORM] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (float)(i0[0] + i1[1] + i2[2] + i3[3]);
}

int main() {
    std::vector<uint32_t> ring(16 * 8);
    std::vector<int8_t> dig(128 * 16);
    std::vector<int> g(16 * 128);
    srand(7);
    for (int m = 0; m < 16; m++)
        for (int i = 0; i < 128; i++) {
            const int c = rand() % 4;                                   // 3 (missing) included: its FP4 value is 1.5
            g[m * 128 + i] = c;
            ring[m * 8 + (i >> 4)] |= (uint32_t)c << (2 * (i & 15));
        }
    for (auto& d : dig) d = (int8_t)(rand() % 16 - 8);
    uint32_t* d_ring; int8_t* d_dig; float* d_out; long long* d_cyc; float* d_sink;
    hipMalloc(&d_ring, ring.size() * 4); hipMalloc(&d_dig, dig.size()); hipMalloc(&d_out, 256 * 4); hipMalloc(&d_cyc, 16); hipMalloc(&d_sink, 64 * 4);
    hipMemcpy(d_ring, ring.data(), ring.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_dig, dig.data(), dig.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d_ring, d_dig, d_out);
    std::vector<float> out(256);
    hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; m++)
        for (int n = 0; n < 16; n++) {
            int s = 0;
            for (int i = 0; i < 128; i++) s += g[m * 128 + i] * dig[i * 16 + n];
            if (out[m * 16 + n] != 0.5f * (float)s) { if (bad < 5) printf("mismatch marker %d plane %d: got %g want %g\n", m, n, out[m * 16 + n], 0.5 * s); bad++; }
        }
    printf("operand maps + exactness (FP4 codes by mask x FP6 E3M2 digits, K = 128): %s (%d of 256 wrong)\n", bad ? "WRONG" : "exact", bad);
    const int iters = 20000;
    hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64), 0, 0, d_cyc, d_sink, iters);
    hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, d_cyc, d_sink, iters);
    long long cyc[2];
    hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per instruction (one wavefront, 4 independent accumulators): v_mfma_i32_16x16x64_i8 %.2f, "
           "v_mfma_scale_f32_16x16x128_f8f6f4 (FP4 x FP6) %.2f  ->  individuals per tick %.1f vs %.1f\n",
           (double)cyc[0] / (4.0 * iters), (double)cyc[1] / (4.0 * iters), 64.0 / ((double)cyc[0] / (4.0 * iters)), 128.0 / ((double)cyc[1] / (4.0 * iters)));
    return bad ? 1 : 0;
}

// ops.hip -- the reference's per-call kernels as gfx950 HIP kernels, one per member
// function on the hot path (file:line relative to /root/reference/):
//   k_dot           Bayes::dot_product               src/bayes.cpp:709-770 (active 749-766)
//   k_update        Phenotype::update_epsilon        src/phenotype.cpp:326-393 (active 375-390)
//   k_offset        Phenotype::offset_epsilon        src/phenotype.cpp:395-411
//   k_sumsq         Phenotype::epsilon_sumsqr        src/phenotype.cpp:251-261
//                   Phenotype::update_epsilon_sigma  src/phenotype.cpp:432-459
//   k_marker_stats  PhenMgr::compute_markers_statistics  src/phenotype.cpp:466-556
// plus the synthetic-genotype generator and the residual-exchange helpers.
//
// All are HBM-streaming byte/word kernels: coalesced 4..16-byte loads of the 2-bit column,
// the 4-entry (a,b) genotype table staged in LDS (the reference's 256x4 dotp_lut rows are
// four copies of these 4 entries, one per 2-bit field), wavefront shuffle reductions.  The
// sums are accumulated on pre-rounded bins (gm_common.h split2), hence exactly and in any
// order, so the f64 atomics that combine workgroups do not make results run-dependent.
#include "gm_common.h"
#include <type_traits>
#include "gm_rng.h"
#include "gm_internal.h"

namespace gm {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- dot: out4 += {sum a*q1, sum a*q2, sum b*q1, sum b*q2} over the column ----------
__global__ __launch_bounds__(256) void k_dot(const uint8_t* __restrict__ col, const double* __restrict__ eps,
                                             size_t nwords, double* out4) {
    __shared__ double2 lut[4];
    __shared__ double red[4][4];
    if (threadIdx.x < 4) lut[threadIdx.x] = make_double2(dcode_a(threadIdx.x), dcode_b(threadIdx.x));   // indexed by the device code
    __syncthreads();
    double sa1 = 0.0, sa2 = 0.0, sb1 = 0.0, sb2 = 0.0;
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(col);
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (size_t)gridDim.x * blockDim.x) {
        const uint32_t word = cw[w];
        const double* e = eps + w * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            double q1, q2;
            split2(e[k], q1, q2);
            const double2 ab = lut[(word >> (2 * k)) & 3u];
            sa1 = fma_(ab.x, q1, sa1); sa2 = fma_(ab.x, q2, sa2);
            sb1 = fma_(ab.y, q1, sb1); sb2 = fma_(ab.y, q2, sb2);
        }
    }
    sa1 = wave_sum(sa1); sa2 = wave_sum(sa2); sb1 = wave_sum(sb1); sb2 = wave_sum(sb2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave][0] = sa1; red[wave][1] = sa2; red[wave][2] = sb1; red[wave][3] = sb2; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        unsafeAtomicAdd(&out4[threadIdx.x], v);
    }
}

hipError_t launch_dot(const uint8_t* col, const uint8_t*, const double* eps, size_t stride, double* out4,
                      hipStream_t st) {
    const size_t nwords = stride / 4;
    int blocks = (int)((nwords + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_dot, dim3(blocks), dim3(256), 0, st, col, eps, nwords, out4);
    return hipGetLastError();
}

// ---- update: eps_i += val[code_i] (device code: val[a] for a = 0, 1, 2, val[3] = 0), code forced to 3 where the phenotype is NA
// Thread = two individuals (half a column byte, 16 bytes of the residual): a wave instruction touches 1 KB of contiguous
// residual (one thread per column dword of 16 individuals read 64 lines per instruction: 5.4 us for 500k individuals).
__global__ __launch_bounds__(256) void k_update(double* __restrict__ eps, const uint8_t* __restrict__ col,
                                                const uint8_t* __restrict__ namask2, size_t npairs,
                                                double v0, double v1, double v2, double v3) {
    __shared__ double val[4];
    if (threadIdx.x == 0) { val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; }
    __syncthreads();
    double2* e2 = reinterpret_cast<double2*>(eps);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += (size_t)gridDim.x * blockDim.x) {
        const uint32_t sh = 4u * (uint32_t)(p & 1);
        const uint32_t m = ((uint32_t)namask2[p >> 1] >> sh) & 0xFu;
        const uint32_t c = (((uint32_t)col[p >> 1] >> sh) | ~m) & 0xFu;
        double2 e = e2[p];
        e.x += val[c & 3u];
        e.y += val[(c >> 2) & 3u];
        e2[p] = e;
    }
}

hipError_t launch_update(double* eps, const uint8_t* col, const uint8_t* namask2, size_t stride,
                         double v0, double v1, double v2, double v3, hipStream_t st) {
    const size_t npairs = stride * 2;
    int blocks = (int)((npairs + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_update, dim3(blocks), dim3(256), 0, st, eps, col, namask2, npairs, v0, v1, v2, v3);
    return hipGetLastError();
}

// ---- offset: eps_i += off where the phenotype is present ------------------------------
__global__ __launch_bounds__(256) void k_offset(double* __restrict__ eps, const uint8_t* __restrict__ namask2,
                                                size_t npairs, double off) {
    double2* e2 = reinterpret_cast<double2*>(eps);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += (size_t)gridDim.x * blockDim.x) {
        const uint32_t m = (uint32_t)namask2[p >> 1] >> (4u * (uint32_t)(p & 1));
        double2 e = e2[p];
        if (m & 1u) e.x += off;
        if (m & 4u) e.y += off;
        e2[p] = e;
    }
}

hipError_t launch_offset(double* eps, const uint8_t* namask2, size_t stride, double off, hipStream_t st) {
    const size_t npairs = stride * 2;
    int blocks = (int)((npairs + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_offset, dim3(blocks), dim3(256), 0, st, eps, namask2, npairs, off);
    return hipGetLastError();
}

// ---- sum of squares over eps[0..n): out2 += {sum q1, sum q2} of split2sq(eps_i^2);
//      outmax = max |eps_i| (as ordered bits) for the range check of the exactness argument
__global__ __launch_bounds__(256) void k_sumsq(const double* __restrict__ eps, const uint8_t* __restrict__ namask2,
                                               size_t n, double* out2, unsigned long long* outmax) {
    __shared__ double red[4][2];
    __shared__ unsigned long long redm[4];
    double s1 = 0.0, s2 = 0.0;
    unsigned long long mx = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double x = eps[i];
        if (namask2 && !((namask2[i >> 2] >> (2 * (i & 3))) & 1u)) x = 0.0;   // eps^2 * na_lut (phenotype.cpp:455)
        double q1, q2;
        split2sq(x * x, q1, q2);
        s1 += q1; s2 += q2;
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(x));
        mx = b > mx ? b : mx;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; redm[wave] = mx; }
    __syncthreads();
    if (threadIdx.x < 2) unsafeAtomicAdd(&out2[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x]
                                                              + red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x == 2) {
        unsigned long long m = redm[0];
        for (int w = 1; w < 4; w++) m = redm[w] > m ? redm[w] : m;
        atomicMax(outmax, m);
    }
}

hipError_t launch_sumsq(const double* eps, const uint8_t* namask2, size_t n, double* out2, double* outmax,
                        hipStream_t st) {
    // every workgroup ends in three atomics on the same words: at 1024 workgroups those took 25 of the kernel's 27.6 us
    // (500k individuals); 128 workgroups of 16 elements per thread read the 4 MB just as fast
    int blocks = (int)((n + 255) / 256);
    if (blocks > 128) blocks = 128;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sumsq, dim3(blocks), dim3(256), 0, st, eps, namask2, n, out2,
                       reinterpret_cast<unsigned long long*>(outmax));
    return hipGetLastError();
}

// ---- marker statistics: one wavefront per marker, genotype counts by popcount ---------
// mave = (2*n0 + n2) / (n0 + n2 + n3);  msig = 1/sqrt((n0*v0^2 + n2*v2^2 + n3*v3^2)/(nonas-1))
// with n_c = #{present individuals with code c}, v0 = 2-mave, v2 = 1-mave, v3 = 0-mave.
__global__ __launch_bounds__(256) void k_marker_stats(const uint8_t* __restrict__ bed,
                                                      const uint8_t* __restrict__ namask2, size_t stride, int M,
                                                      int nonas, double* __restrict__ mave, double* __restrict__ msig,
                                                      uint8_t* __restrict__ nomiss) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    const uint4* col = reinterpret_cast<const uint4*>(bed + (size_t)m * stride);
    const uint4* msk = reinterpret_cast<const uint4*>(namask2);
    const size_t nvec = stride / 16;
    int n0 = 0, n2 = 0, n3 = 0;
    for (size_t v = lane; v < nvec; v += 64) {
        const uint4 w4 = col[v];
        const uint4 m4 = msk[v];
        const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
        const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t p = mm[i] & 0x55555555u;
            const uint32_t lo = ww[i] & 0x55555555u;                 // device code: field = a (10: a = 2, 01: a = 1, 00: a = 0), 11 = missing
            const uint32_t hi = (ww[i] >> 1) & 0x55555555u;
            n0 += __popc(p & ~lo & hi);                              // n0, n2, n3 keep the .bed codes' names: a = 2, 1, 0
            n2 += __popc(p & lo & ~hi);
            n3 += __popc(p & ~lo & ~hi);
        }
    }
    n0 = wave_sum_i(n0); n2 = wave_sum_i(n2); n3 = wave_sum_i(n3);
    if (lane == 0) {
        const double suma = (double)(2ll * n0 + n2);
        const double sumb = (double)((long long)n0 + n2 + n3);
        const double av = suma / sumb;
        const double v0 = 2.0 - av, v2 = 1.0 - av, v3 = 0.0 - av;
        double s = (double)n0 * (v0 * v0);
        s += (double)n2 * (v2 * v2);
        s += (double)n3 * (v3 * v3);
        mave[m] = av;
        msig[m] = 1.0 / __builtin_sqrt(s / ((double)nonas - 1.0));
        nomiss[m] = (n0 + n2 + n3 == nonas) ? 1 : 0;      // no missing genotype among the phenotyped individuals
    }
}

hipError_t launch_marker_stats(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, int nonas,
                               double* mave, double* msig, uint8_t* nomiss, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_marker_stats, dim3((M + 3) / 4), dim3(256), 0, st, bed, namask2, stride, M, nonas, mave, msig, nomiss);
    return hipGetLastError();
}

// ---- .bed code <-> device code over a range of columns (gm_common.h), in place, 16 bytes per thread
__global__ __launch_bounds__(256) void k_recode(uint4* __restrict__ p, size_t nvec, int back) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        if (back) { v.x = dcode_to_bed(v.x); v.y = dcode_to_bed(v.y); v.z = dcode_to_bed(v.z); v.w = dcode_to_bed(v.w); }
        else      { v.x = bed_to_dcode(v.x); v.y = bed_to_dcode(v.y); v.z = bed_to_dcode(v.z); v.w = bed_to_dcode(v.w); }
        p[i] = v;
    }
}
hipError_t launch_recode(uint8_t* bed, size_t nbytes, int back, hipStream_t st) {     // nbytes: a multiple of 16 (whole columns)
    if (nbytes == 0) return hipSuccess;
    const size_t nvec = nbytes / 16;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_recode, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(bed), nvec, back);
    return hipGetLastError();
}

// ---- the sampling step's per-marker inputs in VISIT order (round 4) --------------------------------------------
// The sweep kernel's workgroups all need, for every order position, the marker's id, group, previous effect, mean and
// scale.  Gathered inside the persistent kernel that was 4 loads of 64 random addresses per 64 positions and loader
// wavefront -- the same gathers in every one of the 245 workgroups, each address another page for the address pipeline
// the genotype loads go through.  One pass of this kernel in front of the sweep puts them in order-major arrays; the
// sweep kernel then reads them as five contiguous streams (L2 hits: every workgroup reads the same lines).
__global__ __launch_bounds__(256) void k_order_inputs(const int* __restrict__ order, int count, const int* __restrict__ group,
                                                      const double* __restrict__ betas, const double* __restrict__ mave,
                                                      const double* __restrict__ msig, const uint8_t* __restrict__ nomiss,
                                                      int* __restrict__ o_g, double* __restrict__ o_beta, double* __restrict__ o_mave,
                                                      double* __restrict__ o_msig, uint8_t* __restrict__ o_nm) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < count; p += gridDim.x * blockDim.x) {
        const int m = order[p];
        o_g[p] = group[m]; o_beta[p] = betas[m]; o_mave[p] = mave[m]; o_msig[p] = msig[m]; o_nm[p] = nomiss[m];
    }
}
hipError_t launch_order_inputs(const int* order, int count, const int* group, const double* betas, const double* mave, const double* msig,
                               const uint8_t* nomiss, int* o_g, double* o_beta, double* o_mave, double* o_msig, uint8_t* o_nm, hipStream_t st) {
    if (count <= 0) return hipSuccess;
    int blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_order_inputs, dim3(blocks), dim3(256), 0, st, order, count, group, betas, mave, msig, nomiss, o_g, o_beta, o_mave, o_msig, o_nm);
    return hipGetLastError();
}

// ---- synthetic genotypes, keyed by (seed, global marker, byte) ------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Correlated markers (round 4): ld_block > 1 puts the markers in blocks of ld_block consecutive (global) indices.
// Inside a block a haplotype copies its allele from the marker before it with probability ld_keep (a 16-bit fraction) and draws
// a fresh one otherwise (the first marker of a block always draws): neighbours correlate with r ~ ld_keep, markers k apart
// with r ~ ld_keep^k, blocks are independent -- the shape of real linkage disequilibrium, which the reference's simulation
// recipe (example/data_sim.R:5-41, independent Binomial(2, maf) draws) lacks.  Every allele stays a pure function of
// (seed, marker, individual): a thread walks back to the last fresh draw (at most ld_block - 1 steps).
// (LD = false is the independent generator as it always was: the walk's arrays cost registers and a loop that the 125 GB
//  block of the benchmarks should not pay for -- 549 ms instead of 25 when both shared one kernel.)
template <bool LD>
__global__ __launch_bounds__(256) void k_synth(uint8_t* __restrict__ bed, size_t stride, int N, int M, int S,
                                               uint64_t seed, uint32_t maf16, uint32_t miss16, int ld_block, uint32_t ld_keep16) {
    const size_t mbytes = ((size_t)N + 3) / 4;
    const size_t total = (size_t)M * stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / stride, b = i % stride;
        uint8_t out = 0;
        if (b < mbytes) {
            auto draws = [&](size_t mg, uint64_t& za, uint64_t& zb, uint64_t& zm) {
                const uint64_t key = ((uint64_t)mg * (uint64_t)mbytes + b) * 3ull;
                za = mix64(seed + (key + 0) * 0x9E3779B97F4A7C15ull);
                zb = mix64(seed + (key + 1) * 0x9E3779B97F4A7C15ull);
                zm = mix64(seed + (key + 2) * 0x9E3779B97F4A7C15ull);
            };
            const size_t mg = (size_t)S + m;
            uint64_t za, zb, zm;
            draws(mg, za, zb, zm);
            uint32_t al_a[4], al_b[4];                      // the two haplotypes' alleles of the byte's four individuals
#pragma unroll
            for (int k = 0; k < 4; k++) {
                al_a[k] = ((uint32_t)(za >> (16 * k)) & 0xFFFFu) < maf16;
                al_b[k] = ((uint32_t)(zb >> (16 * k)) & 0xFFFFu) < maf16;
            }
            if constexpr (LD) {
                // walk back through the block: haplotype h of individual k keeps copying while its "keep" draw says so
                const int off = (int)(mg % (size_t)ld_block);
                uint32_t open_a = 0xFu, open_b = 0xFu;       // bit k: the allele is still to be found further back
                uint64_t ka = za, kb = zb;                   // the keep draws come from the high parts of a second mix of the same keys
                size_t mcur = mg;
                for (int back = 0; back < off && (open_a | open_b); back++) {
                    const uint64_t ca = mix64(ka ^ 0xD1B54A32D192ED03ull), cb = mix64(kb ^ 0xD1B54A32D192ED03ull);
                    uint32_t keep_a = 0, keep_b = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        keep_a |= ((((uint32_t)(ca >> (16 * k)) & 0xFFFFu) < ld_keep16) ? 1u : 0u) << k;
                        keep_b |= ((((uint32_t)(cb >> (16 * k)) & 0xFFFFu) < ld_keep16) ? 1u : 0u) << k;
                    }
                    open_a &= keep_a; open_b &= keep_b;      // a haplotype that does not copy here has its allele (found at mcur)
                    if (!(open_a | open_b)) break;
                    mcur--;
                    uint64_t pa, pb, pm;
                    draws(mcur, pa, pb, pm);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((open_a >> k) & 1u) al_a[k] = ((uint32_t)(pa >> (16 * k)) & 0xFFFFu) < maf16;
                        if ((open_b >> k) & 1u) al_b[k] = ((uint32_t)(pb >> (16 * k)) & 0xFFFFu) < maf16;
                    }
                    ka = pa; kb = pb;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (b * 4 + k >= (size_t)N) break;          // pad bits stay 00 as PLINK writes them
                const uint32_t um = (uint32_t)(zm >> (16 * k)) & 0xFFFFu;
                const int copies = (int)al_a[k] + (int)al_b[k];
                uint32_t code = copies == 2 ? 0u : (copies == 1 ? 2u : 3u);
                if (um < miss16) code = 1u;
                out |= (uint8_t)(code << (2 * k));
            }
        }
        bed[i] = (uint8_t)bed_to_dcode(out);                        // stored in the device code (gm_common.h)
    }
}

hipError_t launch_synth(uint8_t* bed, size_t stride, int N, int M, int S, uint64_t seed, double maf, double miss,
                        int ld_block, double ld_keep, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    const uint32_t maf16 = (uint32_t)(maf * 65536.0);
    const uint32_t miss16 = (uint32_t)(miss * 65536.0);
    const uint32_t keep16 = (uint32_t)(ld_keep * 65536.0);
    if (ld_block > 1) hipLaunchKernelGGL(k_synth<true>, dim3(256 * 32), dim3(256), 0, st, bed, stride, N, M, S, seed, maf16, miss16, ld_block, keep16);
    else hipLaunchKernelGGL(k_synth<false>, dim3(256 * 32), dim3(256), 0, st, bed, stride, N, M, S, seed, maf16, miss16, ld_block, keep16);
    return hipGetLastError();
}

// ---- residual exchange helpers -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_delta_export(const double* __restrict__ eps, const double* __restrict__ start,
                                                      double* __restrict__ q, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        double q1, q2;
        split2(eps[i] - start[i], q1, q2);
        q[i] = q1;
        q[n4 + i] = q2;
    }
}
__global__ __launch_bounds__(256) void k_delta_import(double* __restrict__ eps, const double* __restrict__ start,
                                                      const double* __restrict__ q, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        eps[i] = start[i] + (q[i] + q[n4 + i]);
}
hipError_t launch_delta_export(const double* eps, const double* start, double* q, size_t n4, hipStream_t st) {
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_delta_export, dim3(blocks), dim3(256), 0, st, eps, start, q, n4);
    return hipGetLastError();
}
hipError_t launch_delta_import(double* eps, const double* start, const double* q, size_t n4, hipStream_t st) {
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_delta_import, dim3(blocks), dim3(256), 0, st, eps, start, q, n4);
    return hipGetLastError();
}

// ---- device arithmetic self-test (gmrm_selftest_math) --------------------------------------
struct DevMt {                                        // a plain mt19937 in registers/scratch, one thread
    uint32_t mt[624];
    int idx;
    __device__ void seed(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    __device__ uint32_t u32() {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) mt[i] = mt_twist1(mt[i], mt[(i + 1) % 624], mt[(i + 397) % 624]);
            idx = 0;
        }
        return mt_temper(mt[idx++]);
    }
};
// ---- Bayes::predict (src/bayes.cpp:16-284), SURVEY section 8f-3 ---------------------------------
// g_i = sum over this block's markers, in marker order, of ((a - mave) * b * na * msig) * beta_m
// (bayes.cpp:113-122; the "transposed" use of the genotype table).  Thread = 8 individuals kept in
// registers for the whole pass; markers with a zero mean effect add +0.  The four
// values a marker can add (one per genotype code, computed exactly as the reference's expression) are
// staged in LDS per chunk of U markers and picked by code with ONE ds_read per individual: the 4-way
// register select it replaces cost 6 v_cndmask per individual and made the kernel VALU-bound
// (233 GB/s at 500k x 1M; profiles/README.md).
// BW = bytes of every column per thread (4 BW individuals): 1 gives twice the wavefronts of 2 for the same work --
// at 500k individuals that is two per SIMD instead of one, which is what hides the LDS latency.
template <int BW>
__global__ __launch_bounds__(256) void k_predict_g(const uint8_t* __restrict__ bed, const uint8_t* __restrict__ namask2,
                                                   size_t stride, int M, const double* __restrict__ mave,
                                                   const double* __restrict__ msig, const double* __restrict__ beta,
                                                   double* __restrict__ g) {
    constexpr int U = 32;                                           // column words in flight per thread (HBM latency)
    constexpr int NI = 4 * BW;                                      // individuals per thread
    typedef typename std::conditional<BW == 1, uint8_t, uint16_t>::type word_t;
    __shared__ double s_tv[2][U][4];                                // [chunk parity][marker of the chunk][genotype code]
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;          // BW-byte word of every column
    const bool live = w * BW < stride;                                // (threads beyond the column still help with the tables)
    const uint32_t nam = live ? reinterpret_cast<const word_t*>(namask2)[w] : 0u;
    const uint32_t force = ~nam & (BW == 1 ? 0xFFu : 0xFFFFu);     // NA individuals read "missing" (device code 3)
    double acc[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) acc[i] = 0.0;
    // Everything a chunk needs from global memory is requested one chunk ahead: its U column words and, for the
    // 128 threads that build the value table, the effect and the statistics of their marker.
    const int tu = threadIdx.x >> 2, tc = threadIdx.x & 3;          // table entry of this thread: (marker of the chunk, code)
    const bool builder = threadIdx.x < U * 4;
    uint32_t wn[U];
    double nb_ = 0.0, nmv = 0.0, nms = 0.0;
    auto request = [&](int m0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int m = m0 + u < M ? m0 + u : M - 1;
            wn[u] = live ? reinterpret_cast<const word_t*>(bed + (size_t)m * stride)[w] : 0u;
        }
        if (builder) {
            const int m = m0 + tu;
            nb_ = m < M ? beta[m] : 0.0;
            nmv = m < M ? mave[m] : 0.0;
            nms = m < M ? msig[m] : 0.0;
        }
    };
    request(0);
    int par = 0;
    for (int m0 = 0; m0 < M; m0 += U, par ^= 1) {
        uint32_t wd[U];
#pragma unroll
        for (int u = 0; u < U; u++) wd[u] = wn[u];
        if (builder) {
            double tv = 0.0;
            if (nb_ != 0.0) tv = (((dcode_a(tc) - nmv) * dcode_b(tc)) * nms) * nb_;
            s_tv[par][tu][tc] = tv;
        }
        if (m0 + U < M) request(m0 + U);
        __syncthreads();                                            // one barrier per chunk: the other parity is free by now
        // No test per marker: a marker with a zero effect (or behind the end of the block) adds +0.0, which leaves
        // an accumulator as it is (an accumulator is never -0.0: it starts at +0.0 and x + (-x) rounds to +0.0), and a
        // wave-uniform branch per marker would keep the scheduler from issuing the next marker's table reads
        // under this one's additions -- the kernel is bound by exactly those LDS reads.  (Measured and dropped: a
        // 16-entry table per marker serving two individuals with one 16-byte read, 0.82 TB/s against 1.08; half a byte
        // per thread, i.e. four wavefronts per SIMD instead of two, 1.06 against 1.28.)
        const char* tvb = reinterpret_cast<const char*>(&s_tv[par][0][0]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t x = wd[u] | force;
#pragma unroll
            for (int i = 0; i < NI; i++)
                acc[i] += *reinterpret_cast<const double*>(tvb + u * 32 + (((x >> (2 * i)) & 3u) << 3));
        }
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < NI; i++) g[NI * w + i] = acc[i];
    }
}

// Per marker: xtx = sum (a*b*na)^2 (an integer: #(a=1) + 4 #(a=2)), xty = sum a*b*na*y_i (bayes.cpp:188-196).
// Round 1 did this with one wavefront per 8 markers and a 3-way select + FMA per genotype: VALU-bound at 0.8 TB/s.
// ---- k_assoc on the matrix cores ----------------------------------------------------------------------------------
// xty_m = sum_i a_im y_i is the sweep kernel's phase A with y in the place of the residual (sweep.hip, "phase A
// building blocks"): y is scaled by a power of two into (-2^8, 2^8), cut into two exact parts (split2) and those into
// eight signed base-256 digit planes, one byte per individual, in the order the B operand of v_mfma_i32_16x16x64_i8
// wants (position 16 i + 4 j + b of a 64-individual chunk <- individual 16 j + 4 b + i); a wavefront turns the
// 16-byte chunks of 2 x 16 columns into 2-bit fields holding a (one v_and per field and register) and accumulates
// int32 sums of a * digit per plane.  The result is the EXACT sum of the two parts, rounded once -- closer to the
// real number than the reference's left-to-right f64 sum (tests: 1e-12 relative).  xtx (= #(a=1) + 4 #(a=2)) comes
// from two popcounts per dword.  The genotype columns are read once, 16 bytes per lane; the planes of a 2048-individual
// block are staged in LDS for the four wavefronts (32 markers each) of a workgroup.
constexpr int AS_BLK = 512;                       // column bytes per block (256: two wavefronts per SIMD at 200 VGPRs, same speed)
constexpr int AS_SS = AS_BLK / 64;                // super-steps (256 individuals) per block
constexpr int AS_PST = 4 * AS_BLK + 16;           // LDS bytes per plane: the offsets of sweep.hip's Geo<R>::PSTRIDE (bank spreading)
constexpr int AS_PLN = 8 * AS_PST + 64;
constexpr int AS_MW = 32, AS_MB = 4 * AS_MW;      // markers per wavefront / per workgroup
typedef int as_v4i __attribute__((ext_vector_type(4)));

__global__ void k_absmax(const double* __restrict__ y, const uint8_t* __restrict__ namask2, size_t n, unsigned long long* __restrict__ out) {
    unsigned long long m = 0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // individuals without a phenotype do not enter the sums (k_yplanes gives them zero planes): whatever y holds for
        // them -- NaN included -- must not set the scale
        if (((namask2[i >> 2] >> (2 * (i & 3))) & 3u) != 3u) continue;
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(y[i]));
        m = b > m ? b : m;                         // |y| as bits: monotone for finite values; NaN / Inf end up on top
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long v = ((unsigned long long)(unsigned)__shfl_xor((int)(m >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// the power of two that brings max |y| (given as bits) below 2^8
__device__ __forceinline__ int assoc_shift(unsigned long long maxbits) {
    const int ex = (int)((maxbits >> 52) & 0x7ffull);
    return ex == 0 ? 0 : 6 - (ex - 1023);
}
// thread = four consecutive POSITIONS of a plane (one dword of each of the eight planes)
__global__ void k_yplanes(const double* __restrict__ y, const uint8_t* __restrict__ namask2, size_t stride,
                          const unsigned long long* __restrict__ maxbits, uint8_t* __restrict__ planes, size_t npad) {
    const size_t tau = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * tau >= npad) return;
    const size_t c = tau >> 4;
    const int i = (int)(tau >> 2) & 3, j = (int)tau & 3;
    const int sh = assoc_shift(*maxbits);
    const bool nonfinite = ((*maxbits >> 52) & 0x7ffull) == 0x7ffull;
    uint32_t pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const size_t ind = 64 * c + 16 * (size_t)j + 4 * (size_t)b + (size_t)i;
        double v = 0.0;
        if ((ind >> 2) < stride && ((namask2[ind >> 2] >> (2 * (ind & 3))) & 3u) == 3u) v = y[ind];
        double q1, q2;
        split2(nonfinite ? 0.0 : __builtin_ldexp(v, sh), q1, q2);
        const uint32_t z1 = ((uint32_t)(int)(q1 * 0x1p22) + 0x00808080u) ^ 0x00808080u;   // signed base-256 digits
        const uint32_t z2 = ((uint32_t)(int)(q2 * 0x1p53) + 0x00808080u) ^ 0x00808080u;
#pragma unroll
        for (int n = 0; n < 4; n++) {
            pl[n] |= ((z1 >> (8 * n)) & 0xffu) << (8 * b);
            pl[n + 4] |= ((z2 >> (8 * n)) & 0xffu) << (8 * b);
        }
    }
#pragma unroll
    for (int n = 0; n < 8; n++) reinterpret_cast<uint32_t*>(planes + (size_t)n * npad)[tau] = pl[n];
}

__global__ __launch_bounds__(256) void k_assoc_mfma(const uint8_t* __restrict__ bed, const uint8_t* __restrict__ namask2,
                                                    size_t stride, int M, const uint8_t* __restrict__ planes, size_t npad,
                                                    const unsigned long long* __restrict__ maxbits,
                                                    double* __restrict__ xtx, double* __restrict__ xty) {
    __shared__ __attribute__((aligned(16))) char s_pl[2][AS_PLN];
    __shared__ __attribute__((aligned(16))) uint4 s_nm[2][AS_BLK / 16];   // the block's NA mask, staged with the planes
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mrow = lane & 15, kg = lane >> 4;
    const int mbase = blockIdx.x * AS_MB + wave * AS_MW;
    const uint8_t* col[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int m = mbase + 16 * q + mrow;
        col[q] = bed + (size_t)(m < M ? m : M - 1) * stride;
    }
    const size_t nchunk = stride / 16;                               // the column stride is a multiple of 16
    const int nblk = (int)((stride + AS_BLK - 1) / AS_BLK);
    // the planes of block 0 -> LDS buffer 0 (thread t: 8 bytes of every plane)
    typedef typename std::conditional<AS_BLK == 512, uint2, uint32_t>::type pbt;   // 4 * AS_BLK / 256 bytes per thread and plane
    constexpr int PBB = (int)sizeof(pbt);
    pbt pb[8];
#pragma unroll
    for (int n = 0; n < 8; n++) pb[n] = *reinterpret_cast<const pbt*>(planes + (size_t)n * npad + PBB * (size_t)tid);
#pragma unroll
    for (int n = 0; n < 8; n++) *reinterpret_cast<pbt*>(s_pl[0] + n * AS_PST + (n >> 2) * 64 + PBB * tid) = pb[n];
    auto nam_chunk = [&](int blk) -> uint4 {                          // thread tid < 32: chunk tid of the block's mask
        const size_t ci = (size_t)blk * (AS_BLK / 16) + (size_t)tid;
        return (tid < AS_BLK / 16 && ci < nchunk) ? *reinterpret_cast<const uint4*>(namask2 + 16 * ci) : make_uint4(0u, 0u, 0u, 0u);
    };
    uint4 pn = nam_chunk(0);
    if (tid < AS_BLK / 16) s_nm[0][tid] = pn;
    as_v4i acc0[2], acc1[2], acc2[2];
#pragma unroll
    for (int q = 0; q < 2; q++) { acc0[q] = as_v4i{0, 0, 0, 0}; acc1[q] = as_v4i{0, 0, 0, 0}; acc2[q] = as_v4i{0, 0, 0, 0}; }
    int n1[2] = {0, 0}, n2[2] = {0, 0};
    // An int32 accumulator takes up to 2^21 per block (512 individuals of one field x 32 x 128): it is folded into a 64-bit
    // total every AS_FOLD blocks, far below 2^31 (the run over a whole column of 2^22 individuals could reach it)
    constexpr int AS_FOLD = 256;
    long long tx[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto fold = [&]() {
#pragma unroll
        for (int q = 0; q < 2; q++) {
#pragma unroll
            for (int r = 0; r < 4; r++) tx[q][r] += (long long)(acc0[q][r] + (acc1[q][r] >> 2) + (acc2[q][r] >> 4));
            acc0[q] = as_v4i{0, 0, 0, 0}; acc1[q] = as_v4i{0, 0, 0, 0}; acc2[q] = as_v4i{0, 0, 0, 0};
        }
    };
    constexpr uint32_t M0 = 0x03030303u, LO = 0x55555555u;
    // a rolling window of AS_SS super-steps of column chunks: a chunk is requested again (for the next block) as soon
    // as it has been used, so that a block's worth of loads is always in flight
    uint4 wa[AS_SS][2];
    auto request = [&](int blk, int s) {
        const size_t ci = (size_t)blk * (AS_BLK / 16) + 4 * (size_t)s + (size_t)kg;
        const bool in = ci < nchunk;
#pragma unroll
        for (int q = 0; q < 2; q++) wa[s][q] = in ? *reinterpret_cast<const uint4*>(col[q] + 16 * ci) : make_uint4(0u, 0u, 0u, 0u);
    };
#pragma unroll
    for (int s = 0; s < AS_SS; s++) request(0, s);
    for (int blk = 0; blk < nblk; blk++) {
        const int par = blk & 1;
        __syncthreads();                                             // buffer par is complete; buffer par ^ 1 is free
        if (blk + 1 < nblk) {
#pragma unroll
            for (int n = 0; n < 8; n++)
                pb[n] = *reinterpret_cast<const pbt*>(planes + (size_t)n * npad + (size_t)(blk + 1) * 4 * AS_BLK + PBB * (size_t)tid);
            pn = nam_chunk(blk + 1);
        }
        const char* pbase = s_pl[par] + (lane & 7) * AS_PST + ((lane & 7) >> 2) * 64 + kg * 64;
#pragma unroll
        for (int s = 0; s < AS_SS; s++) {
            const as_v4i b0 = *reinterpret_cast<const as_v4i*>(pbase + s * 256);
            const as_v4i b1 = *reinterpret_cast<const as_v4i*>(pbase + s * 256 + 16);
            const as_v4i b2 = *reinterpret_cast<const as_v4i*>(pbase + s * 256 + 32);
            const as_v4i b3 = *reinterpret_cast<const as_v4i*>(pbase + s * 256 + 48);
            const uint4 wn = s_nm[par][4 * s + kg];
            const uint32_t nm[4] = {wn.x, wn.y, wn.z, wn.w};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t w4[4] = {wa[s][q].x, wa[s][q].y, wa[s][q].z, wa[s][q].w};
                uint32_t f[4];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t x = w4[d] | ~nm[d];                       // NA individuals read "missing" (device code 3)
                    const uint32_t t1 = x & ~(x >> 1) & LO;                  // a = 1 (01)
                    const uint32_t t2 = (x >> 1) & ~x & LO;                  // a = 2 (10)
                    n1[q] += __popc(t1);
                    n2[q] += __popc(t2);
                    f[d] = t1 | (t2 << 1);                                   // the field holds a, 0 where the genotype is missing
                }
                const as_v4i a0 = {(int)(f[0] & M0), (int)(f[1] & M0), (int)(f[2] & M0), (int)(f[3] & M0)};
                const as_v4i a1 = {(int)(f[0] & (M0 << 2)), (int)(f[1] & (M0 << 2)), (int)(f[2] & (M0 << 2)), (int)(f[3] & (M0 << 2))};
                const as_v4i a2 = {(int)(f[0] & (M0 << 4)), (int)(f[1] & (M0 << 4)), (int)(f[2] & (M0 << 4)), (int)(f[3] & (M0 << 4))};
                const as_v4i a3 = {(int)((f[0] >> 6) & M0), (int)((f[1] >> 6) & M0), (int)((f[2] >> 6) & M0), (int)((f[3] >> 6) & M0)};
                acc0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc0[q], 0, 0, 0);
                acc1[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc1[q], 0, 0, 0);
                acc2[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, acc2[q], 0, 0, 0);
                acc0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b3, acc0[q], 0, 0, 0);
            }
            if (blk + 1 < nblk) request(blk + 1, s);
        }
        if (blk + 1 < nblk) {
#pragma unroll
            for (int n = 0; n < 8; n++) *reinterpret_cast<pbt*>(s_pl[par ^ 1] + n * AS_PST + (n >> 2) * 64 + PBB * tid) = pb[n];
            if (tid < AS_BLK / 16) s_nm[par ^ 1][tid] = pn;
        }
        if ((blk & (AS_FOLD - 1)) == AS_FOLD - 1) fold();
    }
    fold();
    // C: column n = lane & 15 (digit plane n < 8), rows 4 kg + r (marker of the tile); the four planes of a part meet in a quad
    const int n = lane & 15;
    const int sh = assoc_shift(*maxbits);
    const bool nonfinite = ((*maxbits >> 52) & 0x7ffull) == 0x7ffull;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        // every lane (mrow, kg) counted its own marker over its chunks: the four kg lanes of a marker row add up
        int c1 = n1[q], c2 = n2[q];
        c1 += __shfl_xor(c1, 16, 64); c1 += __shfl_xor(c1, 32, 64);
        c2 += __shfl_xor(c2, 16, 64); c2 += __shfl_xor(c2, 32, 64);
        const int mq = mbase + 16 * q + mrow;
        if (kg == 0 && mq < M) xtx[mq] = (double)((long long)c1 + 4ll * (long long)c2);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            long long sx = tx[q][r] << (8 * (n & 3));
            sx += ((long long)__shfl_xor((int)(sx >> 32), 1, 64) << 32) + (long long)(unsigned)__shfl_xor((int)sx, 1, 64);
            sx += ((long long)__shfl_xor((int)(sx >> 32), 2, 64) << 32) + (long long)(unsigned)__shfl_xor((int)sx, 2, 64);
            // lanes n = 0..3 hold the sum of part 1 (units of 2^-22), n = 4..7 of part 2 (units of 2^-53)
            const long long s2 = ((long long)__shfl((int)(sx >> 32), (lane & 48) + 4, 64) << 32) | (long long)(unsigned)__shfl((int)sx, (lane & 48) + 4, 64);
            const int mr = mbase + 16 * q + 4 * kg + r;
            if (n == 0 && mr < M)
                xty[mr] = nonfinite ? __longlong_as_double(0x7ff8000000000000ll)   // a NaN or Inf in y: no digits to sum
                                    : __builtin_ldexp((double)sx * 0x1p-22 + (double)s2 * 0x1p-53, -sh);
        }
    }
}

__global__ void k_selftest(int op, const double* __restrict__ x, double* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (op == 3) {
        if (i == 0) {
            DevMt e;
            e.seed((uint32_t)x[0]);
            for (int k = 0; k < n; k++) y[k] = norm(e, 0.0, 1.0);
        }
        return;
    }
    if (i >= n) return;
    if (op == 0) y[i] = exp_(x[i]);
    else if (op == 1) y[i] = __builtin_sqrt(x[i]);
    else if (op == 2) y[i] = 1.0 / x[i];
    else if (op == 4) { double q1, q2; split2(x[i], q1, q2); y[2 * i] = q1; y[2 * i + 1] = q2; }
}
hipError_t launch_selftest(int op, const double* x, double* y, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, st, op, x, y, n);
    return hipGetLastError();
}

// ---- k_predict_g on the matrix cores ----------------------------------------------------------
// g_i = na_i * sum_m b_im (a_im C_m - D_m),  C_m = msig_m beta_m,  D_m = mave_m C_m   (bayes.cpp:93-122; upstream adds the terms
// with `omp atomic`, i.e. in no order: the EXACT sum of these terms rounded once is inside the contract, tests: 1e-12).
// Contraction over MARKERS: out[plane][individual] = sum_m digit_plane(C_m) * a_im with v_mfma_i32_16x16x64_i8 -- operand A =
// the eight signed base-256 digit planes of the two exact parts of C_m 2^sh (split2; rows 0..7), operand B = the genotype
// values of 16 individuals x 64 markers.  The .bed block is marker-major (a dword = 16 individuals of ONE marker), so the
// 2-bit codes are transposed on the way: the 16 lanes of a DPP row each load one marker's dword and a 16 x 16 transpose of
// 2-bit fields (four butterfly stages: DPP moves + v_alignbit + v_bfi) leaves lane n with individual n's codes of the 16
// markers.  Four such dwords are the lane's 16 bytes of operand B; as in the sweep kernel, MFMA i takes field i of every
// byte (one v_and per register) and the factor 4^i is divided out of that field's accumulator.  For a block whose markers
// have no missing genotype among the phenotyped individuals b = 1 wherever the output is kept, so
// g_i = sum_m a_im C_m - sum_m D_m with the second sum one exact constant (k_pg_planes).  A block with missing genotypes
// (DIRTY) adds the term of the missing-genotype indicator z_im = 1 - b_im:  g_i = sum_m a'_im C_m - sum_m D_m + sum_m z_im D_m
// (a' = a with 0 for a missing genotype): operand A carries the digit planes of D_m in rows 8..15 next to those of C_m in
// rows 0..7, a second MFMA set takes the indicator fields as operand B, and of each set the rows of the other are dropped.
// k_predict_g<BW> above (same values to 1e-12, in-order f64 sum) remains as the reference kernel (GMRM_PREDICT_LUT=1).
constexpr int PG_KB = 2048;                         // markers per LDS stage (8 super-steps of 256)
constexpr int PG_STAGES = 16;                       // stages per workgroup: 32 768 markers
typedef int pg_v4i __attribute__((ext_vector_type(4)));

__global__ void k_pg_absmax(const double* __restrict__ mave, const double* __restrict__ msig, const double* __restrict__ beta,
                            int M, unsigned long long* __restrict__ out) {
    unsigned long long mx = 0ull;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const double c = msig[m] * beta[m];
        const double d = mave[m] * c;
        const unsigned long long bc = (unsigned long long)__double_as_longlong(fabs(c)), bd = (unsigned long long)__double_as_longlong(fabs(d));
        mx = bc > mx ? bc : mx;
        mx = bd > mx ? bd : mx;                    // |x| as bits: monotone for finite values; NaN / Inf end up on top
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long v = ((unsigned long long)(unsigned)__shfl_xor((int)(mx >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)mx, o, 64);
        mx = v > mx ? v : mx;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, mx);
}
// plane byte of marker mu (within the padded block) for operand A: super-step ss = mu >> 8, then [kg][field i][dword j][byte b]
// with mu & 255 = 64 kg + 16 j + 4 b + i
__device__ __forceinline__ size_t pg_plane_index(size_t mu) {
    const size_t r = mu & 255;
    return (mu & ~(size_t)255) + (r & 192) + ((r & 3) << 4) + (((r >> 4) & 3) << 2) + ((r >> 2) & 3);
}
__global__ void k_pg_planes(const double* __restrict__ mave, const double* __restrict__ msig, const double* __restrict__ beta,
                            int M, size_t Mpad, const unsigned long long* __restrict__ maxbits, uint8_t* __restrict__ planes,
                            long long* __restrict__ dsum, int dirty) {
    const size_t mu = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mu >= Mpad) return;
    const int sh = assoc_shift(*maxbits);
    long long d1 = 0, d2 = 0;
    uint32_t z1 = 0, z2 = 0, y1 = 0, y2 = 0;
    const bool nonfinite = ((*maxbits >> 52) & 0x7ffull) == 0x7ffull;
    if (mu < (size_t)M && !nonfinite) {
        const double c = msig[mu] * beta[mu];
        const double d = mave[mu] * c;
        double q1, q2;
        split2(__builtin_ldexp(c, sh), q1, q2);
        z1 = ((uint32_t)(int)(q1 * 0x1p22) + 0x00808080u) ^ 0x00808080u;           // signed base-256 digits
        z2 = ((uint32_t)(int)(q2 * 0x1p53) + 0x00808080u) ^ 0x00808080u;
        split2(__builtin_ldexp(d, sh), q1, q2);
        d1 = (long long)(q1 * 0x1p22);
        d2 = (long long)(q2 * 0x1p53);
        y1 = ((uint32_t)(int)d1 + 0x00808080u) ^ 0x00808080u;
        y2 = ((uint32_t)(int)d2 + 0x00808080u) ^ 0x00808080u;
    }
    const size_t at = pg_plane_index(mu);
#pragma unroll
    for (int pl = 0; pl < 4; pl++) {
        planes[(size_t)pl * Mpad + at] = (uint8_t)(z1 >> (8 * pl));
        planes[(size_t)(pl + 4) * Mpad + at] = (uint8_t)(z2 >> (8 * pl));
        if (dirty) {
            planes[(size_t)(pl + 8) * Mpad + at] = (uint8_t)(y1 >> (8 * pl));
            planes[(size_t)(pl + 12) * Mpad + at] = (uint8_t)(y2 >> (8 * pl));
        }
    }
    // sum of D over the block, exact: the two parts as integers (wavefront sum, then one atomic per wavefront)
    for (int o = 32; o >= 1; o >>= 1) {
        d1 += ((long long)__shfl_xor((int)(d1 >> 32), o, 64) << 32) + (long long)(unsigned)__shfl_xor((int)d1, o, 64);
        d2 += ((long long)__shfl_xor((int)(d2 >> 32), o, 64) << 32) + (long long)(unsigned)__shfl_xor((int)d2, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && (d1 | d2) != 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(dsum), (unsigned long long)d1);
        atomicAdd(reinterpret_cast<unsigned long long*>(dsum) + 1, (unsigned long long)d2);
    }
}

// 16 x 16 transpose of 2-bit fields across the 16 lanes of a DPP row: in: lane r holds x_r (field f = element (r, f)); out:
// lane n holds element (f, n) in field f.  Stage d = 8, 4, 2, 1 swaps the off-diagonal d x d blocks with lane r ^ d.
template <int CTRL> __device__ __forceinline__ uint32_t pg_dpp(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xf, 0xf, true);          // every lane of a row has a source: no "old" value needed
}
struct PgLane { uint32_t sel8, sel4, keep2, keep1, rot2, rot1; };
__device__ __forceinline__ PgLane pg_lane_consts(int r) {
    PgLane c;
    // stages 8 and 4 move whole bytes (4 fields): one v_perm_b32 of {partner, own} with a per-lane selector
    c.sel8 = (r & 8) ? 0x03020706u : 0x05040100u;   // hi lanes: {p.b2, p.b3, x.b2, x.b3}; lo lanes: {x.b0, x.b1, p.b0, p.b1}
    c.sel4 = (r & 4) ? 0x03070105u : 0x06020400u;   // hi lanes: {p.b1, x.b1, p.b3, x.b3}; lo lanes: {x.b0, p.b0, x.b2, p.b2}
    c.keep2 = (r & 2) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;  c.rot2 = (r & 2) ? 28 : 4;         // rotl by 4 (lo lanes) / rotr by 4 (hi lanes)
    c.keep1 = (r & 1) ? 0xCCCCCCCCu : 0x33333333u;  c.rot1 = (r & 1) ? 30 : 2;
    return c;
}
__device__ __forceinline__ uint32_t pg_rotl(uint32_t x, uint32_t k) { return __builtin_amdgcn_alignbit(x, x, 32u - k); }
__device__ __forceinline__ uint32_t pg_bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
__device__ __forceinline__ uint32_t pg_transpose16(uint32_t x, const PgLane& c) {
    uint32_t p;
    p = pg_dpp<0x141>(pg_dpp<0x140>(x));                  // lane ^ 15 then lane ^ 7: lane ^ 8
    x = __builtin_amdgcn_perm(p, x, c.sel8);
    p = pg_dpp<0x1B>(pg_dpp<0x141>(x));                   // lane ^ 7 then lane ^ 3: lane ^ 4
    x = __builtin_amdgcn_perm(p, x, c.sel4);
    p = pg_dpp<0x4E>(x);                                  // lane ^ 2
    x = pg_bfi(c.keep2, x, pg_rotl(p, c.rot2));
    p = pg_dpp<0xB1>(x);                                  // lane ^ 1
    x = pg_bfi(c.keep1, x, pg_rotl(p, c.rot1));
    return x;
}
// .bed code -> genotype value in the 2-bit field (00 -> 2, 10 -> 1, 11 -> 0, 01 (missing) -> 3), as sweep.hip's ring
__device__ __forceinline__ uint32_t pg_recode(uint32_t w) { return w; }   // (the block is stored in the device code since round 4)

template <bool DIRTY>
__global__ __launch_bounds__(256) void k_pg_mfma(const uint8_t* __restrict__ bed, size_t stride, int M, const uint8_t* __restrict__ planes,
                                                 size_t Mpad, unsigned long long* __restrict__ gacc) {
    constexpr int NPL = DIRTY ? 16 : 8;             // digit planes staged per marker: C (and D)
    __shared__ __attribute__((aligned(16))) uint8_t s_pl[NPL * PG_KB];
    // The workgroup's 64 bytes (256 individuals) of the 256 columns of a super-step, staged through LDS: the loads are issued so
    // that four lanes cover one column's 64 bytes (16 whole sectors per wave instruction; one lane per column and wavefront --
    // 64 quarter sectors per instruction -- ran at 2.6 TB/s), the consumers pick their wavefront's 16-byte piece of their column.
    // Piece p of column c sits at c * 64 + 16 * (p ^ ((c >> 2) & 3)): the 16 lanes of a row read 16 different bank groups.
    __shared__ __attribute__((aligned(16))) uint8_t s_g[2][256 * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    // Workgroups are dispatched round-robin over the 8 XCDs (each with an L2 of its own, 128-byte lines): workgroups that
    // read neighbouring 64-byte pieces of the same columns must share an XCD, or every line is fetched by two L2s.  Dispatch
    // index bid = 8 a + c (c = XCD): the eight pieces 8 (8 (a >> 3) + c) + (a & 7) of one 512-byte run go to XCD c, 8 launches apart.
    const unsigned bid = blockIdx.x, xa = bid >> 3, xc = bid & 7u;
    const size_t piece = 8 * (8 * (size_t)(xa >> 3) + xc) + (xa & 7u);         // gridDim.x is a multiple of 64
    const size_t ibyte = piece * 64 + (size_t)wave * 16;                        // this wavefront's 16 bytes (64 individuals) of every column
    const bool live = ibyte < stride;                                            // the stride is a multiple of 16
    const size_t mstart = (size_t)blockIdx.y * PG_STAGES * PG_KB;
    const PgLane lc = pg_lane_consts(n);
    pg_v4i acc0[4], acc1[4], acc2[4];
    pg_v4i zcc0[DIRTY ? 4 : 1], zcc1[DIRTY ? 4 : 1], zcc2[DIRTY ? 4 : 1];
#pragma unroll
    for (int t = 0; t < 4; t++) { acc0[t] = pg_v4i{0, 0, 0, 0}; acc1[t] = pg_v4i{0, 0, 0, 0}; acc2[t] = pg_v4i{0, 0, 0, 0}; }
#pragma unroll
    for (int t = 0; t < (DIRTY ? 4 : 1); t++) { zcc0[t] = pg_v4i{0, 0, 0, 0}; zcc1[t] = pg_v4i{0, 0, 0, 0}; zcc2[t] = pg_v4i{0, 0, 0, 0}; }
    constexpr uint32_t M0 = 0x03030303u, M1 = 0x01010101u;
    for (int st = 0; st < PG_STAGES; st++) {
        const size_t mb = mstart + (size_t)st * PG_KB;
        if (mb >= (size_t)M) break;                                               // (uniform)
        __syncthreads();                                                          // the previous stage's planes are no longer read
        // NPL planes x PG_KB bytes of this stage -> LDS (64 or 128 bytes per thread)
#pragma unroll
        for (int q = 0; q < NPL / 2; q++) {
            const int off = (q * 256 + tid) * 16;                                 // byte offset within the NPL x PG_KB block, plane-major
            const int pl = off / PG_KB, in = off % PG_KB;
            *reinterpret_cast<uint4*>(s_pl + off) = *reinterpret_cast<const uint4*>(planes + (size_t)pl * Mpad + mb + in);
        }
        __syncthreads();
        // the column bytes of a super-step are requested one super-step ahead and parked in the other LDS buffer at its end
        const int lpiece = lane & 3;
        const bool lin = piece * 64 + 16 * (size_t)lpiece < stride;
        auto request = [&](int ss, uint4 (&dst)[4]) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = 64 * wave + 16 * q + (lane >> 2);
                const size_t mu = mb + (size_t)ss * 256 + (size_t)c;
                dst[q] = (lin && mu < (size_t)M) ? *reinterpret_cast<const uint4*>(bed + mu * stride + piece * 64 + 16 * (size_t)lpiece)
                                                 : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // code 11: a = 0
            }
        };
        auto park = [&](int buf, const uint4 (&src)[4]) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = 64 * wave + 16 * q + (lane >> 2);
                *reinterpret_cast<uint4*>(s_g[buf] + c * 64 + 16 * (lpiece ^ ((c >> 2) & 3))) = src[q];
            }
        };
        uint4 wn[4];
        request(0, wn);
        park(0, wn);
        __syncthreads();
#pragma unroll 1
        for (int ss = 0; ss < PG_KB / 256; ss++) {
            if (ss + 1 < PG_KB / 256) request(ss + 1, wn);
            uint4 w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = 64 * kg + 16 * j + n;
                w[j] = *reinterpret_cast<const uint4*>(s_g[ss & 1] + c * 64 + 16 * (wave ^ ((c >> 2) & 3)));
            }
            // operand A: plane row n (clean blocks: rows 8..15 repeat 0..7 and their outputs are dropped), this lane group's 16 markers,
            // one chunk per field
            const uint8_t* ab = s_pl + (DIRTY ? n : (n & 7)) * PG_KB + ss * 256 + kg * 64;
            const pg_v4i a0 = *reinterpret_cast<const pg_v4i*>(ab), a1 = *reinterpret_cast<const pg_v4i*>(ab + 16),
                         a2 = *reinterpret_cast<const pg_v4i*>(ab + 32), a3 = *reinterpret_cast<const pg_v4i*>(ab + 48);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                uint32_t y[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t x = t == 0 ? w[j].x : (t == 1 ? w[j].y : (t == 2 ? w[j].z : w[j].w));
                    // (clean blocks: a field that reads 3 -- missing genotype -- can only belong to an individual without a phenotype:
                    // it spoils that individual's own sums, which k_pg_finish discards)
                    y[j] = pg_transpose16(pg_recode(x), lc);
                }
                if constexpr (DIRTY) {
                    uint32_t z[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        z[j] = y[j] & (y[j] >> 1) & 0x55555555u;                    // 1 in the low bit of every field that reads 3
                        y[j] &= ~(z[j] | (z[j] << 1));                               // a' = 0 there
                    }
                    const pg_v4i z0 = {(int)(z[0] & M1), (int)(z[1] & M1), (int)(z[2] & M1), (int)(z[3] & M1)};
                    const pg_v4i z1 = {(int)(z[0] & (M1 << 2)), (int)(z[1] & (M1 << 2)), (int)(z[2] & (M1 << 2)), (int)(z[3] & (M1 << 2))};
                    const pg_v4i z2 = {(int)(z[0] & (M1 << 4)), (int)(z[1] & (M1 << 4)), (int)(z[2] & (M1 << 4)), (int)(z[3] & (M1 << 4))};
                    const pg_v4i z3 = {(int)((z[0] >> 6) & M1), (int)((z[1] >> 6) & M1), (int)((z[2] >> 6) & M1), (int)((z[3] >> 6) & M1)};
                    zcc0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, z0, zcc0[t], 0, 0, 0);
                    zcc1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, z1, zcc1[t], 0, 0, 0);
                    zcc2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, z2, zcc2[t], 0, 0, 0);
                    zcc0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, z3, zcc0[t], 0, 0, 0);
                }
                const pg_v4i b0 = {(int)(y[0] & M0), (int)(y[1] & M0), (int)(y[2] & M0), (int)(y[3] & M0)};
                const pg_v4i b1 = {(int)(y[0] & (M0 << 2)), (int)(y[1] & (M0 << 2)), (int)(y[2] & (M0 << 2)), (int)(y[3] & (M0 << 2))};
                const pg_v4i b2 = {(int)(y[0] & (M0 << 4)), (int)(y[1] & (M0 << 4)), (int)(y[2] & (M0 << 4)), (int)(y[3] & (M0 << 4))};
                const pg_v4i b3 = {(int)((y[0] >> 6) & M0), (int)((y[1] >> 6) & M0), (int)((y[2] >> 6) & M0), (int)((y[3] >> 6) & M0)};
                acc0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc1[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, acc2[t], 0, 0, 0);
                acc0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b3, acc0[t], 0, 0, 0);
            }
            if (ss + 1 < PG_KB / 256) park((ss + 1) & 1, wn);
            __syncthreads();
        }
    }
    // C: column n = individual of the tile, rows 4 kg + r = planes.  kg = 0: the four digits of part 1 of C, kg = 1: of part 2;
    // DIRTY: kg = 2, 3: the same of D, from the indicator set.
    if (live && (DIRTY || kg < 2)) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            long long sx = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int v = acc0[t][r] + (acc1[t][r] >> 2) + (acc2[t][r] >> 4);
                if constexpr (DIRTY) {
                    const int vz = zcc0[t][r] + (zcc1[t][r] >> 2) + (zcc2[t][r] >> 4);
                    v = kg < 2 ? v : vz;
                }
                sx += (long long)v << (8 * r);
            }
            const size_t ind = 4 * ibyte + 16 * (size_t)t + (size_t)n;
            if (sx != 0) atomicAdd(gacc + 2 * ind + (kg & 1), (unsigned long long)sx);
        }
    }
}
__global__ void k_pg_finish(const unsigned long long* __restrict__ gacc, const long long* __restrict__ dsum, const uint8_t* __restrict__ namask2,
                            size_t n4, const unsigned long long* __restrict__ maxbits, double* __restrict__ g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const bool present = ((namask2[i >> 2] >> (2 * (i & 3))) & 3u) == 3u;
    const int sh = assoc_shift(*maxbits);
    const bool nonfinite = ((*maxbits >> 52) & 0x7ffull) == 0x7ffull;
    const long long t1 = (long long)gacc[2 * i] - dsum[0], t2 = (long long)gacc[2 * i + 1] - dsum[1];
    const double v = __builtin_ldexp((double)t1 * 0x1p-22 + (double)t2 * 0x1p-53, -sh);
    g[i] = !present ? 0.0 : (nonfinite ? __longlong_as_double(0x7ff8000000000000ll) : v);
}
size_t predict_workspace_bytes(size_t stride, int M) {
    const size_t Mpad = ((size_t)(M > 0 ? M : 1) + PG_KB - 1) / PG_KB * PG_KB;
    return 16 * Mpad + 16 * 4 * stride + 64;
}
// ws: predict_workspace_bytes(stride, M) bytes.  dirty = 0 only if every marker of the block is free of missing genotypes among
// the phenotyped individuals (the caller checks the marker statistics' flags); g receives 4 * stride doubles.
hipError_t launch_predict_g_mfma(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, const double* mave,
                                 const double* msig, const double* beta, double* g, void* ws, int dirty, hipStream_t st) {
    if (stride == 0 || M <= 0) return hipSuccess;
    const size_t Mpad = ((size_t)M + PG_KB - 1) / PG_KB * PG_KB;
    uint8_t* planes = static_cast<uint8_t*>(ws);
    unsigned long long* gacc = reinterpret_cast<unsigned long long*>(planes + 16 * Mpad);
    unsigned long long* maxbits = gacc + 2 * 4 * stride;
    long long* dsum = reinterpret_cast<long long*>(maxbits + 2);
    hipError_t e = hipMemsetAsync(gacc, 0, 16 * 4 * stride + 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pg_absmax, dim3(256), dim3(256), 0, st, mave, msig, beta, M, maxbits);
    hipLaunchKernelGGL(k_pg_planes, dim3((unsigned)((Mpad + 255) / 256)), dim3(256), 0, st, mave, msig, beta, M, Mpad, maxbits, planes, dsum, dirty);
    const unsigned gx = (unsigned)((stride + 63) / 64 + 63) / 64 * 64, gy = (unsigned)(((size_t)M + (size_t)PG_STAGES * PG_KB - 1) / ((size_t)PG_STAGES * PG_KB));
    if (dirty) hipLaunchKernelGGL(k_pg_mfma<true>, dim3(gx, gy), dim3(256), 0, st, bed, stride, M, planes, Mpad, gacc);
    else hipLaunchKernelGGL(k_pg_mfma<false>, dim3(gx, gy), dim3(256), 0, st, bed, stride, M, planes, Mpad, gacc);
    hipLaunchKernelGGL(k_pg_finish, dim3((unsigned)((4 * stride + 255) / 256)), dim3(256), 0, st, gacc, dsum, namask2, 4 * stride, maxbits, g);
    return hipGetLastError();
}

hipError_t launch_predict_g(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, const double* mave,
                            const double* msig, const double* beta, double* g, hipStream_t st) {
    if (stride == 0 || M <= 0) return hipSuccess;
    // two bytes per thread only when that alone fills the SIMDs twice over (256 CUs x 4 SIMDs x 2 wavefronts)
    if (stride / 2 >= (size_t)256 * 4 * 2 * 64)
        hipLaunchKernelGGL(k_predict_g<2>, dim3((unsigned)((stride / 2 + 255) / 256)), dim3(256), 0, st, bed, namask2, stride, M, mave, msig, beta, g);
    else
        hipLaunchKernelGGL(k_predict_g<1>, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, st, bed, namask2, stride, M, mave, msig, beta, g);
    return hipGetLastError();
}
size_t assoc_workspace_bytes(size_t stride) {
    const size_t npad = (4 * stride + 4 * AS_BLK - 1) / (4 * AS_BLK) * (4 * AS_BLK);
    return 8 * npad + 64;
}
// ws: assoc_workspace_bytes(stride) bytes of device memory (digit planes of y, the scale)
hipError_t launch_assoc(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, const double* y,
                        double* xtx, double* xty, void* ws, hipStream_t st) {
    if (M <= 0 || stride == 0) return hipSuccess;
    const size_t npad = (4 * stride + 4 * AS_BLK - 1) / (4 * AS_BLK) * (4 * AS_BLK);
    uint8_t* planes = static_cast<uint8_t*>(ws);
    unsigned long long* maxbits = reinterpret_cast<unsigned long long*>(planes + 8 * npad);
    hipError_t e = hipMemsetAsync(maxbits, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, y, namask2, 4 * stride, maxbits);
    hipLaunchKernelGGL(k_yplanes, dim3((unsigned)((npad / 4 + 255) / 256)), dim3(256), 0, st, y, namask2, stride, maxbits, planes, npad);
    hipLaunchKernelGGL(k_assoc_mfma, dim3((unsigned)((M + AS_MB - 1) / AS_MB)), dim3(256), 0, st, bed, namask2, stride, M,
                       planes, npad, maxbits, xtx, xty);
    return hipGetLastError();
}

}  // namespace gm

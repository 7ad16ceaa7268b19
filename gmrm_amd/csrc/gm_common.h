// gm_common.h -- arithmetic building blocks shared by the host C++ and the HIP device
// code of libgmrm_hip: the 2-level pre-rounding that makes every reduction on the path
// order-independent, the path's exp(), and the 2-bit genotype decode rule.
//
// Everything here is a fixed sequence of IEEE-754 binary64 operations (add, mul, fma,
// div, sqrt) so that a CPU and a gfx950 evaluation give the same bits.  The library is
// built with -ffp-contract=off: an a*b+c written as two operations stays two roundings.
#pragma once
#include <cstdint>
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline
#endif

namespace gm {

// ---- genotype decode: reference src/dotp_lut.hpp (generator src/lut/mk_lut.cpp:25-33,54-62)
// 2-bit code c of individual k in a byte = (byte >> 2k) & 3:
//   c = 0 -> (a,b) = (2,1);  c = 1 -> (0,0) missing;  c = 2 -> (1,1);  c = 3 -> (0,1)
GM_HD double code_a(int c) { return c == 0 ? 2.0 : (c == 2 ? 1.0 : 0.0); }
GM_HD double code_b(int c) { return c == 1 ? 0.0 : 1.0; }

// ---- the genotype block as the DEVICE holds it ("device code", round 4) ----------------------------------
// HBM keeps the .bed layout (marker-major columns, four individuals per byte, LSB first) but the 2-bit field is
// recoded once, when a block arrives (gmrm_upload_bed / gmrm_load_bed_file / gmrm_synth_bed), so that it IS the
// reference's dotp_lut_a value:
//   .bed 00 (a = 2, b = 1) -> 10      .bed 10 (a = 1, b = 1) -> 01      .bed 11 (a = 0, b = 1) -> 00      .bed 01 (missing) -> 11
// Every kernel reads this code: the sweep kernel's loaders then move column slices HBM -> LDS with no register
// pass in between (global_load_lds) and feed the matrix cores from registers as they land; the per-call kernels index
// their (a, b) table with it.  gmrm_download_bed converts back.  A missing genotype is field value 3; an individual
// without a phenotype is forced to 3 by OR-ing the inverted NA mask (namask2: 11 = phenotype present).
GM_HD uint32_t bed_to_dcode(uint32_t w) { return (~w & 0xAAAAAAAAu) | ((w ^ (w >> 1)) & 0x55555555u); }
GM_HD uint32_t dcode_to_bed(uint32_t d) { return (~d & 0xAAAAAAAAu) | ((~(d >> 1) ^ d) & 0x55555555u); }
GM_HD double dcode_a(int c) { return c == 3 ? 0.0 : (double)c; }
GM_HD double dcode_b(int c) { return c == 3 ? 0.0 : 1.0; }

// ---- order-independent summation -------------------------------------------------
// split2(x): q1 = x rounded to a multiple of 2^-22, q2 = (x - q1) rounded to a multiple
// of 2^-53, remainder (< 2^-54) dropped.  For |x| < 2^8 and <= 2^22 terms scaled by
// a in {0,1,2}, every partial sum of a*q1 (and of a*q2) is exactly representable, so the
// totals do not depend on the order of accumulation (thread, wavefront, workgroup, GPU).
constexpr double SPLIT_C1 = 0x1.8p+30;   // 1.5 * 2^(52-22)
constexpr double SPLIT_C2 = 0x1.8p-1;    // 1.5 * 2^(52-53)
constexpr double SPLIT_S1 = 0x1.8p+38;   // squares (< 2^16): grid 2^-14
constexpr double SPLIT_S2 = 0x1.8p+7;    //                   grid 2^-45
constexpr double EPS_ABS_LIMIT = 256.0;  // |residual| bound the exactness argument needs
constexpr int    MAX_LOG2_N = 22;        // individuals <= 2^22

GM_HD void split2(double x, double& q1, double& q2) {
    const double t = x + SPLIT_C1;
    q1 = t - SPLIT_C1;
    const double r = x - q1;
    const double u = r + SPLIT_C2;
    q2 = u - SPLIT_C2;
}
GM_HD void split2sq(double x, double& q1, double& q2) {
    const double t = x + SPLIT_S1;
    q1 = t - SPLIT_S1;
    const double r = x - q1;
    const double u = r + SPLIT_S2;
    q2 = u - SPLIT_S2;
}

// ---- the fixed-point residual ----------------------------------------------------------
// The residual lives on the grid 2^-44: every value is k * 2^-44 with |k| < 2^52 (|eps| < 2^8), an exactly
// representable double, and every update ADDS a grid value -- eps + v is exact, no rounding per individual --
// so that a dot product after an update equals the dot product before it plus an integer-linear correction:
// the sweep kernel walks past a marker whose effect changes without recomputing the dots behind it
// (sweep.hip, "continuation").  The update values of one marker are linear in the genotype value a:
//   v(a) = beta_ + a * alpha_,   alpha_ = grid(bs_),  beta_ = grid(mdb * bs_)
// with bs_ = dbeta * msig and mdb = -mave of phenotype.cpp:328-329; the reference evaluates
// fl(fl(mdb * b + a) * bs_), the same real number to ~1 ulp (|difference| <= 2^-44 per individual and update).
constexpr double GRID     = 0x1p-44;
constexpr double GRID_INV = 0x1p+44;
GM_HD double grid(double x) { return __builtin_rint(x * GRID_INV) * GRID; }   // nearest multiple, ties to even (v_rndne_f64 / rint)
// v[c] for the four .bed codes c (00: a = 2, 01: missing, 10: a = 1, 11: a = 0), from {dbeta, mave, msig}
GM_HD void update_values(double dbeta, double mave, double msig, double& alpha_, double& beta_) {
    const double bs_ = dbeta * msig;             // phenotype.cpp:328
    const double mdb = -mave;                    // phenotype.cpp:329
    alpha_ = grid(bs_);
    beta_ = grid(mdb * bs_);
}

GM_HD double fma_(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}

// ---- the path's exp() --------------------------------------------------------------
// Used where the reference calls exp() in the Gibbs step (bayes.cpp:441,472) and in the
// ziggurat wedge tests.  Argument reduction by ln2 (hi/lo), degree-13 Taylor/Horner in
// fma, scaling by two exact powers of two.  ~1 ulp.
GM_HD double exp_(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // same values as the branching form below, written with selects (no divergent branches)
    const bool is_nan = x != x;
    const bool over = x > 0x1.62e42fefa39efp+9;
    const bool under = x < -0x1.74910d52d3052p+9;
    const double xc = (is_nan || over || under) ? 0.0 : x;
#else
    if (x != x) return x;
    if (x > 0x1.62e42fefa39efp+9) return __builtin_huge_val();
    if (x < -0x1.74910d52d3052p+9) return 0.0;
    const double xc = x;
#endif
    const double t = xc * 0x1.71547652b82fep+0;
    const double kd = (t + 0x1.8p52) - 0x1.8p52;
    double r = fma_(-kd, 0x1.62e42fee00000p-1, xc);
    r = fma_(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = fma_(p, r, 0x1.1eed8eff8d898p-29);
    p = fma_(p, r, 0x1.ae64567f544e4p-26);
    p = fma_(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma_(p, r, 0x1.71de3a556c734p-19);
    p = fma_(p, r, 0x1.a01a01a01a01ap-16);
    p = fma_(p, r, 0x1.a01a01a01a01ap-13);
    p = fma_(p, r, 0x1.6c16c16c16c17p-10);
    p = fma_(p, r, 0x1.1111111111111p-7);
    p = fma_(p, r, 0x1.5555555555555p-5);
    p = fma_(p, r, 0x1.5555555555555p-3);
    p = fma_(p, r, 0x1.0000000000000p-1);
    p = fma_(p, r, 1.0);
    p = fma_(p, r, 1.0);
    const int k = (int)kd;
    const int k1 = k / 2, k2 = k - k1;
    const double s1 = __builtin_bit_cast(double, (uint64_t)(k1 + 1023) << 52);
    const double s2 = __builtin_bit_cast(double, (uint64_t)(k2 + 1023) << 52);
    const double v = (p * s1) * s2;
#if defined(__HIP_DEVICE_COMPILE__)
    return is_nan ? x : (over ? __builtin_huge_val() : (under ? 0.0 : v));
#else
    return v;
#endif
}

GM_HD uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
GM_HD uint32_t mt_twist1(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    uint32_t v = far ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    return v;
}

}  // namespace gm

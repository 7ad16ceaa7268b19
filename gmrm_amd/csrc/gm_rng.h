// gm_rng.h -- the RNG spec of the build (host + device).
//
// The reference draws through Boost.Random (src/distributions.hpp:5-61, boost 1.76 per
// setup/Make.intel_ioampi:6), which is not vendored in the reference and not installed in
// this image.  This header restates Boost's published algorithms; DESIGN.md "RNG spec"
// lists each one and marks the layer PARITY UNPINNED (no reference fixture pins a draw).
//
//   mt19937                      32-bit Mersenne Twister, boost::mt19937(seed)
//   uniform_real(0,1)/uniform_01 one output * 2^-32               (distributions.hpp:55-59)
//   normal(mean, sqrt(var))      128-layer ziggurat               (distributions.hpp:48-53)
//   exponential                  256-layer ziggurat (tail = shifted restart)
//   gamma(shape, scale)          shape==1: exponential; >1: tan/Cheng rejection; <1: GS
//   beta(a,b)                    X/(X+Y)                          (distributions.hpp:39-46)
//   inv_scaled_chisq(a,b)        1/gamma(a/2, 2/(a*b))            (distributions.hpp:24-30)
//   shuffle                      std::random_shuffle(first,last,uniform_int generator)
//                                                                 (phenotype.cpp:314-323)
// The ziggurat draws are templates over a "word source" (anything with u32()) so the same
// code serves the host engine and the sweep kernel's LDS-resident stream.
#pragma once
#include "gm_common.h"
#include "zig_tables.h"

namespace gm {

#if defined(__HIP_DEVICE_COMPILE__)
// device copies of the layer tables (constant address space)
static __device__ __constant__ double d_zig_norm_x[129] = { GM_ZIG_NORM_X_VALUES };
static __device__ __constant__ double d_zig_norm_y[129] = { GM_ZIG_NORM_Y_VALUES };
static __device__ __constant__ double d_zig_exp_x[257] = { GM_ZIG_EXP_X_VALUES };
static __device__ __constant__ double d_zig_exp_y[257] = { GM_ZIG_EXP_Y_VALUES };
#define GM_ZNX d_zig_norm_x
#define GM_ZNY d_zig_norm_y
#define GM_ZEX d_zig_exp_x
#define GM_ZEY d_zig_exp_y
#else
#define GM_ZNX gm_zig_norm_x
#define GM_ZNY gm_zig_norm_y
#define GM_ZEX gm_zig_exp_x
#define GM_ZEY gm_zig_exp_y
#endif

template <class Src> GM_HD double u01(Src& s) {
    for (;;) {
        const double r = (double)s.u32() * (1.0 / 4294967296.0);
        if (r < 1.0) return r;
    }
}
// uniform_real_distribution<double>(0,1): numerator / 2^32 * (1-0) + 0
template <class Src> GM_HD double unif(Src& s) {
    for (;;) {
        const double r = (double)s.u32() / 4294967296.0 * (1.0 - 0.0) + 0.0;
        if (r < 1.0) return r;
    }
}
GM_HD double unif_from_word(uint32_t w) { return (double)w / 4294967296.0 * (1.0 - 0.0) + 0.0; }

// 8 bucket bits + 53 fraction bits out of two 32-bit outputs
template <class Src> GM_HD double int_float_pair(Src& s, int& bucket) {
    const uint32_t u1 = s.u32();
    bucket = (int)(u1 & 0xFFu);
    double x = (double)(u1 >> 8) * (1.0 / 16777216.0);
    const uint32_t u2 = s.u32();
    x += (double)(u2 & 0x1FFFFFFFu);
    x *= (1.0 / 536870912.0);
    return x;
}

template <class Src> GM_HD double unit_exponential(Src& s) {
    double shift = 0.0;
    for (;;) {
        int i;
        const double x = int_float_pair(s, i) * GM_ZEX[i];
        if (x < GM_ZEX[i + 1]) return shift + x;
        if (i == 0) { shift += GM_ZEX[1]; continue; }
        const double y01 = u01(s);
        const double y = GM_ZEY[i] + y01 * (GM_ZEY[i + 1] - GM_ZEY[i]);
        const double y_above_ubound = (GM_ZEX[i] - GM_ZEX[i + 1]) * y01 - (GM_ZEX[i] - x);
        const double y_above_lbound = y - (GM_ZEY[i + 1] + (GM_ZEX[i + 1] - x) * GM_ZEY[i + 1]);
        if (y_above_ubound < 0.0 && (y_above_lbound < 0.0 || y < exp_(-x))) return x + shift;
    }
}

template <class Src> GM_HD double unit_normal(Src& s) {
    for (;;) {
        int b;
        const double x01 = int_float_pair(s, b);
        const int sign = (b & 1) * 2 - 1;
        const int i = b >> 1;
        const double x = x01 * GM_ZNX[i];
        if (x < GM_ZNX[i + 1]) return x * sign;
        if (i == 0) {
            const double tail_start = GM_ZNX[1];
            for (;;) {
                const double xx = unit_exponential(s) / tail_start;
                const double yy = unit_exponential(s);
                if (2.0 * yy > xx * xx) return (xx + tail_start) * sign;
            }
        }
        const double y01 = u01(s);
        const double y = GM_ZNY[i] + y01 * (GM_ZNY[i + 1] - GM_ZNY[i]);
        double y_above_ubound, y_above_lbound;
        if (GM_ZNX[i] >= 1.0) {
            y_above_ubound = (GM_ZNX[i] - GM_ZNX[i + 1]) * y01 - (GM_ZNX[i] - x);
            y_above_lbound = y - (GM_ZNY[i] + (GM_ZNX[i] - x) * GM_ZNY[i] * GM_ZNX[i]);
        } else {
            y_above_lbound = (GM_ZNX[i] - GM_ZNX[i + 1]) * y01 - (GM_ZNX[i] - x);
            y_above_ubound = y - (GM_ZNY[i] + (GM_ZNX[i] - x) * GM_ZNY[i] * GM_ZNX[i]);
        }
        if (y_above_ubound < 0.0 && (y_above_lbound < 0.0 || y < exp_(-(x * x / 2.0)))) return x * sign;
    }
}

// normal_distribution(mean, sigma)(eng) = unit * sigma + mean, sigma = sqrt(variance)
template <class Src> GM_HD double norm(Src& s, double mean, double sigma2) {
    const double sigma = __builtin_sqrt(sigma2);
    return unit_normal(s) * sigma + mean;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host engine --------------------------------------------------------------------
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    void seed(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void twist() {
        for (int i = 0; i < 624; i++) mt[i] = mt_twist1(mt[i], mt[(i + 1) % 624], mt[(i + 397) % 624]);
        idx = 0;
    }
    uint32_t u32() {
        if (idx >= 624) twist();
        return mt_temper(mt[idx++]);
    }
};

inline double rgamma(Mt19937& e, double alpha, double beta) {
    if (alpha == 1.0) return unit_exponential(e) * beta;
    if (alpha > 1.0) {
        const double pi = 3.14159265358979323846;
        for (;;) {
            const double y = std::tan(pi * u01(e));
            const double x = std::sqrt(2.0 * alpha - 1.0) * y + alpha - 1.0;
            if (x <= 0.0) continue;
            if (u01(e) > (1.0 + y * y) * std::exp((alpha - 1.0) * std::log(x / (alpha - 1.0))
                                                  - std::sqrt(2.0 * alpha - 1.0) * y))
                continue;
            return x * beta;
        }
    }
    const double p = std::exp(1.0) / (alpha + std::exp(1.0));
    for (;;) {
        const double u = u01(e);
        const double y = unit_exponential(e);
        double x, q;
        if (u < p) { x = std::exp(-y / alpha); q = p * std::exp(-x); }
        else       { x = 1.0 + y;              q = p + (1.0 - p) * std::pow(x, alpha - 1.0); }
        if (u >= q) continue;
        return x * beta;
    }
}
inline double rbeta(Mt19937& e, double a, double b) {
    const double x = rgamma(e, a, 1.0);
    const double y = rgamma(e, b, 1.0);
    return x / (x + y);
}
inline double inv_scaled_chisq(Mt19937& e, double a, double b) {
    const double ga = 0.5 * a, gb = 0.5 * a * b;
    return 1.0 / rgamma(e, ga, 1.0 / gb);
}
inline uint32_t uniform_int(Mt19937& e, uint32_t n) {   // [0, n-1]
    const uint32_t range = n - 1;
    if (range == 0) return 0;
    uint32_t bucket = 0xFFFFFFFFu / (range + 1u);
    if (0xFFFFFFFFu % (range + 1u) == range) ++bucket;
    for (;;) {
        const uint32_t r = e.u32() / bucket;
        if (r <= range) return r;
    }
}
inline void shuffle(Mt19937& e, int* v, int n) {
    for (int i = 1; i < n; i++) {
        const int j = (int)uniform_int(e, (uint32_t)i + 1u);
        if (i != j) { const int t = v[i]; v[i] = v[j]; v[j] = t; }
    }
}
#endif

}  // namespace gm

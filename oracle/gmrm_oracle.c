/*
 * oracle/gmrm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See gmrm_oracle.h for scope, pinning status and the two summation modes.
 * Every function cites the reference source it restates (paths relative to
 * /root/reference/).  Build: oracle/Makefile (strict IEEE: -O2 -ffp-contract=off).
 */
#include "gmrm_oracle.h"
#include "zig_tables.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KMAX 32

/* ------------------------------------------------------------------------- */
/* Lookup tables.  src/dotp_lut.hpp:3-2055, src/na_lut.hpp:3-68; rule from    */
/* src/lut/mk_lut.cpp:25-33,54-62 and src/lut/mk_lut_na.cpp:24-31.            */
/* 2-bit code c = (byte >> 2k) & 3 of individual k (LSB first):               */
/*   00 -> a=2,b=1   01 -> a=0,b=0 (missing)   10 -> a=1,b=1   11 -> a=0,b=1  */
/* ------------------------------------------------------------------------- */
static double g_lut_a[1024], g_lut_b[1024], g_lut_na[64];
static int g_luts_ready = 0;

static void build_luts(void) {
    if (g_luts_ready) return;
    for (int byte = 0; byte < 256; byte++)
        for (int k = 0; k < 4; k++) {
            int c = (byte >> (2 * k)) & 3;
            g_lut_a[byte * 4 + k] = (c == 0) ? 2.0 : (c == 2) ? 1.0 : 0.0;
            g_lut_b[byte * 4 + k] = (c == 1) ? 0.0 : 1.0;
        }
    for (int m = 0; m < 16; m++)
        for (int k = 0; k < 4; k++) g_lut_na[m * 4 + k] = ((m >> k) & 1) ? 1.0 : 0.0;
    g_luts_ready = 1;
}
const double* orc_dotp_lut_a(void) { build_luts(); return g_lut_a; }
const double* orc_dotp_lut_b(void) { build_luts(); return g_lut_b; }
const double* orc_na_lut(void)     { build_luts(); return g_lut_na; }

/* ------------------------------------------------------------------------- */
/* Per-call kernels in the reference's loop order.                            */
/* ------------------------------------------------------------------------- */

/* bayes.cpp:749-766 (the active, non-MANVEC branch) */
double orc_dot_product(const uint8_t* bed, const double* phen, int mbytes,
                       double mu, double sigma_inv) {
    build_luts();
    double dpa = 0.0, dpb = 0.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+:dpa,dpb)
#endif
    for (int i = 0; i < mbytes; i++) {
        for (int j = 0; j < 4; j++) {
            dpa += g_lut_a[bed[i] * 4 + j] * phen[i * 4 + j];
            dpb += g_lut_b[bed[i] * 4 + j] * phen[i * 4 + j];
        }
    }
    return sigma_inv * (dpa - mu * dpb);
}

/* phenotype.cpp:326-331,375-390 */
void orc_update_epsilon(double* epsilon, const double* dbeta, const uint8_t* bed,
                        const uint8_t* mask4, int im4) {
    build_luts();
    const double bs_ = dbeta[0] * dbeta[2];
    const double mdb = -dbeta[1];
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (int i = 0; i < im4; i++) {
        const int bedi = bed[i] * 4;
        const int masi = mask4[i] * 4;
        for (int j = 0; j < 4; j++) {
            double a = g_lut_a[bedi + j];
            double b = g_lut_b[bedi + j];
            double m = g_lut_na[masi + j];
            epsilon[i * 4 + j] += (mdb * b + a) * bs_ * m;
        }
    }
}

/* phenotype.cpp:395-411 */
void orc_offset_epsilon(double* epsilon, double offset, const uint8_t* mask4, int im4) {
    build_luts();
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (int i = 0; i < im4; i++) {
        const int masi = mask4[i] * 4;
        for (int j = 0; j < 4; j++) epsilon[i * 4 + j] += offset * g_lut_na[masi + j];
    }
}

/* phenotype.cpp:251-261 */
double orc_epsilon_sumsqr(const double* epsilon, int N) {
    double sumsqr = 0.0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+:sumsqr)
#endif
    for (int i = 0; i < N; i++) sumsqr += epsilon[i] * epsilon[i];
    return sumsqr;
}

/* phenotype.cpp:448-457; returns the value set_sigmae() receives */
double orc_epsilon_sigma(const double* epsilon, const uint8_t* mask4, int im4, int nonas) {
    build_luts();
    double sigmae = 0.0;
    for (int i = 0; i < im4; i++)
        for (int j = 0; j < 4; j++)
            sigmae += epsilon[i * 4 + j] * epsilon[i * 4 + j] * g_lut_na[mask4[i] * 4 + j];
    return sigmae / (double)nonas * 0.5;
}

/* phenotype.cpp:525-550 */
void orc_marker_stats(const uint8_t* bed, int N, int M, int mbytes, const uint8_t* mask4,
                      int nonas, double* mave, double* msig) {
    build_luts();
    const int im4 = (N % 4 == 0) ? N / 4 : N / 4 + 1;
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (int i = 0; i < M; i++) {
        const uint8_t* bedm = &bed[(size_t)i * (size_t)mbytes];
        double suma = 0.0, sumb = 0.0;
        for (int j = 0; j < im4; j++)
            for (int k = 0; k < 4; k++) {
                suma += g_lut_a[bedm[j] * 4 + k] * g_lut_na[mask4[j] * 4 + k];
                sumb += g_lut_b[bedm[j] * 4 + k] * g_lut_na[mask4[j] * 4 + k];
            }
        mave[i] = suma / sumb;
        double sumsqr = 0.0;
        for (int j = 0; j < im4; j++)
            for (int k = 0; k < 4; k++) {
                double val = (g_lut_a[bedm[j] * 4 + k] - mave[i]) * g_lut_b[bedm[j] * 4 + k]
                             * g_lut_na[mask4[j] * 4 + k];
                sumsqr += val * val;
            }
        msig[i] = 1.0 / sqrt(sumsqr / ((double)nonas - 1.0));
    }
}

/* ------------------------------------------------------------------------- */
/* Order-independent ("canon") forms of the same reductions.                  */
/*                                                                            */
/* split2(x) = (q1, q2): q1 = x rounded to a multiple of 2^-22, q2 = (x - q1) */
/* rounded to a multiple of 2^-53; the remainder (< 2^-54) is dropped.  For   */
/* |x| < 2^8 and up to 2^22 individuals, sums of a*q1 (a in {0,1,2}) stay     */
/* below 2^31 on a 2^-22 grid and sums of a*q2 below 1 on a 2^-53 grid: every */
/* partial sum is exactly representable, so any summation order (CPU loop,    */
/* wavefront shuffles, cross-workgroup) gives the same bits.                  */
/* ------------------------------------------------------------------------- */
#define ORC_C1  0x1.8p+30    /* 1.5 * 2^(52-22) */
#define ORC_C2  0x1.8p-1     /* 1.5 * 2^(52-53) */
#define ORC_S1  0x1.8p+38    /* for squares (< 2^16): grid 2^-14 */
#define ORC_S2  0x1.8p+7     /*                         grid 2^-45 */

void orc_split2(double x, double* q1, double* q2) {
    double t = x + ORC_C1;
    double a = t - ORC_C1;
    double r = x - a;
    double u = r + ORC_C2;
    *q1 = a;
    *q2 = u - ORC_C2;
}
static inline void split2sq(double x, double* q1, double* q2) {
    double t = x + ORC_S1;
    double a = t - ORC_S1;
    double r = x - a;
    double u = r + ORC_S2;
    *q1 = a;
    *q2 = u - ORC_S2;
}

double orc_dot_product_canon(const uint8_t* bed, const double* phen, int mbytes,
                             double mu, double sigma_inv) {
    build_luts();
    double sa1 = 0.0, sa2 = 0.0, sb1 = 0.0, sb2 = 0.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+:sa1,sa2,sb1,sb2)
#endif
    for (int i = 0; i < mbytes; i++)
        for (int j = 0; j < 4; j++) {
            double q1, q2;
            orc_split2(phen[i * 4 + j], &q1, &q2);
            const double a = g_lut_a[bed[i] * 4 + j], b = g_lut_b[bed[i] * 4 + j];
            sa1 += a * q1; sa2 += a * q2;
            sb1 += b * q1; sb2 += b * q2;
        }
    const double dpa = sa1 + sa2;
    const double dpb = sb1 + sb2;
    return sigma_inv * (dpa - mu * dpb);
}

double orc_epsilon_sumsqr_canon(const double* epsilon, int N) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < N; i++) {
        double q1, q2;
        split2sq(epsilon[i] * epsilon[i], &q1, &q2);
        s1 += q1; s2 += q2;
    }
    return s1 + s2;
}

double orc_epsilon_sigma_canon(const double* epsilon, const uint8_t* mask4, int im4, int nonas) {
    build_luts();
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < im4; i++)
        for (int j = 0; j < 4; j++) {
            double q1, q2;
            split2sq(epsilon[i * 4 + j] * epsilon[i * 4 + j] * g_lut_na[mask4[i] * 4 + j], &q1, &q2);
            s1 += q1; s2 += q2;
        }
    return (s1 + s2) / (double)nonas * 0.5;
}

/* ---- the fixed-point residual of the canon mode -------------------------------------------
 * In canon mode the residual lives on the grid 2^-44: every value is k * 2^-44 with |k| < 2^52
 * (|eps| < 2^8), i.e. an exactly representable double, and every update ADDS a grid value, so
 * eps + v is exact (no rounding per individual) and a dot product after an update equals the dot
 * product before it plus an integer-linear correction.  That is what lets the HIP sweep kernel walk
 * past a marker whose effect changes without recomputing the dots behind it (DESIGN.md 5.1,
 * "continuation"); the ref mode keeps the reference's plain f64 update (phenotype.cpp:375-390).
 * The update values of one marker are LINEAR in the genotype value a: v(a) = beta_ + a * alpha_ with
 *   alpha_ = grid(bs_)          bs_ = dbeta * msig                    (phenotype.cpp:328)
 *   beta_  = grid(mdb * bs_)    mdb = -mave                           (phenotype.cpp:329)
 * against the reference's fl(fl(mdb * b + a) * bs_): the same real number to ~1 ulp, then the grid
 * (|error| <= 2^-44 per individual and update, nine orders of magnitude inside the 1e-6 bar). */
#define ORC_GRID     0x1p-44
#define ORC_GRID_INV 0x1p+44
double orc_grid(double x) { return rint(x * ORC_GRID_INV) * ORC_GRID; }
void orc_grid_array(double* x, int n) { for (int i = 0; i < n; i++) x[i] = orc_grid(x[i]); }

void orc_update_epsilon_canon(double* epsilon, const double* dbeta, const uint8_t* bed,
                              const uint8_t* mask4, int im4) {
    const double bs_ = dbeta[0] * dbeta[2];
    const double mdb = -dbeta[1];
    const double alpha_ = orc_grid(bs_);
    const double beta_ = orc_grid(mdb * bs_);
    const double v1 = beta_ + alpha_, v2 = v1 + alpha_;      /* exact while in range */
    for (int i = 0; i < im4; i++)
        for (int j = 0; j < 4; j++) {
            if (!((mask4[i] >> j) & 1)) continue;            /* na_lut == 0 */
            const int c = (bed[i] >> (2 * j)) & 3;
            if (c == 1) continue;                            /* missing genotype: a = b = 0 */
            epsilon[i * 4 + j] += (c == 0) ? v2 : (c == 2) ? v1 : beta_;
        }
}

void orc_offset_epsilon_canon(double* epsilon, double offset, const uint8_t* mask4, int im4) {
    const double off = orc_grid(offset);
    for (int i = 0; i < im4; i++)
        for (int j = 0; j < 4; j++)
            if ((mask4[i] >> j) & 1) epsilon[i * 4 + j] += off;
}

/* genotype-code counts among non-NA individuals: cnt[c], c = 2-bit code */
void orc_marker_counts(const uint8_t* bedm, int mbytes, const uint8_t* mask4, int64_t cnt[4]) {
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    for (int j = 0; j < mbytes; j++)
        for (int k = 0; k < 4; k++)
            if ((mask4[j] >> k) & 1) cnt[(bedm[j] >> (2 * k)) & 3]++;
}

/* Same quantities as orc_marker_stats; the sums over individuals are taken as
 * integer counts (exact), sum((a-mave)^2) as n0*v0^2 + n2*v2^2 + n3*v3^2. */
void orc_marker_stats_canon(const uint8_t* bed, int N, int M, int mbytes, const uint8_t* mask4,
                            int nonas, double* mave, double* msig) {
    (void)N;
    for (int i = 0; i < M; i++) {
        int64_t n[4];
        orc_marker_counts(&bed[(size_t)i * (size_t)mbytes], mbytes, mask4, n);
        const double suma = (double)(2 * n[0] + n[2]);
        const double sumb = (double)(n[0] + n[2] + n[3]);
        const double av = suma / sumb;
        const double v0 = 2.0 - av, v2 = 1.0 - av, v3 = 0.0 - av;
        double s = (double)n[0] * (v0 * v0);
        s += (double)n[2] * (v2 * v2);
        s += (double)n[3] * (v3 * v3);
        mave[i] = av;
        msig[i] = 1.0 / sqrt(s / ((double)nonas - 1.0));
    }
}

/* The path's exp(): one fixed sequence of IEEE-754 operations (fma, mul, add), so
 * the CPU oracle and the GPU sampler produce the same bits.  |error| ~ 1 ulp.
 * (The reference calls libm exp at bayes.cpp:441,472; "ref" mode below does too.) */
double orc_exp(double x) {
    if (x != x) return x;
    if (x > 0x1.62e42fefa39efp+9) return INFINITY;
    if (x < -0x1.74910d52d3052p+9) return 0.0;
    const double t = x * 0x1.71547652b82fep+0;
    const double kd = (t + 0x1.8p52) - 0x1.8p52;
    double r = fma(-kd, 0x1.62e42fee00000p-1, x);
    r = fma(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = fma(p, r, 0x1.1eed8eff8d898p-29);
    p = fma(p, r, 0x1.ae64567f544e4p-26);
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma(p, r, 0x1.71de3a556c734p-19);
    p = fma(p, r, 0x1.a01a01a01a01ap-16);
    p = fma(p, r, 0x1.a01a01a01a01ap-13);
    p = fma(p, r, 0x1.6c16c16c16c17p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0x1.0000000000000p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int k = (int)kd;
    const int k1 = k / 2, k2 = k - k1;
    union { uint64_t u; double d; } s1, s2;
    s1.u = (uint64_t)(k1 + 1023) << 52;
    s2.u = (uint64_t)(k2 + 1023) << 52;
    return (p * s1.d) * s2.d;
}

/* ------------------------------------------------------------------------- */
/* Phenotype preparation.  phenotype.cpp:587-673 (read_file), after the text  */
/* has been tokenised: y[i] = 3rd column, isna[i] = (token == "NA").          */
/* ------------------------------------------------------------------------- */
void orc_phen_prepare(const double* y, const uint8_t* isna, int N,
                      double* epsilon, uint8_t* mask4, int* nonas_out) {
    const int im4 = (N % 4 == 0) ? N / 4 : N / 4 + 1;
    int nonas = 0;
    double sum = 0.0;
    for (int i = 0; i < im4; i++) mask4[i] = 0x0F;
    for (int i = 0; i < N; i++) {
        if (isna[i]) mask4[i / 4] &= (uint8_t)~(1u << (i % 4));
        else { nonas++; sum += y[i]; }
    }
    if (N % 4 != 0)
        for (int i = N % 4; i < 4; i++) mask4[N / 4] &= (uint8_t)~(1u << i);
    const double avg = sum / (double)nonas;
    double sqn = 0.0;
    for (int i = 0; i < N; i++) {
        if (isna[i]) epsilon[i] = 0.0;
        else { epsilon[i] = y[i] - avg; sqn += epsilon[i] * epsilon[i]; }
    }
    sqn = sqrt((double)(nonas - 1) / sqn);
    for (int i = 0; i < N; i++) epsilon[i] *= sqn;
    /* the reference leaves epsilon_[N..4*im4) uninitialised and
     * relies on zero pages; the restatement zeroes it. */
    for (int i = N; i < 4 * im4; i++) epsilon[i] = 0.0;
    *nonas_out = nonas;
}

/* ------------------------------------------------------------------------- */
/* RNG spec.  distributions.hpp:5-61 draws through Boost.Random (boost 1.76,   */
/* setup/Make.intel_ioampi:6), which is neither vendored in the reference nor  */
/* installed here.  The algorithms below restate Boost's published ones:       */
/*   mt19937                 : Matsumoto & Nishimura, 32-bit, seed recurrence  */
/*   uniform_real(0,1), uniform_01 : one 32-bit output * 2^-32                 */
/*   normal_distribution     : 128-layer ziggurat, 8-bit bucket (1 sign bit +  */
/*                             7 layer bits) + 53-bit fraction from 2 outputs  */
/*   exponential_distribution: 256-layer ziggurat, tail = restart with shift   */
/*   gamma_distribution      : alpha==1 exponential; alpha>1 Cheng/tan          */
/*                             rejection; alpha<1 Ahrens-Dieter GS             */
/*   beta_distribution       : X/(X+Y) of two gammas                           */
/*   uniform_int (shuffle)   : bucket rejection on one 32-bit output           */
/*   random_shuffle          : libstdc++ std::random_shuffle(first,last,rand)  */
/* PARITY UNPINNED: no reference test or fixture pins any draw.                */
/* ------------------------------------------------------------------------- */
void orc_rng_seed(orc_rng* r, uint32_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
}
static void rng_twist(orc_rng* r) {
    uint32_t* mt = r->mt;
    for (int i = 0; i < 624; i++) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[i] = v;
    }
    r->idx = 0;
}
uint32_t orc_rng_u32(orc_rng* r) {
    if (r->idx >= 624) rng_twist(r);
    uint32_t y = r->mt[r->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* boost/random/uniform_real_distribution.hpp generate_uniform_real (integer engine);
 * distributions.hpp:55-59 */
double orc_rng_unif(orc_rng* r) {
    for (;;) {
        double result = (double)orc_rng_u32(r) / 4294967296.0 * (1.0 - 0.0) + 0.0;
        if (result < 1.0) return result;
    }
}
static double rng_u01(orc_rng* r) {          /* boost uniform_01<double> */
    for (;;) {
        double result = (double)orc_rng_u32(r) * (1.0 / 4294967296.0);
        if (result < 1.0) return result;
    }
}
/* boost/random/detail/int_float_pair.hpp, w = 8, 32-bit engine, 53-bit double */
static double rng_int_float_pair(orc_rng* r, int* bucket) {
    uint32_t u1 = orc_rng_u32(r);
    *bucket = (int)(u1 & 0xFFu);
    double x = (double)(u1 >> 8) * (1.0 / 16777216.0);
    uint32_t u2 = orc_rng_u32(r);
    x += (double)(u2 & 0x1FFFFFFFu);
    x *= (1.0 / 536870912.0);
    return x;
}
/* boost/random/exponential_distribution.hpp unit_exponential_distribution */
double orc_rng_exponential(orc_rng* r) {
    const double* tx = gm_zig_exp_x;
    const double* ty = gm_zig_exp_y;
    double shift = 0.0;
    for (;;) {
        int i;
        double x = rng_int_float_pair(r, &i) * tx[i];
        if (x < tx[i + 1]) return shift + x;
        if (i == 0) { shift += tx[1]; continue; }
        double y01 = rng_u01(r);
        double y = ty[i] + y01 * (ty[i + 1] - ty[i]);
        double y_above_ubound = (tx[i] - tx[i + 1]) * y01 - (tx[i] - x);
        double y_above_lbound = y - (ty[i + 1] + (tx[i + 1] - x) * ty[i + 1]);
        if (y_above_ubound < 0.0 && (y_above_lbound < 0.0 || y < orc_exp(-x))) return x + shift;
    }
}
/* boost/random/normal_distribution.hpp unit_normal_distribution */
static double rng_unit_normal(orc_rng* r) {
    const double* tx = gm_zig_norm_x;
    const double* ty = gm_zig_norm_y;
    for (;;) {
        int b;
        double x01 = rng_int_float_pair(r, &b);
        int sign = (b & 1) * 2 - 1;
        int i = b >> 1;
        double x = x01 * tx[i];
        if (x < tx[i + 1]) return x * sign;
        if (i == 0) {
            const double tail_start = tx[1];
            for (;;) {
                double xx = orc_rng_exponential(r) / tail_start;
                double yy = orc_rng_exponential(r);
                if (2.0 * yy > xx * xx) return (xx + tail_start) * sign;
            }
        }
        double y01 = rng_u01(r);
        double y = ty[i] + y01 * (ty[i + 1] - ty[i]);
        double y_above_ubound, y_above_lbound;
        if (tx[i] >= 1.0) {
            y_above_ubound = (tx[i] - tx[i + 1]) * y01 - (tx[i] - x);
            y_above_lbound = y - (ty[i] + (tx[i] - x) * ty[i] * tx[i]);
        } else {
            y_above_lbound = (tx[i] - tx[i + 1]) * y01 - (tx[i] - x);
            y_above_ubound = y - (ty[i] + (tx[i] - x) * ty[i] * tx[i]);
        }
        if (y_above_ubound < 0.0 && (y_above_lbound < 0.0 || y < orc_exp(-(x * x / 2.0))))
            return x * sign;
    }
}
/* distributions.hpp:48-53: normal(mean, sqrt(sigma2)) */
double orc_rng_norm(orc_rng* r, double mean, double sigma2) {
    const double sigma = sqrt(sigma2);
    return rng_unit_normal(r) * sigma + mean;
}
/* boost/random/gamma_distribution.hpp; distributions.hpp:32-37 */
double orc_rng_gamma(orc_rng* r, double alpha, double beta) {
    if (alpha == 1.0) return orc_rng_exponential(r) * beta;
    if (alpha > 1.0) {
        const double pi = 3.14159265358979323846;
        for (;;) {
            double y = tan(pi * rng_u01(r));
            double x = sqrt(2.0 * alpha - 1.0) * y + alpha - 1.0;
            if (x <= 0.0) continue;
            if (rng_u01(r) > (1.0 + y * y) * exp((alpha - 1.0) * log(x / (alpha - 1.0))
                                                 - sqrt(2.0 * alpha - 1.0) * y))
                continue;
            return x * beta;
        }
    }
    const double p = exp(1.0) / (alpha + exp(1.0));
    for (;;) {
        double u = rng_u01(r);
        double y = orc_rng_exponential(r);
        double x, q;
        if (u < p) { x = exp(-y / alpha); q = p * exp(-x); }
        else       { x = 1.0 + y;         q = p + (1.0 - p) * pow(x, alpha - 1.0); }
        if (u >= q) continue;
        return x * beta;
    }
}
/* boost/random/beta_distribution.hpp; distributions.hpp:39-46 */
double orc_rng_beta(orc_rng* r, double a, double b) {
    double x = orc_rng_gamma(r, a, 1.0);
    double y = orc_rng_gamma(r, b, 1.0);
    return x / (x + y);
}
/* distributions.hpp:24-30 */
double orc_rng_inv_scaled_chisq(orc_rng* r, double a, double b) {
    const double ga = 0.5 * a, gb = 0.5 * a * b;
    return 1.0 / orc_rng_gamma(r, ga, 1.0 / gb);
}
/* boost/random/uniform_int_distribution.hpp generate_uniform_int, range n-1 < 2^32-1 */
static uint32_t rng_uniform_int(orc_rng* r, uint32_t n) {   /* in [0, n-1] */
    const uint32_t range = n - 1;
    if (range == 0) return 0;
    uint32_t bucket_size = 0xFFFFFFFFu / (range + 1u);
    if (0xFFFFFFFFu % (range + 1u) == range) ++bucket_size;
    for (;;) {
        uint32_t result = orc_rng_u32(r) / bucket_size;
        if (result <= range) return result;
    }
}
/* phenotype.cpp:314-323 -> std::random_shuffle(first, last, rand) (libstdc++) */
void orc_rng_shuffle(orc_rng* r, int* v, int n) {
    for (int i = 1; i < n; i++) {
        int j = (int)rng_uniform_int(r, (uint32_t)i + 1u);
        if (i != j) { int t = v[i]; v[i] = v[j]; v[j] = t; }
    }
}

/* ------------------------------------------------------------------------- */
/* One phenotype's chain on one rank.                                         */
/* ------------------------------------------------------------------------- */
struct orc_chain {
    int N, M, Mt, S, G, K, im4, nonas, rank;
    size_t mbytes;
    const uint8_t* bed;
    double* eps;
    uint8_t* mask4;
    double *mave, *msig, *betas, *acum;
    int *comp, *midx;
    int *group_index, *mtotgrp;
    double *cva, *cvai, *pi_prior, *pi_est;
    double *sigmag, *beta_sqn;
    int *cass, *m0;
    double sigmae, mu, epssum;
    orc_rng dist_m, dist_d;
    int shuffle, mimic_hydra, canon;
    long n_updates;
};

static const double V0E = 0.0001, S02E = 0.0001, V0G = 0.0001, S02G = 0.0001; /* bayes.hpp:14-17 */

orc_chain* orc_chain_create(int N, int M, int Mt, int S, int G, int K,
                            const uint8_t* bed_local, const double* eps0,
                            const uint8_t* mask4, int nonas,
                            const int* group_index, const double* cva,
                            uint32_t seed, int rank, int shuffle, int mimic_hydra, int canon) {
    if (K > ORC_KMAX || K < 2) return NULL;
    orc_chain* c = (orc_chain*)calloc(1, sizeof(orc_chain));
    c->N = N; c->M = M; c->Mt = Mt; c->S = S; c->G = G; c->K = K; c->rank = rank;
    c->im4 = (N % 4 == 0) ? N / 4 : N / 4 + 1;          /* phenotype.cpp:22 */
    c->mbytes = (size_t)c->im4;                           /* bayes.cpp:776 */
    c->nonas = nonas;
    c->bed = bed_local;
    c->eps = (double*)malloc(sizeof(double) * 4 * (size_t)c->im4);
    memcpy(c->eps, eps0, sizeof(double) * 4 * (size_t)c->im4);
    if (canon) orc_grid_array(c->eps, 4 * c->im4);            /* the canon residual lives on the 2^-44 grid */
    c->mask4 = (uint8_t*)malloc((size_t)c->im4);
    memcpy(c->mask4, mask4, (size_t)c->im4);
    c->mave = (double*)calloc((size_t)M, sizeof(double));
    c->msig = (double*)calloc((size_t)M, sizeof(double));
    c->betas = (double*)calloc((size_t)M, sizeof(double));   /* phenotype.cpp:38 */
    c->acum = (double*)calloc((size_t)M, sizeof(double));
    c->comp = (int*)calloc((size_t)M, sizeof(int));
    c->midx = (int*)calloc((size_t)M, sizeof(int));
    c->group_index = (int*)malloc(sizeof(int) * (size_t)Mt);
    memcpy(c->group_index, group_index, sizeof(int) * (size_t)Mt);
    c->mtotgrp = (int*)calloc((size_t)G, sizeof(int));
    for (int i = 0; i < Mt; i++) c->mtotgrp[group_index[i]] += 1;     /* bayes.cpp:807-809 */
    c->cva = (double*)malloc(sizeof(double) * (size_t)(G * K));
    c->cvai = (double*)calloc((size_t)(G * K), sizeof(double));
    memcpy(c->cva, cva, sizeof(double) * (size_t)(G * K));
    for (int g = 0; g < G; g++)
        for (int j = 1; j < K; j++) c->cvai[g * K + j] = 1.0 / cva[g * K + j];   /* options.cpp:282 */
    c->pi_prior = (double*)calloc((size_t)(G * K), sizeof(double));
    c->pi_est = (double*)calloc((size_t)(G * K), sizeof(double));
    for (int g = 0; g < G; g++) {                                      /* bayes.hpp:37-47 */
        double sum_cva = 0.0;
        for (int j = 0; j < K - 1; j++) sum_cva += cva[g * K + j + 1];
        c->pi_prior[g * K + 0] = 0.5;
        for (int j = 1; j < K; j++) c->pi_prior[g * K + j] = c->pi_prior[g * K + 0] * cva[g * K + j] / sum_cva;
    }
    c->sigmag = (double*)calloc((size_t)G, sizeof(double));
    c->beta_sqn = (double*)calloc((size_t)G, sizeof(double));
    c->cass = (int*)calloc((size_t)(G * K), sizeof(int));
    c->m0 = (int*)calloc((size_t)G, sizeof(int));
    c->sigmae = 0.0; c->mu = 0.0; c->epssum = 0.0;                     /* phenotype.hpp:52-55 */
    c->shuffle = shuffle; c->mimic_hydra = mimic_hydra; c->canon = canon;
    /* bayes.cpp:796-803 */
    orc_rng_seed(&c->dist_m, (uint32_t)(seed + (uint32_t)rank));
    if (mimic_hydra) orc_rng_seed(&c->dist_d, (uint32_t)(seed + (uint32_t)rank * 1000u));
    else             orc_rng_seed(&c->dist_d, (uint32_t)(seed + (uint32_t)(rank + 1) * 1000u));
    /* bayes.cpp:788 */
    if (canon) orc_marker_stats_canon(c->bed, N, M, (int)c->mbytes, c->mask4, nonas, c->mave, c->msig);
    else       orc_marker_stats(c->bed, N, M, (int)c->mbytes, c->mask4, nonas, c->mave, c->msig);
    return c;
}

void orc_chain_destroy(orc_chain* c) {
    if (!c) return;
    free(c->eps); free(c->mask4); free(c->mave); free(c->msig); free(c->betas); free(c->acum);
    free(c->comp); free(c->midx); free(c->group_index); free(c->mtotgrp); free(c->cva);
    free(c->cvai); free(c->pi_prior); free(c->pi_est); free(c->sigmag); free(c->beta_sqn);
    free(c->cass); free(c->m0); free(c);
}

/* bayes.cpp:322-335 */
void orc_chain_init(orc_chain* c) {
    for (int i = 0; i < c->M; i++) c->midx[i] = i;                 /* phenotype.cpp:308-312 */
    for (int g = 0; g < c->G; g++) {
        c->sigmag[g] = orc_rng_beta(&c->dist_d, 1.0, 1.0);
        if (c->mtotgrp[g] == 0) c->sigmag[g] = 0.0;
    }
    memcpy(c->pi_est, c->pi_prior, sizeof(double) * (size_t)(c->G * c->K));
}

static double chain_dot(orc_chain* c, int mloc) {
    const uint8_t* col = &c->bed[(size_t)mloc * c->mbytes];
    return c->canon ? orc_dot_product_canon(col, c->eps, (int)c->mbytes, c->mave[mloc], c->msig[mloc])
                    : orc_dot_product(col, c->eps, (int)c->mbytes, c->mave[mloc], c->msig[mloc]);
}
static double chain_exp(const orc_chain* c, double x) { return c->canon ? orc_exp(x) : exp(x); }
static void chain_offset(orc_chain* c, double off) {
    if (c->canon) orc_offset_epsilon_canon(c->eps, off, c->mask4, c->im4);
    else          orc_offset_epsilon(c->eps, off, c->mask4, c->im4);
}
static void chain_update(orc_chain* dst, const double* d3, const uint8_t* col) {
    if (dst->canon) orc_update_epsilon_canon(dst->eps, d3, col, dst->mask4, dst->im4);
    else            orc_update_epsilon(dst->eps, d3, col, dst->mask4, dst->im4);
}

/* bayes.cpp:348-358: add the old mu back, (it==1) initial sigmae, draw the new mu */
double orc_chain_prologue_draw(orc_chain* c, int it) {
    chain_offset(c, c->mu);
    if (it == 1)
        c->sigmae = c->canon ? orc_epsilon_sigma_canon(c->eps, c->mask4, c->im4, c->nonas)
                             : orc_epsilon_sigma(c->eps, c->mask4, c->im4, c->nonas);
    /* phenotype.cpp:279-282: epssum is never updated, so the mean is 0/nonas */
    return orc_rng_norm(&c->dist_d, c->epssum / (double)c->nonas, c->sigmae / (double)c->nonas);
}
/* bayes.cpp:358-367: adopt mu, subtract it, shuffle, reset counters */
void orc_chain_prologue_apply(orc_chain* c, double mu) {
    c->mu = mu;
    chain_offset(c, -c->mu);
    if (c->shuffle) orc_rng_shuffle(c->mimic_hydra ? &c->dist_d : &c->dist_m, c->midx, c->M);
    for (int g = 0; g < c->G; g++) c->m0[g] = 0;
    for (int i = 0; i < c->G * c->K; i++) c->cass[i] = 0;
}
/* bayes.cpp:348-368 */
void orc_chain_prologue(orc_chain* c, int it) {
    orc_chain_prologue_apply(c, orc_chain_prologue_draw(c, it));
}

/* bayes.cpp:384-492 for one marker: the Gibbs draw, everything up to (not including) the residual
 * update.  Returns 1 when the effect changed (share_mrk, bayes.cpp:483-488) with d3 = {dbeta, mave, msig}. */
static int chain_marker_decide(orc_chain* c, int mloc, double d3[3]) {
    const int K = c->K, N = c->N;
    d3[0] = d3[1] = d3[2] = 0.0;
    const int mglo = c->S + mloc;
    const int mgrp = c->group_index[mglo];

    if (c->sigmag[mgrp] == 0.0) {            /* bayes.cpp:396-400 (no draw, no residual update) */
        c->acum[mloc] = 1.0;
        c->betas[mloc] = 0.0;
        return 0;
    }
    double beta = c->betas[mloc];
    double sige_g = c->sigmae / c->sigmag[mgrp];
    double sigg_e = 1.0 / sige_g;
    double inv2sige = 1.0 / (2.0 * c->sigmae);
    double denom[ORC_KMAX], muk[ORC_KMAX], logl[ORC_KMAX];
    muk[0] = 0.0;
    for (int i = 1; i <= K - 1; ++i)
        denom[i - 1] = (double)(N - 1) + sige_g * c->cvai[mgrp * K + i];

    double num = chain_dot(c, mloc);
    num += beta * (double)(c->nonas - 1);

    for (int i = 1; i <= K - 1; ++i) muk[i] = num / denom[i - 1];
    for (int i = 0; i < K; i++) {
        logl[i] = log(c->pi_est[mgrp * K + i]);
        if (i > 0)
            logl[i] += -0.5 * log(sigg_e * (double)(c->nonas - 1) * c->cva[mgrp * K + i] + 1.0)
                       + muk[i] * num * inv2sige;
    }
    double prob = orc_rng_unif(&c->dist_d);

    int zero_acum = 0;
    double tmp1 = 0.0;
    for (int i = 0; i < K; i++) {
        if (fabs(logl[i] - logl[0]) > 700.0) zero_acum = 1;
        tmp1 += chain_exp(c, logl[i] - logl[0]);
    }
    tmp1 = zero_acum ? 0.0 : 1.0 / tmp1;
    c->acum[mloc] = tmp1;

    double dbeta = c->betas[mloc];
    for (int i = 0; i < K; i++) {
        if (prob <= c->acum[mloc] || i == K - 1) {
            if (i == 0) c->betas[mloc] = 0.0;
            else c->betas[mloc] = orc_rng_norm(&c->dist_d, muk[i], c->sigmae / denom[i - 1]);
            c->cass[mgrp * K + i] += 1;
            c->comp[mloc] = i;
            break;
        } else {
            int zero_inc = 0;
            for (int j = i + 1; j < K; j++)
                if (fabs(logl[j] - logl[i + 1]) > 700.0) zero_inc = 1;
            if (!zero_inc) {
                double esum = 0.0;
                for (int k = 0; k < K; k++) esum += chain_exp(c, logl[k] - logl[i + 1]);
                c->acum[mloc] = c->acum[mloc] + 1.0 / esum;
            }
        }
    }
    dbeta -= c->betas[mloc];
    if (fabs(dbeta) > 0.0) {                 /* bayes.cpp:483-488 */
        d3[0] = dbeta; d3[1] = c->mave[mloc]; d3[2] = c->msig[mloc];
        return 1;
    }
    return 0;
}

/* one marker of a single rank: the draw, then bayes.cpp:681-706 -> phenotype.cpp:326 */
static void chain_marker_step(orc_chain* c, int mloc) {
    double d3[3];
    if (chain_marker_decide(c, mloc, d3)) {
        chain_update(c, d3, &c->bed[(size_t)mloc * c->mbytes]);
        c->n_updates++;
    }
}

/* bayes.cpp:375-553 restricted to this rank's own markers */
void orc_chain_markers(orc_chain* c) {
    for (int mrki = 0; mrki < c->M; mrki++) chain_marker_step(c, c->midx[mrki]);
}

/* positions [first, first + count) of the visit order (a part of the sweep: orc_nk_iterate, the build's --sync-every k) */
void orc_chain_markers_range(orc_chain* c, int first, int count) {
    for (int mrki = first; mrki < first + count && mrki < c->M; mrki++) chain_marker_step(c, c->midx[mrki]);
}

/* bayes.cpp:565-568 */
void orc_chain_local_sums(orc_chain* c) {
    for (int g = 0; g < c->G; g++) c->beta_sqn[g] = 0.0;
    for (int i = 0; i < c->M; i++)
        c->beta_sqn[c->group_index[c->S + i]] += c->betas[i] * c->betas[i];
}

/* bayes.cpp:590-651 (after the all-reduces of beta_sqn and cass) */
void orc_chain_epilogue(orc_chain* c) {
    const int G = c->G, K = c->K;
    for (int g = 0; g < G; g++) {
        if (c->mtotgrp[g] == 0) continue;
        c->m0[g] = c->mtotgrp[g] - c->cass[g * K + 0];
        int cass_sum = 0;
        for (int k = 0; k < K; k++) cass_sum += c->cass[g * K + k];
        if (c->m0[g] == 0 || cass_sum == 0) { c->sigmag[g] = 0.0; continue; }
        const double m0 = (double)c->m0[g];
        c->sigmag[g] = orc_rng_inv_scaled_chisq(&c->dist_d, V0G + m0,
                           (c->beta_sqn[g] * m0 + V0G * S02G) / (V0G + m0));
        /* phenotype.cpp:227-237 */
        double sum = 0.0;
        for (int i = 0; i < K; i++) {
            double val = orc_rng_gamma(&c->dist_d, (double)c->cass[g * K + i] + 1.0, 1.0);
            c->pi_est[g * K + i] = val;
            sum += val;
        }
        for (int i = 0; i < K; i++) c->pi_est[g * K + i] = c->pi_est[g * K + i] / sum;
    }
    const double e_sqn = c->canon ? orc_epsilon_sumsqr_canon(c->eps, c->N)
                                  : orc_epsilon_sumsqr(c->eps, c->N);
    c->sigmae = orc_rng_inv_scaled_chisq(&c->dist_d, V0E + (double)c->N,
                    (e_sqn + V0E * S02E) / (V0E + (double)c->N));
}

void orc_chain_iterate(orc_chain* c, int it) {
    orc_chain_prologue(c, it);
    orc_chain_markers(c);
    orc_chain_local_sums(c);
    orc_chain_epilogue(c);
}

/* The build's sweep-synchronous schedule for R ranks holding disjoint marker shards
 * of ONE phenotype (DESIGN.md "Multi-GPU"); replaces bayes.cpp:495-553's per-step
 * exchange with one residual exchange per sweep:
 *   1. every rank draws mu on its own RNG; rank 0's draw is adopted by all, then each
 *      rank subtracts it and shuffles its own marker order;
 *   2. every rank sweeps its own markers against its own residual replica;
 *   3. delta_r = eps_r - eps_start is split with split2() and the two parts are summed
 *      over ranks (exact, hence order-free): eps = eps_start + (sum q1 + sum q2);
 *   4. cass is summed (ints), beta_sqn is summed in rank order (bayes.cpp:575-588);
 *   5. every rank runs the epilogue on its own RNG, then adopts rank 0's sigmag,
 *      pi_est and sigmae (bayes.cpp:626,638,649).
 * With R == 1 this is orc_chain_iterate (no exchange). */
void orc_ns_iterate(orc_chain** ch, int R, int it) {
    if (R == 1) { orc_chain_iterate(ch[0], it); return; }
    const int n4 = 4 * ch[0]->im4, G = ch[0]->G, K = ch[0]->K;
    double mu0 = 0.0;
    for (int r = 0; r < R; r++) {
        double mu_r = orc_chain_prologue_draw(ch[r], it);   /* every rank consumes its draw */
        if (r == 0) mu0 = mu_r;
    }
    for (int r = 0; r < R; r++) orc_chain_prologue_apply(ch[r], mu0);
    double* start = (double*)malloc(sizeof(double) * (size_t)n4);
    double* s1 = (double*)calloc((size_t)n4, sizeof(double));
    double* s2 = (double*)calloc((size_t)n4, sizeof(double));
    memcpy(start, ch[0]->eps, sizeof(double) * (size_t)n4);
    for (int r = 0; r < R; r++) {
        orc_chain_markers(ch[r]);
        orc_chain_local_sums(ch[r]);
        for (int i = 0; i < n4; i++) {
            double q1, q2;
            orc_split2(ch[r]->eps[i] - start[i], &q1, &q2);
            s1[i] += q1; s2[i] += q2;
        }
    }
    int* cass = (int*)calloc((size_t)(G * K), sizeof(int));
    double* bsq = (double*)calloc((size_t)G, sizeof(double));
    for (int r = 0; r < R; r++) {
        for (int i = 0; i < G * K; i++) cass[i] += ch[r]->cass[i];
        for (int g = 0; g < G; g++) bsq[g] += ch[r]->beta_sqn[g];
    }
    for (int r = 0; r < R; r++) {
        for (int i = 0; i < n4; i++) ch[r]->eps[i] = start[i] + (s1[i] + s2[i]);
        memcpy(ch[r]->cass, cass, sizeof(int) * (size_t)(G * K));
        memcpy(ch[r]->beta_sqn, bsq, sizeof(double) * (size_t)G);
        orc_chain_epilogue(ch[r]);
    }
    for (int r = 1; r < R; r++) {
        memcpy(ch[r]->sigmag, ch[0]->sigmag, sizeof(double) * (size_t)G);
        memcpy(ch[r]->pi_est, ch[0]->pi_est, sizeof(double) * (size_t)(G * K));
        ch[r]->sigmae = ch[0]->sigmae;
    }
    free(start); free(s1); free(s2); free(cass); free(bsq);
}

/* orc_ns_iterate with the residual exchange every k marker positions instead of once per sweep (the build's
 * `--sync-every k`, 1 < k < M; no upstream counterpart -- upstream exchanges after every marker, orc_ps_iterate):
 * part p = positions [p k, (p + 1) k) of every rank's own visit order, swept against the rank's own replica; behind
 * every part the replicas are reconciled as in step 3 of orc_ns_iterate.  k >= max M_r is orc_ns_iterate. */
void orc_nk_iterate(orc_chain** ch, int R, int it, int k) {
    const int n4 = 4 * ch[0]->im4, G = ch[0]->G, K = ch[0]->K;
    double mu0 = 0.0;
    for (int r = 0; r < R; r++) {
        double mu_r = orc_chain_prologue_draw(ch[r], it);
        if (r == 0) mu0 = mu_r;
    }
    int Mm = 0;
    for (int r = 0; r < R; r++) {
        orc_chain_prologue_apply(ch[r], mu0);
        if (ch[r]->M > Mm) Mm = ch[r]->M;
    }
    double* start = (double*)malloc(sizeof(double) * (size_t)n4);
    double* s1 = (double*)malloc(sizeof(double) * (size_t)n4);
    double* s2 = (double*)malloc(sizeof(double) * (size_t)n4);
    for (int first = 0; first < Mm; first += k) {
        memcpy(start, ch[0]->eps, sizeof(double) * (size_t)n4);
        memset(s1, 0, sizeof(double) * (size_t)n4);
        memset(s2, 0, sizeof(double) * (size_t)n4);
        for (int r = 0; r < R; r++) {
            const int end = first + k < ch[r]->M ? first + k : ch[r]->M;
            for (int mrki = first; mrki < end; mrki++) chain_marker_step(ch[r], ch[r]->midx[mrki]);
            for (int i = 0; i < n4; i++) {
                double q1, q2;
                orc_split2(ch[r]->eps[i] - start[i], &q1, &q2);
                s1[i] += q1; s2[i] += q2;
            }
        }
        if (R > 1)
            for (int r = 0; r < R; r++)
                for (int i = 0; i < n4; i++) ch[r]->eps[i] = start[i] + (s1[i] + s2[i]);
    }
    int* cass = (int*)calloc((size_t)(G * K), sizeof(int));
    double* bsq = (double*)calloc((size_t)G, sizeof(double));
    for (int r = 0; r < R; r++) {
        orc_chain_local_sums(ch[r]);
        for (int i = 0; i < G * K; i++) cass[i] += ch[r]->cass[i];
        for (int g = 0; g < G; g++) bsq[g] += ch[r]->beta_sqn[g];
    }
    for (int r = 0; r < R; r++) {
        memcpy(ch[r]->cass, cass, sizeof(int) * (size_t)(G * K));
        memcpy(ch[r]->beta_sqn, bsq, sizeof(double) * (size_t)G);
        orc_chain_epilogue(ch[r]);
    }
    for (int r = 1; r < R; r++) {
        memcpy(ch[r]->sigmag, ch[0]->sigmag, sizeof(double) * (size_t)G);
        memcpy(ch[r]->pi_est, ch[0]->pi_est, sizeof(double) * (size_t)(G * K));
        ch[r]->sigmae = ch[0]->sigmae;
    }
    free(start); free(s1); free(s2); free(cass); free(bsq);
}

/* The reference's own schedule for R ranks holding disjoint marker shards of ONE phenotype
 * (bayes.cpp:340-651 read as R MPI tasks in one loop):
 *   prologue  every rank draws AND uses its own mu from its own stream (bayes.cpp:348-358; seeds :796-803),
 *             so the residual replicas differ by that offset, as upstream's do;
 *   step mrki of max_r M_r (bayes.cpp:374): every rank with mrki < M_r draws the effect of its marker
 *             midx_r[mrki] against its own replica (bayes.cpp:381-492); then every replica applies the changed
 *             markers of ALL ranks in rank order (Allgather/Allgatherv + update_epsilon, bayes.cpp:495-553,
 *             681-706);
 *   epilogue  cass summed, beta_sqn summed in rank order (bayes.cpp:575-588), every rank draws the hyper-
 *             parameters on its own stream and adopts rank 0's (bayes.cpp:626,638,649). */
void orc_ps_iterate(orc_chain** ch, int R, int it) {
    const int G = ch[0]->G, K = ch[0]->K;
    int Mm = 0;
    for (int r = 0; r < R; r++) {
        orc_chain_prologue(ch[r], it);
        if (ch[r]->M > Mm) Mm = ch[r]->M;
    }
    double (*d3)[3] = (double (*)[3])malloc(sizeof(double[3]) * (size_t)R);
    int* share = (int*)malloc(sizeof(int) * (size_t)R);
    int* mloc = (int*)malloc(sizeof(int) * (size_t)R);
    for (int mrki = 0; mrki < Mm; mrki++) {
        for (int r = 0; r < R; r++) {
            share[r] = 0; mloc[r] = 0;
            if (mrki < ch[r]->M) {
                mloc[r] = ch[r]->midx[mrki];
                share[r] = chain_marker_decide(ch[r], mloc[r], d3[r]);
                ch[r]->n_updates += share[r];
            }
        }
        for (int dst = 0; dst < R; dst++)
            for (int r = 0; r < R; r++)
                if (share[r])
                    chain_update(ch[dst], d3[r], &ch[r]->bed[(size_t)mloc[r] * ch[r]->mbytes]);
    }
    int* cass = (int*)calloc((size_t)(G * K), sizeof(int));
    double* bsq = (double*)calloc((size_t)G, sizeof(double));
    for (int r = 0; r < R; r++) {
        orc_chain_local_sums(ch[r]);
        for (int i = 0; i < G * K; i++) cass[i] += ch[r]->cass[i];
        for (int g = 0; g < G; g++) bsq[g] += ch[r]->beta_sqn[g];
    }
    for (int r = 0; r < R; r++) {
        memcpy(ch[r]->cass, cass, sizeof(int) * (size_t)(G * K));
        memcpy(ch[r]->beta_sqn, bsq, sizeof(double) * (size_t)G);
        orc_chain_epilogue(ch[r]);
    }
    for (int r = 1; r < R; r++) {
        memcpy(ch[r]->sigmag, ch[0]->sigmag, sizeof(double) * (size_t)G);
        memcpy(ch[r]->pi_est, ch[0]->pi_est, sizeof(double) * (size_t)(G * K));
        ch[r]->sigmae = ch[0]->sigmae;
    }
    free(d3); free(share); free(mloc); free(cass); free(bsq);
}

double* orc_chain_eps(orc_chain* c) { return c->eps; }
double* orc_chain_betas(orc_chain* c) { return c->betas; }
double* orc_chain_acum(orc_chain* c) { return c->acum; }
int*    orc_chain_comp(orc_chain* c) { return c->comp; }
int*    orc_chain_midx(orc_chain* c) { return c->midx; }
int*    orc_chain_cass(orc_chain* c) { return c->cass; }
int*    orc_chain_m0(orc_chain* c) { return c->m0; }
double* orc_chain_sigmag(orc_chain* c) { return c->sigmag; }
double* orc_chain_pi_est(orc_chain* c) { return c->pi_est; }
double* orc_chain_beta_sqn(orc_chain* c) { return c->beta_sqn; }
double* orc_chain_mave(orc_chain* c) { return c->mave; }
double* orc_chain_msig(orc_chain* c) { return c->msig; }
double  orc_chain_sigmae(orc_chain* c) { return c->sigmae; }
double  orc_chain_mu(orc_chain* c) { return c->mu; }
void    orc_chain_set_sigmae(orc_chain* c, double v) { c->sigmae = v; }
long    orc_chain_nupdates(orc_chain* c) { return c->n_updates; }
orc_rng* orc_chain_rng_d(orc_chain* c) { return &c->dist_d; }
orc_rng* orc_chain_rng_m(orc_chain* c) { return &c->dist_m; }
int orc_chain_m0_sum(orc_chain* c) {       /* phenotype.cpp:98-103 */
    int s = 0;
    for (int g = 0; g < c->G; g++) s += c->m0[g];
    return s;
}

/* xfiles.cpp:17-42: one .csv record */
int orc_csv_line(char* buf, size_t len, unsigned it, const double* sigmag, int G,
                 double sigmae, int m0_sum, const double* pi_est, int K) {
    size_t n = 0;
    n += (size_t)snprintf(buf + n, len - n, "%5d, %4d", it, G);
    double sigmag_sum = 0.0;
    for (int i = 0; i < G; i++) n += (size_t)snprintf(buf + n, len - n, ", %20.15f", sigmag[i]);
    for (int i = 0; i < G; i++) sigmag_sum += sigmag[i];
    n += (size_t)snprintf(buf + n, len - n, ", %20.15f, %20.15f, %7d, %4d, %2d",
                          sigmae, sigmag_sum / (sigmae + sigmag_sum), m0_sum, G, K);
    for (int i = 0; i < G; i++)
        for (int j = 0; j < K; j++) n += (size_t)snprintf(buf + n, len - n, ", %20.15f", pi_est[i * K + j]);
    n += (size_t)snprintf(buf + n, len - n, "\n");
    return (int)n;
}

/* ---- Bayes::predict (bayes.cpp:16-284), restated; SURVEY section 8f-3 ----------------------------
 * Pinning: the genotype / NA tables are pinned (ref_luts.bin); the p-value uses
 * boost::math::gamma_p(0.5, x) upstream (bayes.cpp:205) -- Boost is absent here, so it is restated
 * as erf(sqrt(x)) (the same function; PARITY UNPINNED at the last digits of the p-value). */

/* bayes.cpp:93-122: g_k over markers [0, M) of a block in marker order (the reference's OpenMP
 * loop adds in an unspecified order; this is its one-thread order) */
void orc_predict_g(const uint8_t* bed, int M, int mbytes, const uint8_t* mask4, int im4,
                   const double* mave, const double* msig, const double* beta, double* g_k) {
    build_luts();
    for (int i = 0; i < im4 * 4; i++) g_k[i] = 0.0;
    for (int mrki = 0; mrki < M; mrki++) {
        const uint8_t* bedm = bed + (size_t)mrki * (size_t)mbytes;
        for (int j = 0; j < im4; j++)
            for (int k = 0; k < 4; k++) {
                const double val = (g_lut_a[bedm[j] * 4 + k] - mave[mrki]) * g_lut_b[bedm[j] * 4 + k] * g_lut_na[mask4[j] * 4 + k] * msig[mrki];
                g_k[j * 4 + k] += val * beta[mrki];
            }
    }
}

/* bayes.cpp:172-196: xtx, xty of every marker of the block */
void orc_assoc(const uint8_t* bed, int M, int mbytes, const uint8_t* mask4, int im4, const double* y_k,
               double* xtx_out, double* xty_out) {
    build_luts();
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (int mrki = 0; mrki < M; mrki++) {
        const uint8_t* bedm = bed + (size_t)mrki * (size_t)mbytes;
        double xtx = 0.0, xty = 0.0;
        for (int j = 0; j < im4; j++)
            for (int k = 0; k < 4; k++) {
                const double val = g_lut_a[bedm[j] * 4 + k] * g_lut_b[bedm[j] * 4 + k] * g_lut_na[mask4[j] * 4 + k];
                xtx += val * val;
                xty += val * y_k[j * 4 + k];
            }
        xtx_out[mrki] = xtx;
        xty_out[mrki] = xty;
    }
}

/* bayes.cpp:198-205: beta, tdist, se, pval from xtx, xty, sigma */
void orc_mlma_stats(double xtx, double xty, double sigma, double* beta, double* tdist, double* se, double* pval) {
    *beta = xty / xtx;
    *tdist = xty / sqrt(sigma * xtx);
    *se = *beta / *tdist;
    *pval = 1.0 - erf(sqrt(*tdist * *tdist * 0.5));        /* 1 - gamma_p(1/2, t^2/2) */
}

/* bayes.cpp:233-234: one .mlma record (123 bytes for ids of <= 20 characters) */
int orc_mlma_line(char* buf, size_t len, const char* id, int mglo, int rmglo, double beta, double tdist, double se, double pval) {
    return snprintf(buf, len, "%20s %8d %8d %20.15f %20.15f %20.15f %20.15f\n", id, mglo, rmglo, beta, tdist, se, pval);
}

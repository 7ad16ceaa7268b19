/*
 * oracle/gmrm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the per-marker Gibbs update hot path of
 * medical-genomics-group/gmrm (reference checkout: /root/reference, read-only).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped HIP path (gmrm_amd/) never links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - genotype / NA lookup tables: PINNED against the reference's own
 *     src/dotp_lut.hpp and src/na_lut.hpp (compiled in place by oracle/Makefile
 *     into oracle/_ref/, dumped into tests/golden/ref_luts.bin).
 *   - .csv / .bet / .cpn record layout: PINNED against the reference's own
 *     src/xfiles.cpp compiled in place (tests/golden/ref_xfiles_*).
 *   - arithmetic of dot_product / update_epsilon / offset_epsilon / sumsqr /
 *     marker statistics / the scalar Gibbs step: restated line by line from the
 *     cited sources; the reference ships no tests, golden vectors or fixtures for
 *     them and its translation units need Boost (absent here), so these are
 *     "parity unpinned" beyond the tables above.
 *   - random draws: Boost.Random (boost 1.76 per setup/Make.intel_ioampi:6) is not
 *     vendored and not installed: its published algorithms are restated in
 *     orc_rng_* below -- PARITY UNPINNED.
 *
 * Two summation modes exist for every reduction on the path:
 *   "ref"   : the reference's own loop order, single thread (bayes.cpp:756-763 ...).
 *   "canon" : the same terms accumulated exactly (2-level pre-rounded bins), which
 *             makes the result independent of summation order.  The reference's
 *             OpenMP reductions have no defined order, so
 *             "canon" is one admissible evaluation of the same expression; it is the
 *             mode the HIP kernels implement, so oracle and GPU agree bit for bit.
 */
#ifndef GMRM_ORACLE_H
#define GMRM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- lookup tables (src/dotp_lut.hpp, src/na_lut.hpp) ---- */
const double* orc_dotp_lut_a(void);   /* [1024] */
const double* orc_dotp_lut_b(void);   /* [1024] */
const double* orc_na_lut(void);       /* [64]   */

/* ---- per-call kernels, reference loop order ---- */
double orc_dot_product(const uint8_t* bedcol, const double* phen, int mbytes,
                       double mu, double sigma_inv);                   /* bayes.cpp:749-766 */
void   orc_update_epsilon(double* eps, const double* dbeta3, const uint8_t* bedcol,
                          const uint8_t* mask4, int im4);              /* phenotype.cpp:375-390 */
void   orc_offset_epsilon(double* eps, double offset, const uint8_t* mask4, int im4); /* phenotype.cpp:395-411 */
double orc_epsilon_sumsqr(const double* eps, int N);                   /* phenotype.cpp:251-261 */
double orc_epsilon_sigma(const double* eps, const uint8_t* mask4, int im4, int nonas); /* phenotype.cpp:448-457 */
void   orc_marker_stats(const uint8_t* bed, int N, int M, int mbytes, const uint8_t* mask4,
                        int nonas, double* mave, double* msig);        /* phenotype.cpp:525-550 */

/* ---- the same reductions, order-independent ("canon") ---- */
double orc_dot_product_canon(const uint8_t* bedcol, const double* phen, int mbytes,
                             double mu, double sigma_inv);
double orc_epsilon_sumsqr_canon(const double* eps, int N);
double orc_epsilon_sigma_canon(const double* eps, const uint8_t* mask4, int im4, int nonas);
void   orc_marker_stats_canon(const uint8_t* bed, int N, int M, int mbytes, const uint8_t* mask4,
                              int nonas, double* mave, double* msig);
void   orc_marker_counts(const uint8_t* bedcol, int mbytes, const uint8_t* mask4, int64_t cnt[4]);
void   orc_split2(double x, double* q1, double* q2);   /* the 2-level pre-rounding */
/* the canon residual lives on the grid 2^-44 (exact doubles; updates are exact additions) */
double orc_grid(double x);
void   orc_grid_array(double* x, int n);
void   orc_update_epsilon_canon(double* eps, const double* dbeta3, const uint8_t* bedcol,
                                const uint8_t* mask4, int im4);       /* phenotype.cpp:375-390 on the grid */
void   orc_offset_epsilon_canon(double* eps, double offset, const uint8_t* mask4, int im4);
double orc_exp(double x);                              /* the path's exp(), shared spec */

/* ---- phenotype preparation (phenotype.cpp:587-673) ---- */
/* y[N], isna[N] -> eps[4*im4] (centred, scaled, 0 at NA and in the tail), mask4[im4] */
void orc_phen_prepare(const double* y, const uint8_t* isna, int N,
                      double* eps, uint8_t* mask4, int* nonas);

/* ---- RNG spec (distributions.hpp:5-61; phenotype.cpp:314-323) ---- */
typedef struct { uint32_t mt[624]; int idx; } orc_rng;
void     orc_rng_seed(orc_rng* r, uint32_t seed);
uint32_t orc_rng_u32(orc_rng* r);
double   orc_rng_unif(orc_rng* r);
double   orc_rng_norm(orc_rng* r, double mean, double sigma2);
double   orc_rng_exponential(orc_rng* r);
double   orc_rng_gamma(orc_rng* r, double shape, double scale);
double   orc_rng_beta(orc_rng* r, double a, double b);
double   orc_rng_inv_scaled_chisq(orc_rng* r, double a, double b);
void     orc_rng_shuffle(orc_rng* r, int* v, int n);

/* ---- one phenotype's chain on one rank (Phenotype + Bayes state) ---- */
typedef struct orc_chain orc_chain;
orc_chain* orc_chain_create(int N, int M, int Mt, int S, int G, int K,
                            const uint8_t* bed_local, const double* eps0 /*4*im4*/,
                            const uint8_t* mask4, int nonas,
                            const int* group_index /*Mt*/, const double* cva /*G*K*/,
                            uint32_t seed, int rank, int shuffle, int mimic_hydra, int canon);
void orc_chain_destroy(orc_chain* c);
void orc_chain_init(orc_chain* c);                 /* bayes.cpp:322-335 */
void orc_chain_iterate(orc_chain* c, int it);      /* bayes.cpp:340-651, nranks == 1 */
/* pieces, for schedules that interleave ranks */
void orc_chain_prologue(orc_chain* c, int it);     /* bayes.cpp:348-368 */
double orc_chain_prologue_draw(orc_chain* c, int it);      /* :348-358, returns the drawn mu */
void   orc_chain_prologue_apply(orc_chain* c, double mu);  /* :358-367 */
void orc_chain_markers(orc_chain* c);              /* bayes.cpp:375-553, own markers only */
void orc_chain_markers_range(orc_chain* c, int first, int count);   /* a part of the visit order */
void orc_chain_local_sums(orc_chain* c);           /* bayes.cpp:565-568 (beta_sqn), cass stays local */
void orc_chain_epilogue(orc_chain* c);             /* bayes.cpp:590-651 */
/* the build's sweep-synchronous multi-rank schedule (DESIGN.md "Multi-GPU"); not the
 * reference's per-step exchange.  All chains share one phenotype and hold disjoint shards. */
void orc_ns_iterate(orc_chain** chains, int nranks, int it);
/* the reference's own per-step multi-rank schedule (bayes.cpp:374-553, 681-706): every rank keeps its own mu,
 * after every marker step all replicas apply the changed markers of all ranks in rank order. */
void orc_ps_iterate(orc_chain** chains, int nranks, int it);
/* orc_ns_iterate with the residual exchange every k marker positions (the build's --sync-every k; no upstream counterpart) */
void orc_nk_iterate(orc_chain** chains, int nranks, int it, int k);

/* getters (pointers stay owned by the chain) */
double* orc_chain_eps(orc_chain* c);
double* orc_chain_betas(orc_chain* c);
double* orc_chain_acum(orc_chain* c);
int*    orc_chain_comp(orc_chain* c);
int*    orc_chain_midx(orc_chain* c);
int*    orc_chain_cass(orc_chain* c);
int*    orc_chain_m0(orc_chain* c);
double* orc_chain_sigmag(orc_chain* c);
double* orc_chain_pi_est(orc_chain* c);
double* orc_chain_beta_sqn(orc_chain* c);
double* orc_chain_mave(orc_chain* c);
double* orc_chain_msig(orc_chain* c);
double  orc_chain_sigmae(orc_chain* c);
double  orc_chain_mu(orc_chain* c);
void    orc_chain_set_sigmae(orc_chain* c, double v);
int     orc_chain_m0_sum(orc_chain* c);
long    orc_chain_nupdates(orc_chain* c);
orc_rng* orc_chain_rng_d(orc_chain* c);
orc_rng* orc_chain_rng_m(orc_chain* c);

/* ---- output records (xfiles.cpp:6-47, xfiles.hpp:14-38) ---- */
int orc_csv_line(char* buf, size_t len, unsigned it, const double* sigmag, int G,
                 double sigmae, int m0_sum, const double* pi_est, int K);


/* Bayes::predict (bayes.cpp:16-284) */
void orc_predict_g(const uint8_t* bed, int M, int mbytes, const uint8_t* mask4, int im4,
                   const double* mave, const double* msig, const double* beta, double* g_k);
void orc_assoc(const uint8_t* bed, int M, int mbytes, const uint8_t* mask4, int im4, const double* y_k,
               double* xtx_out, double* xty_out);
void orc_mlma_stats(double xtx, double xty, double sigma, double* beta, double* tdist, double* se, double* pval);
int  orc_mlma_line(char* buf, size_t len, const char* id, int mglo, int rmglo, double beta, double tdist, double se, double pval);

#ifdef __cplusplus
}
#endif
#endif

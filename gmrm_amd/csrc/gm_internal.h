// gm_internal.h -- declarations shared by the translation units of libgmrm_hip
// (ops.hip: per-call kernels, sweep.hip: the persistent marker-loop kernel,
//  capi.cpp: context + C ABI, sampler.cpp: host-side Bayes::process mirror).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace gm {

constexpr int SW_TPB  = 256;   // threads per workgroup of the sweep kernel (4 wavefronts)
constexpr int SW_VMAX = 512;   // exchanged values per batch: 4 per marker (sa1,sa2,sb1,sb2) or 2 per marker + 2 per batch in the
                               // no-missing-genotype layout (<= 240 markers), plus one per marker behind a crossed stop
constexpr int NSTOP = 2;       // markers with a non-zero effect the walk may cross inside one batch (sweep.hip, "continuation")
constexpr int KMAX = 8;
constexpr int GMAX = 64;

// Arguments of the persistent sweep kernel (one chain = one phenotype on one GPU).
struct SweepArgs {
    int N, M, W, Wpad, G, K;
    size_t stride;                 // bytes per marker column in HBM (ceil(N/4) padded to 16)
    const uint8_t* bed;            // [M][stride] 2-bit genotypes in the device code (gm_common.h)
    const uint8_t* namask2;        // [stride] 2-bit NA mask (11 = phenotype present)
    const int* order;              // [M] visit order (local marker ids)
    const int* group;              // [M] group of each local marker
    const double* mave;            // [M]
    const double* msig;            // [M]
    const uint8_t* nomiss;         // [M] 1: the marker has no missing genotype among the phenotyped individuals
    const double* betas_in;        // [M] effects before this sweep
    // the sampling step's inputs in visit order (k_order_inputs, run in front of the sweep): entry p = those of marker order[p]
    const int* o_g; const double* o_beta; const double* o_mave; const double* o_msig; const uint8_t* o_nm;
    double* betas_out;             // [M] effects after this sweep
    int* comp;                     // [M]
    double* acum;                  // [M]
    double* eps;                   // [4*stride] residual, in/out
    const double* sigmag;          // [G]
    const double* denom;           // [G*K], entry k (k>=1) = (N-1) + sigmae/sigmag * cvai[k]
    const double* logpi;           // [G*K] log(pi_est)
    const double* mhl;             // [G*K] -0.5*log(sigmag/sigmae*(nonas-1)*cva + 1)
    double sigmae, inv2sige, nm1;
    uint32_t* rng_state;           // [624] in/out
    int* rng_index;                // in/out
    int* cass;                     // [G*K] out
    long long* stats;              // [4] out: updates, batches, max batch, spare
    int* err;                      // out: 0 ok
    double* P;                     // [SW_VMAX][Wpad] per-workgroup partial sums
    double* Tt;                    // [SW_VMAX] totals
    unsigned* cnt;                 // [96] arrival counters / abort word, zeroed per launch
    int batch_init;
    int totals_delay;              // long-batch kernel: s_sleep units (64 clocks) before wavefront 0's first look at the totals
    int totals_delay2;             // ... the same in the other kernels
    int direct_pub;                // short-batch kernels: a batch of four tiles or more publishes its packed rows from the tile passes (A/B knob)
    int tile_trim;                 // 1: a batch that would give one wavefront a tile more than the others drops its last, partial tile
    int bcap;                      // long-batch kernels: longest batch (0: the kernel's own cap; env GMRM_BATCH_CAP)
    unsigned long long* trace;     // diagnostic build: [W][64][8] wall-clock stamps of rounds 2000..2063, or null
    int nb_factor16;               // next batch >= nb_factor16/16 x the run-length EMA, as a power of two (default 24 = 1.5x)
    int reduce4;                   // kernels without long batches: 1 (default) all four wavefronts take the reduce role, 0 (GMRM_REDUCE_W0=1) wavefront 0 alone
    int screen_min_run16;          // the sampling screen is tried when the run-length EMA (1/16 marker) is at least this
    int miss_mode;                 // markers with a missing genotype among the phenotyped individuals: 0 none, 1 some, 2 all
    unsigned long long spin_ticks; // every grid-wide wait gives up after this many s_memrealtime ticks (100 MHz)
    // LDS carve of this launch (sweep.hip, carve_for: depends on G, K and the kernel): byte offsets of the component counts, the
    // per-group tables, the planes and the LDS tiles; the number of LDS tile slots (with the multiplier of `% nl`) and the length of
    // the tile window that follows from it
    int lds_cass, lds_tab, lds_pln, lds_ring, nl, win;
    unsigned nl_magic;
    int cross;                     // > 0: the walk may cross a marker whose effect was non-zero when at least cross/16 of the batch
                                   // lies behind it (a crossing costs about half a round; 0: never)
    int long_mixed;                // the per-marker layout (some markers with missing genotypes) on the long-batch kernel: sparse model, few such markers
    int long_cross;                // with cross > 0 in the layout without missing genotypes: the long-batch kernel that crosses stops
                                   // (sparse models) instead of the short-batch one (dense models)
};

// sweep.hip
hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st, int grid);   // grid == a.W (tests may launch one short)
hipError_t sweep_occupancy(int R, int* blocks_per_cu);                          // resident workgroups per CU (runtime query)
int  sweep_pick_R(size_t stride, int max_wg, int* W_out);   // bytes per thread, or -1
size_t sweep_lds_bytes();

// ops.hip
hipError_t launch_dot(const uint8_t* col, const uint8_t* namask2, const double* eps, size_t stride,
                      double* out4 /*zeroed*/, hipStream_t st);
hipError_t launch_update(double* eps, const uint8_t* col, const uint8_t* namask2, size_t stride,
                         double v0, double v1, double v2, double v3, hipStream_t st);
hipError_t launch_offset(double* eps, const uint8_t* namask2, size_t stride, double off, hipStream_t st);
hipError_t launch_sumsq(const double* eps, const uint8_t* namask2 /*or null*/, size_t n, double* out2 /*zeroed*/,
                        double* outmax, hipStream_t st);
hipError_t launch_marker_stats(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, int nonas,
                               double* mave, double* msig, uint8_t* nomiss, hipStream_t st);
hipError_t launch_order_inputs(const int* order, int count, const int* group, const double* betas, const double* mave, const double* msig,
                               const uint8_t* nomiss, int* o_g, double* o_beta, double* o_mave, double* o_msig, uint8_t* o_nm, hipStream_t st);
hipError_t launch_recode(uint8_t* bed, size_t nbytes, int back, hipStream_t st);   // .bed code <-> device code, in place
hipError_t launch_synth(uint8_t* bed, size_t stride, int N, int M, int S, uint64_t seed,
                        double maf, double miss, int ld_block, double ld_keep, hipStream_t st);
hipError_t launch_predict_g(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, const double* mave,
                            const double* msig, const double* beta, double* g, hipStream_t st);
size_t predict_workspace_bytes(size_t stride, int M);
hipError_t launch_predict_g_mfma(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, const double* mave,
                                 const double* msig, const double* beta, double* g, void* ws, int dirty, hipStream_t st);
size_t assoc_workspace_bytes(size_t stride);
hipError_t launch_assoc(const uint8_t* bed, const uint8_t* namask2, size_t stride, int M, const double* y,
                        double* xtx, double* xty, void* ws, hipStream_t st);
hipError_t launch_selftest(int op, const double* x, double* y, int n, hipStream_t st);
hipError_t launch_delta_export(const double* eps, const double* start, double* q, size_t n4, hipStream_t st);
hipError_t launch_delta_import(double* eps, const double* start, const double* q, size_t n4, hipStream_t st);

}  // namespace gm

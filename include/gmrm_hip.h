/*
 * gmrm_hip.h -- C ABI of libgmrm_hip.so: the MI355X (gfx950) implementation of gmrm's
 * per-marker Gibbs update hot path.
 *
 * The reference has no plugin / FFI boundary: the path sits behind ordinary C++ member
 * calls made from Bayes::process() (reference src/bayes.cpp:318-677).  Each entry point
 * below names the member it replaces (file:line relative to /root/reference/).  Plain
 * pointers and sizes only; no torch / HIP types.  INTEGRATION.md shows the call-site
 * patch a gmrm maintainer would apply.
 *
 * Conventions
 *   - every function returns 0 on success, a negative GMRM_E* code on failure;
 *     gmrm_last_error() returns a message for the calling thread's last failure.
 *   - the reference aborts (MPI_Abort / exit(1), src/utilities.cpp:15-30) on failure and
 *     has no return codes; callers that want that behaviour abort on a non-zero return.
 *   - the caller owns every host buffer; the context owns every device buffer, except
 *     where a function takes a `dev_*` pointer (caller-owned device memory, e.g. a
 *     torch tensor's data_ptr() used for the RCCL exchange).
 *   - one context = one GPU = one contiguous block of markers (src/bayes.cpp:903-925)
 *     x T phenotypes.  Not thread-safe per context (as the reference's objects).
 *   - there is NO CPU fallback: without a HIP device every compute entry fails with
 *     GMRM_ENODEV.
 */
#ifndef GMRM_HIP_H
#define GMRM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GMRM_OK        0
#define GMRM_EINVAL   -1   /* bad argument / shape */
#define GMRM_ENODEV   -2   /* no usable HIP device */
#define GMRM_EHIP     -3   /* HIP runtime error (message in gmrm_last_error) */
#define GMRM_ENOMEM   -4
#define GMRM_ESTATE   -5   /* call order violated (e.g. sweep before marker_stats) */
#define GMRM_EKERNEL  -6   /* a device-side check failed (timeout word, RNG window, range) */
#define GMRM_EIO      -7

#define GMRM_KMAX      8   /* mixture components per group supported on device */

typedef struct gmrm_ctx gmrm_ctx;
typedef struct gmrm_sampler gmrm_sampler;

const char* gmrm_last_error(void);
int  gmrm_abi_version(void);
int  gmrm_device_count(void);            /* 0 when no HIP device is visible */

/* ------------------------------------------------------------------------------------
 * Context: Bayes ctor + setup_processing() (src/bayes.hpp:20-56, src/bayes.cpp:774-812).
 *   N        individuals (invariant over ranks)        Bayes::N
 *   M        markers held by this context              Bayes::M
 *   Mt, S    total markers / first global marker       Bayes::Mt, Bayes::S
 *   T        phenotypes                                PhenMgr::phens.size()
 * ---------------------------------------------------------------------------------- */
int gmrm_ctx_create(gmrm_ctx** out, int device, int N, int M, int Mt, int S, int T);
int gmrm_ctx_destroy(gmrm_ctx* ctx);
int gmrm_ctx_sync(gmrm_ctx* ctx);                      /* wait for every stream of ctx */
/* Launch geometry of the persistent marker-loop kernel chosen for this context (no reference
 * counterpart: upstream's parallelism is `#pragma omp parallel for` inside each call).  The kernel's
 * workgroups wait for each other, so all of them must be resident at once: gmrm_ctx_create checks
 * W * conc against the occupancy query and refuses geometries that cannot be co-resident. */
typedef struct gmrm_geometry {
    int R;                 /* genotype bytes of a column per thread (1, 2 or 4)                  */
    int W;                 /* workgroups of one chain's launch (one per CU)                      */
    int conc;              /* chains of this context that sweep side by side                     */
    int num_cu;            /* compute units of the device                                        */
    int max_resident_wg;   /* occupancy query x num_cu for the sweep kernel                      */
    int hw_queues;         /* GPU_MAX_HW_QUEUES as the environment holds it (the library asks for 8 */
                           /* when it is loaded: a persistent launch holds its queue; chains beyond */
                           /* the queues run one after another).  0: unset.  Without effect if the  */
                           /* HIP runtime had been initialised before the library was loaded.       */
} gmrm_geometry;
int gmrm_ctx_geometry(const gmrm_ctx* ctx, gmrm_geometry* out);

/* Genotypes: Bayes::load_genotype (src/bayes.cpp:867-900).  `cols` is marker-major,
 * ceil(N/4) bytes per marker, PLINK 2-bit codes LSB first (no 3-byte magic). */
int gmrm_upload_bed(gmrm_ctx* ctx, const uint8_t* cols, size_t first_marker, size_t n_markers);

/* .bed ingest at scale: the whole of Bayes::load_genotype
 * (src/bayes.cpp:867-900, chunked MPI_File_read_at of src/utilities.hpp:28-53) for this context's
 * marker block, straight from the file: markers [file_first_marker, file_first_marker + M) of the
 * PLINK .bed at `path` go to device markers [0, M).  The 3 magic bytes and the file size are
 * checked (the reference checks neither).  `nthreads` reader threads (1..64) fill pinned buffers
 * with parallel pread()s while the previous chunk is copied.  `stats` may be NULL. */
typedef struct gmrm_ingest_stats {
    size_t bytes;          /* genotype bytes moved */
    double seconds;        /* wall time of the call */
    double read_seconds;   /* part of it spent inside pread() (not overlapped with the copies) */
    int threads;
    size_t chunk_bytes;
} gmrm_ingest_stats;
int gmrm_load_bed_file(gmrm_ctx* ctx, const char* path, size_t file_first_marker, int nthreads,
                       gmrm_ingest_stats* stats);
int gmrm_download_bed(gmrm_ctx* ctx, uint8_t* cols, size_t first_marker, size_t n_markers);
/* Synthetic genotypes generated on the device, keyed by (seed, global marker, individual):
 * copies of A1 ~ Binomial(2, maf) (example/data_sim.R:15), code 01 with prob. miss_rate. */
int gmrm_synth_bed(gmrm_ctx* ctx, uint64_t seed, double maf, double miss_rate);
/* ... with linkage disequilibrium (no upstream counterpart; example/data_sim.R draws every marker independently): markers in
 * blocks of ld_block consecutive global indices, inside a block a haplotype copies its allele from the marker before it with
 * probability ld_keep (r ~ ld_keep between neighbours, ld_keep^k at distance k), blocks independent.  ld_block <= 1: as above. */
int gmrm_synth_bed_ld(gmrm_ctx* ctx, uint64_t seed, double maf, double miss_rate, int ld_block, double ld_keep);

/* Phenotype::read_file after tokenising (src/phenotype.cpp:587-673), host-only: y[N] values,
 * isna[N] flags ("NA" tokens) -> eps[4*ceil(N/4)] centred and scaled to unit variance (0 at
 * NA and in the tail), mask4[ceil(N/4)] (bit k = individual 4i+k present), *nonas. */
int gmrm_phen_prepare(const double* y, const uint8_t* isna, int N, double* eps, uint8_t* mask4, int* nonas);

/* Phenotype t: Phenotype ctor + read_file (src/phenotype.cpp:18-55,587-673): the centred,
 * scaled residual eps[4*ceil(N/4)] (0 at NA and in the tail), mask4[ceil(N/4)], nonas. */
int gmrm_upload_trait(gmrm_ctx* ctx, int t, const double* eps, const uint8_t* mask4, int nonas);
int gmrm_download_eps(gmrm_ctx* ctx, int t, double* eps);
int gmrm_upload_eps(gmrm_ctx* ctx, int t, const double* eps);

/* ------------------------------------------------------------------------------------
 * The reference's per-call kernels, one entry each.
 * ---------------------------------------------------------------------------------- */
/* PhenMgr::compute_markers_statistics (src/phenotype.cpp:466-556), phenotype t. */
int gmrm_marker_stats(gmrm_ctx* ctx, int t);
int gmrm_get_marker_stats(gmrm_ctx* ctx, int t, double* mave, double* msig);
int gmrm_set_marker_stats(gmrm_ctx* ctx, int t, const double* mave, const double* msig);
/* double Bayes::dot_product(mloc, phen, mu, sigma_inv) (src/bayes.hpp:66, src/bayes.cpp:709-770) */
int gmrm_dot(gmrm_ctx* ctx, int t, int mloc, double mu, double sigma_inv, double* num);
/* void Phenotype::update_epsilon(const double* dbeta[3], bed) (src/phenotype.hpp:153,
 * src/phenotype.cpp:326-393); dbeta3 = {dbeta, mave, msig}; the column is marker mloc. */
int gmrm_update_eps(gmrm_ctx* ctx, int t, int mloc, const double* dbeta3);
/* Bayes::update_epsilon(counts, dbetas, bed) for one sender (src/bayes.cpp:681-706): the same update with the
 * column of ANOTHER shard -- marker mloc of context src, which upstream receives through MPI_Allgatherv
 * (src/bayes.cpp:537-541).  Copied device to device when src sits on another GPU. */
int gmrm_update_eps_from(gmrm_ctx* ctx, int t, gmrm_ctx* src, int mloc, const double* dbeta3);
/* void Phenotype::offset_epsilon(double) (src/phenotype.hpp:108, src/phenotype.cpp:395-411) */
int gmrm_offset_eps(gmrm_ctx* ctx, int t, double offset);
/* double Phenotype::epsilon_sumsqr() (src/phenotype.hpp:154, src/phenotype.cpp:251-261) */
int gmrm_sumsqr(gmrm_ctx* ctx, int t, double* out);
/* void Phenotype::update_epsilon_sigma() (src/phenotype.hpp:110, src/phenotype.cpp:432-459);
 * *sigmae receives the value set_sigmae() would. */
int gmrm_eps_sigma(gmrm_ctx* ctx, int t, double* sigmae);

/* Bayes::predict building blocks (src/bayes.cpp:16-284).
 * gmrm_predict_g: g_i = sum over this context's markers, in marker order, of
 *   ((a - mave) * b * na * msig) * beta_local[m]          (bayes.cpp:93-122, g_k)
 *   beta_local: M posterior-mean effects of this block (host), g: N doubles (host, out).
 * gmrm_assoc: per marker xtx = sum (a*b*na)^2, xty = sum a*b*na*yk_i   (bayes.cpp:172-196)
 *   yk: N doubles (host; NULL = the phenotype's residual as it stands), xtx/xty: M doubles (host, out).
 * Both need gmrm_upload_trait + gmrm_marker_stats for phenotype t. */
int gmrm_predict_g(gmrm_ctx* ctx, int t, const double* beta_local, double* g);
int gmrm_assoc(gmrm_ctx* ctx, int t, const double* yk, double* xtx, double* xty);

/* ------------------------------------------------------------------------------------
 * Fused marker loop: the body of `for (mrki...)` in Bayes::process for this context's
 * markers and phenotype t (src/bayes.cpp:375-553 -> dot_product, Gibbs step 396-492,
 * update_epsilon 681-706) as ONE persistent kernel launch.
 * ---------------------------------------------------------------------------------- */
typedef struct gmrm_sweep_in {
    int G, K;
    const int*    order;       /* [M] shuffled local marker indices (Phenotype::midx)       */
    const double* sigmag;      /* [G]                                                        */
    const double* pi_est;      /* [G*K]                                                      */
    const double* cva;         /* [G*K] mixture variances (Options::cva)                     */
    double        sigmae;
    uint32_t      rng_state[624];   /* Distributions dist_d: mt19937 state words ...          */
    int           rng_index;        /* ... and position, 0..624                               */
    /* A PART of a sweep (this build only: --sync-every k, residual exchange every k markers): positions [first, first +
     * count) of `order`; count == 0 means the whole order.  The parts of one sweep are launched in order of position
     * with the same `order`, each continuing the RNG stream the previous one returned; the component counts
     * accumulate on the device over the parts, and the new effects become current with the part that ends at M. */
    int           first, count;
} gmrm_sweep_in;

typedef struct gmrm_sweep_out {
    int*     cass;             /* [G*K] component counts of this sweep (Phenotype::cass)     */
    uint32_t rng_state[624];
    int      rng_index;
    long long n_updates;       /* visits with dbeta != 0                                     */
    long long n_batches;       /* grid-wide synchronisation rounds                           */
    double   device_ms;        /* kernel time, HIP events on the launch stream               */
    long long n_planned_stops; /* rounds that ended at a marker whose effect was non-zero    */
                               /* before the visit (a stop known in advance)                 */
    long long n_stale_dots;    /* dot products computed behind a stop and thrown away        */
    long long n_fast_batches;  /* batches exchanged in the 2-value layout (all their markers */
                               /* free of missing genotypes among phenotyped individuals)    */
    long long n_crossed_stops; /* residual updates the walk went past inside a round (the    */
                               /* dot products behind them were patched exactly)             */
    long long n_screen_tries;  /* passes of 64 markers in which the cheap certain bound was  */
                               /* tried in front of the exact probabilities (sweep.hip)      */
    long long n_screened_passes; /* ... and held for every lane: the exact code was skipped  */
} gmrm_sweep_out;

/* Per-marker group labels (Bayes::group_index restricted to [S, S+M)), shared by all t. */
int gmrm_set_groups(gmrm_ctx* ctx, const int* group_local);
int gmrm_sweep_launch(gmrm_ctx* ctx, int t, const gmrm_sweep_in* in);   /* asynchronous */
int gmrm_sweep_finish(gmrm_ctx* ctx, int t, gmrm_sweep_out* out);       /* waits, collects */
/* Per-marker chain state of phenotype t (Phenotype::betas / comp / acum).  acum is the reference's scratch of ONE marker
 * step (bayes.cpp:445-474 writes and reads it inside the step, nothing reads it later): the per-step entries
 * (gmrm_sampler_step) keep it, the sweep kernel does not store it. */
int gmrm_get_betas(gmrm_ctx* ctx, int t, double* betas);
int gmrm_get_comp(gmrm_ctx* ctx, int t, int* comp);
int gmrm_get_acum(gmrm_ctx* ctx, int t, double* acum);
int gmrm_set_betas(gmrm_ctx* ctx, int t, const double* betas);
int gmrm_set_comp(gmrm_ctx* ctx, int t, const int* comp);
int gmrm_set_acum(gmrm_ctx* ctx, int t, const double* acum);

/* ------------------------------------------------------------------------------------
 * Multi-GPU residual exchange (replaces the per-step MPI_Allgatherv of src/bayes.cpp:
 * 500-547 with one exchange per sweep; DESIGN.md "Multi-GPU").  The caller all-reduces
 * (sum, f64) the 2*n4 doubles between the two calls, e.g. torch.distributed / RCCL on a
 * tensor whose data_ptr() is `dev_q`.  n4 = 4*ceil(N/4).
 * ---------------------------------------------------------------------------------- */
int gmrm_eps_snapshot(gmrm_ctx* ctx, int t);                         /* eps_start = eps      */
int gmrm_eps_delta_export(gmrm_ctx* ctx, int t, double* dev_q);      /* split2(eps-eps_start) */
int gmrm_eps_delta_import(gmrm_ctx* ctx, int t, const double* dev_q);/* eps = start+(q1+q2)  */

/* ------------------------------------------------------------------------------------
 * Device arithmetic self-test (not on the reference's path): evaluates, on the GPU, the
 * building blocks whose bit-for-bit agreement with the host the parity claim rests on.
 *   op 0: y = exp_(x) (the path's exp)      op 1: y = sqrt(x)       op 2: y = 1.0 / x
 *   op 3: y[i] = i-th normal_distribution(0,1) draw from mt19937(seed = (uint32)x[0]) (n <= 65536)
 *   op 4: y[2i], y[2i+1] = split2(x[i])  (y holds 2n doubles)
 * ---------------------------------------------------------------------------------- */
int gmrm_selftest_math(int device, int op, const double* x, double* y, int n);
/* The marker shuffle of Phenotype::shuffle_midx (src/phenotype.cpp:314-323) as the host sampler performs it: v = 0..n-1
 * shuffled with an mt19937 seeded `seed` (host only, no device needed).  tests/ hold it to permutations produced by
 * libstdc++'s own std::random_shuffle + std::mt19937 (tests/golden/stl_shuffle.txt.gz). */
int gmrm_selftest_shuffle(uint32_t seed, int n, int* v);

/* ------------------------------------------------------------------------------------
 * Host-side sampler: Bayes::process() around the marker loop (src/bayes.cpp:318-371,
 * 556-669) -- prologue draws, shuffle, hyper-parameter updates -- on the RNG spec of
 * DESIGN.md, driving the context above.  One object per context, all T phenotypes.
 * ---------------------------------------------------------------------------------- */
typedef struct gmrm_sampler_opts {
    uint32_t seed;             /* --seed                                                     */
    int      rank, nranks;     /* marker-shard index / count (MPI rank / size upstream)      */
    int      shuffle;          /* --shuffle-markers                                          */
    int      mimic_hydra;      /* --mimic-hydra                                              */
    int      G, K;
    const double* cva;         /* [G*K] from --group-mixture-file                            */
    const int*    group_index; /* [Mt]  from --group-index-file (all markers)                */
} gmrm_sampler_opts;

typedef struct gmrm_hyper {    /* one phenotype's state after an iteration                   */
    double sigmae, mu;
    int    m0_sum;
    double sigmag[64];
    double pi_est[64 * GMRM_KMAX];
    long long n_updates, n_batches;
    double sweep_device_ms;
    long long n_planned_stops, n_stale_dots, n_fast_batches, n_crossed_stops;   /* see gmrm_sweep_out */
    long long n_screen_tries, n_screened_passes;
} gmrm_hyper;

int gmrm_sampler_create(gmrm_sampler** out, gmrm_ctx* ctx, const gmrm_sampler_opts* opts);
int gmrm_sampler_destroy(gmrm_sampler* s);
int gmrm_sampler_init(gmrm_sampler* s);                                  /* bayes.cpp:322-335 */
/* single shard (nranks == 1): one full iteration for every phenotype */
int gmrm_sampler_iterate(gmrm_sampler* s, int it);
/* sharded: the same iteration cut at its exchange points (see DESIGN.md "Multi-GPU") */
int gmrm_sampler_draw_mu(gmrm_sampler* s, int it, double* mu_drawn /*[T]*/);
int gmrm_sampler_begin_sweep(gmrm_sampler* s, const double* mu_use /*[T]*/);  /* launches  */
/* gmrm_sampler_begin_sweep = gmrm_sampler_launch_sweep (bayes.cpp:358-367 with the adopted mu and the launch, nothing
 * else) + gmrm_sampler_preshuffle (the NEXT iteration's marker shuffle, on the idle host while the GPU sweeps).  A host
 * that drives several shards launches all of them before it shuffles for any (gmrm_group_iterate does). */
int gmrm_sampler_launch_sweep(gmrm_sampler* s, const double* mu_use /*[T]*/);
int gmrm_sampler_preshuffle(gmrm_sampler* s);
/* A sweep in PARTS (this build only; `--sync-every k`, 1 < k < M: marker shards exchange their residuals every k markers
 * instead of once per sweep -- closer to the reference's exchange after every marker, bayes.cpp:495-553, at k launches per
 * sweep): gmrm_sampler_begin_parts = bayes.cpp:358-367 with the adopted mu, no launch; then, in order of position,
 * gmrm_sampler_launch_part(first, count) (asynchronous; with several shards the residual is remembered first, as
 * gmrm_eps_snapshot) and gmrm_sampler_finish_part (waits) -- the caller exchanges the residual deltas of the shards between
 * a finish and the next launch (gmrm_eps_delta_export / _import) --, then gmrm_sampler_end_sweep and _epilogue as usual.
 * One part [0, M) is gmrm_sampler_launch_sweep; on one shard any partition gives the same chain, bit for bit. */
int gmrm_sampler_begin_parts(gmrm_sampler* s, const double* mu_use /*[T]*/);
int gmrm_sampler_launch_part(gmrm_sampler* s, int first, int count);
int gmrm_sampler_finish_part(gmrm_sampler* s);
int gmrm_sampler_end_sweep(gmrm_sampler* s, int* cass /*[T*G*K]*/, double* beta_sqn /*[T*G]*/);
int gmrm_sampler_epilogue(gmrm_sampler* s, const int* cass, const double* beta_sqn);
int gmrm_sampler_adopt(gmrm_sampler* s, int t, const double* sigmag, const double* pi_est, double sigmae);
int gmrm_sampler_get(gmrm_sampler* s, int t, gmrm_hyper* out);
/* The reference's per-step schedule (src/bayes.cpp:374-553 as several MPI tasks run it), for callers that want
 * the chain of `mpiexec -n R gmrm` rather than the sweep kernel's speed: between gmrm_sampler_begin_steps
 * (bayes.cpp:358-367 with this shard's own mu, nothing launched) and gmrm_sampler_end_steps (effects back to the
 * device, local cass / beta_sqn as gmrm_sampler_end_sweep), gmrm_sampler_step(mrki) draws the effect of this
 * shard's mrki-th marker for every phenotype (one gmrm_dot + the host restatement of bayes.cpp:396-492) and
 * returns *mloc and dbeta3[3t..3t+2] = {dbeta, mave, msig}, zero when nothing changed or mrki >= M.  The residual
 * is not touched: the caller applies every shard's changed marker to every replica in shard order with
 * gmrm_update_eps_from (bayes.cpp:681-706).  gmrm_group_iterate_steps does all of that. */
int gmrm_sampler_begin_steps(gmrm_sampler* s, const double* mu_use /*[T]*/);
int gmrm_sampler_step(gmrm_sampler* s, int mrki, int* mloc, double* dbeta3 /*[3T]*/);
int gmrm_sampler_end_steps(gmrm_sampler* s, int* cass /*[T*G*K]*/, double* beta_sqn /*[T*G]*/);
/* Leave a per-step sweep that cannot be completed (an error in _begin_steps / _step, here or in another shard): the
 * residual gets mu back, the device keeps the effects of the last completed sweep.  gmrm_sampler_begin_sweep /
 * _begin_steps / _save / _load return GMRM_ESTATE while a per-step sweep is open or a kernel sweep is in flight. */
int gmrm_sampler_abort_steps(gmrm_sampler* s);
/* one .csv record of phenotype t as write_ofile_csv formats it (src/xfiles.cpp:17-42) */
int gmrm_sampler_csv_line(gmrm_sampler* s, int t, int it, char* buf, size_t len);
/* Checkpoint / restart (no reference counterpart: Bayes::process deletes its outputs at start,
 * src/bayes.cpp:323, and cannot resume).  gmrm_sampler_save writes, after iteration `it`, everything iteration
 * it + 1 reads (residual, effects, components, visit order, hyper-parameters, both RNG streams) to `path`;
 * gmrm_sampler_load restores it into a sampler created with the same data, options and seed and returns the
 * iteration in *it.  A run resumed this way continues the chain bit for bit. */
int gmrm_sampler_save(gmrm_sampler* s, const char* path, int it);
int gmrm_sampler_load(gmrm_sampler* s, const char* path, int* it);

/* ------------------------------------------------------------------------------------
 * Several marker shards in ONE host process (one context + sampler per GPU): the per-sweep
 * exchange that replaces the MPI calls of Bayes::process (src/bayes.cpp:495-553,575-588) -- one
 * all-reduce of the residual per phenotype and sweep (RCCL ncclAllReduce over xGMI when
 * `want_rccl` and every shard has its own device; staged through host memory otherwise).
 * ctxs[r] / smps[r]: shard r, created with opts.rank = r, opts.nranks = n.  The group borrows them.
 * gmrm_group_iterate = one iteration of every phenotype on every shard.
 * ---------------------------------------------------------------------------------- */
typedef struct gmrm_group gmrm_group;
int gmrm_group_create(gmrm_group** out, int n, gmrm_ctx** ctxs, gmrm_sampler** smps, int G, int K, int want_rccl);
int gmrm_group_uses_rccl(const gmrm_group* g);
int gmrm_group_iterate(gmrm_group* g, int it);
/* ... with the residual exchange every k marker positions instead of once per sweep (`--sync-every k`, 1 < k < M; this
 * build only -- upstream exchanges after every marker step: gmrm_group_iterate_steps).  k >= M: gmrm_group_iterate's chain. */
int gmrm_group_iterate_parts(gmrm_group* g, int it, int k);
/* the same iteration on the reference's per-step schedule (see gmrm_sampler_step): the chain of
 * `mpiexec -n <shards> gmrm`, at the reference's cost of one exchange per marker step */
int gmrm_group_iterate_steps(gmrm_group* g, int it);
int gmrm_group_destroy(gmrm_group* g);
/* one-rank exercise of the RCCL entry points the group uses (what a one-GPU box can check) */
int gmrm_rccl_selftest(int device);

#ifdef __cplusplus
}
#endif
#endif

// sampler.cpp -- host-side mirror of Bayes::process() (reference src/bayes.cpp:318-677)
// around the marker loop: initial draws (322-335), per-iteration prologue (348-368), the
// hyper-parameter updates (562-651) and the .csv record (src/xfiles.cpp:6-47).  The marker
// loop itself (375-553) is the persistent HIP kernel behind gmrm_sweep_launch/finish.
// All draws follow the RNG spec of gm_rng.h on the reference's two streams per phenotype
// (dist_m: shuffling, dist_d: everything else; seeds src/bayes.cpp:796-803).
#include "../../include/gmrm_hip.h"
#include "gm_common.h"
#include "gm_rng.h"
#include "gm_internal.h"
#include "gm_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

namespace {

const double V0E = 0.0001, S02E = 0.0001, V0G = 0.0001, S02G = 0.0001;   // bayes.hpp:14-17

struct Chain {                       // class Phenotype, the host-resident part
    gm::Mt19937 dist_m, dist_d;
    std::vector<int> midx, cass, m0;
    std::vector<double> sigmag, pi_est, beta_sqn, betas;
    double sigmae = 0.0, mu = 0.0, epssum = 0.0;
    long long n_updates = 0, n_batches = 0, n_planned = 0, n_stale = 0, n_fastb = 0, n_cross = 0, n_scrt = 0, n_scr = 0;
    double sweep_ms = 0.0;
    bool preshuffled = false;        // midx already holds the NEXT iteration's order (shuffled while the GPU swept)
    // host copies the per-step schedule works on (gmrm_sampler_begin_steps .. _end_steps)
    std::vector<int> comp;
    std::vector<double> acum, mave, msig;
    bool stepping = false;
    long long steps_taken = 0;       // gmrm_sampler_step calls of the open per-step sweep
    bool abandoned = false;          // a per-step sweep was left after steps had been taken: effects and residual may no longer agree
};

const char CKP_MAGIC[8] = {'G', 'M', 'R', 'M', 'C', 'K', 'P', '1'};                // checkpoint files (gmrm_sampler_save / _load)
template <class T> bool put(FILE* f, const T* p, size_t n) { return std::fwrite(p, sizeof(T), n, f) == n; }
template <class T> bool get(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

}  // namespace

struct gmrm_sampler {
    gmrm_ctx* ctx = nullptr;
    uint32_t seed = 0;
    int rank = 0, nranks = 1, shuffle = 1, mimic_hydra = 0, G = 0, K = 0;
    std::vector<double> cva, pi_prior;
    std::vector<int> group_index, mtotgrp;
    std::vector<Chain> ch;
};

using gm::fail;

extern "C" {

int gmrm_selftest_shuffle(uint32_t seed, int n, int* v) {
    if (!v || n < 0) return gm::fail(GMRM_EINVAL, "bad argument");
    gm::Mt19937 e;
    e.seed(seed);
    for (int i = 0; i < n; i++) v[i] = i;
    gm::shuffle(e, v, n);
    return GMRM_OK;
}

int gmrm_sampler_create(gmrm_sampler** out, gmrm_ctx* ctx, const gmrm_sampler_opts* o) {
    if (!out || !ctx || !o || !o->cva || !o->group_index) return fail(GMRM_EINVAL, "null argument");
    *out = nullptr;
    const int G = o->G, K = o->K;
    if (G < 1 || G > gm::GMAX || K < 2 || K > gm::KMAX) return fail(GMRM_EINVAL, "G or K outside the supported range (G<=64, 2<=K<=8)");
    if (o->rank < 0 || o->nranks < 1 || o->rank >= o->nranks) return fail(GMRM_EINVAL, "bad rank / nranks");
    if (o->mimic_hydra && ctx->T > 1) return fail(GMRM_EINVAL, "with mimic_hydra only a single phenotype can be processed"); // options.cpp:216-219
    for (int i = 0; i < ctx->Mt; i++)
        if (o->group_index[i] < 0 || o->group_index[i] >= G) return fail(GMRM_EINVAL, "group index outside [0, G)");
    for (int g = 0; g < G; g++) {                                  // options.cpp:266-284
        if (o->cva[g * K] != 0.0) return fail(GMRM_EINVAL, "first element of a group mixture must be 0.0");
        for (int j = 1; j < K; j++)
            if (!(o->cva[g * K + j] > o->cva[g * K + j - 1])) return fail(GMRM_EINVAL, "mixtures must be given in ascending order");
    }
    gmrm_sampler* s = new gmrm_sampler();
    s->ctx = ctx; s->seed = o->seed; s->rank = o->rank; s->nranks = o->nranks;
    s->shuffle = o->shuffle; s->mimic_hydra = o->mimic_hydra; s->G = G; s->K = K;
    s->cva.assign(o->cva, o->cva + (size_t)G * K);
    s->group_index.assign(o->group_index, o->group_index + ctx->Mt);
    s->mtotgrp.assign(G, 0);
    for (int i = 0; i < ctx->Mt; i++) s->mtotgrp[s->group_index[i]] += 1;         // bayes.cpp:807-809
    s->pi_prior.assign((size_t)G * K, 0.0);
    for (int g = 0; g < G; g++) {                                                  // bayes.hpp:37-47
        double sum_cva = 0.0;
        for (int j = 0; j < K - 1; j++) sum_cva += s->cva[g * K + j + 1];
        s->pi_prior[g * K] = 0.5;
        for (int j = 1; j < K; j++) s->pi_prior[g * K + j] = s->pi_prior[g * K] * s->cva[g * K + j] / sum_cva;
    }
    if (int r = gmrm_set_groups(ctx, s->group_index.data() + ctx->S)) { delete s; return r; }
    s->ch.resize(ctx->T);
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        if (!ctx->tr[t].have_stats)
            if (int r = gmrm_marker_stats(ctx, t)) { delete s; return r; }        // bayes.cpp:788
        c.dist_m.seed((uint32_t)(s->seed + (uint32_t)s->rank));                    // bayes.cpp:796-803
        if (s->mimic_hydra) c.dist_d.seed((uint32_t)(s->seed + (uint32_t)s->rank * 1000u));
        else                c.dist_d.seed((uint32_t)(s->seed + (uint32_t)(s->rank + 1) * 1000u));
        c.midx.resize(ctx->M);
        c.cass.assign((size_t)G * K, 0);
        c.m0.assign(G, 0);
        c.sigmag.assign(G, 0.0);
        c.pi_est.assign((size_t)G * K, 0.0);
        c.beta_sqn.assign(G, 0.0);
        c.betas.assign(ctx->M, 0.0);
    }
    *out = s;
    return GMRM_OK;
}

int gmrm_sampler_destroy(gmrm_sampler* s) { delete s; return GMRM_OK; }

// bayes.cpp:322-335
int gmrm_sampler_init(gmrm_sampler* s) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    for (auto& c : s->ch) {
        for (int i = 0; i < s->ctx->M; i++) c.midx[i] = i;                         // phenotype.cpp:308-312
        c.preshuffled = false;
        for (int g = 0; g < s->G; g++) {
            c.sigmag[g] = gm::rbeta(c.dist_d, 1.0, 1.0);
            if (s->mtotgrp[g] == 0) c.sigmag[g] = 0.0;
        }
        c.pi_est = s->pi_prior;
    }
    return GMRM_OK;
}

// A per-step sweep is open (gmrm_sampler_begin_steps .. _end_steps: the device's effects / components are stale against
// the host copies, the residual is offset by -mu) or a kernel sweep is in flight: the chain state cannot be read or replaced.
static const char* busy_reason(const gmrm_sampler* s) {
    for (int t = 0; t < s->ctx->T; t++) {
        if (s->ch[t].stepping) return "a per-step sweep is open (gmrm_sampler_end_steps or gmrm_sampler_abort_steps first)";
        if (s->ctx->tr[t].in_flight) return "a sweep is in flight (gmrm_sampler_end_sweep first)";
    }
    return nullptr;
}
// ... or a sweep in parts has finished some of its parts but not the last: the components are partly rewritten, the old
// effects are still current and the residual already holds the finished parts' updates -- nothing to save or to close.
static const char* open_parts_reason(const gmrm_sampler* s) {
    for (int t = 0; t < s->ctx->T; t++)
        if (s->ctx->tr[t].part_next != 0) return "a sweep in parts is open (launch and finish its remaining parts first)";
    return nullptr;
}
// A per-step sweep was abandoned after steps had been taken (gmrm_sampler_abort_steps): the caller has applied the
// residual updates of some of those steps -- in a group possibly to some replicas only -- while the device keeps the
// effects of the last completed sweep.  Sweeping on would sample against a residual that no set of effects explains.
static const char* abandoned_reason(const gmrm_sampler* s) {
    for (int t = 0; t < s->ctx->T; t++)
        if (s->ch[t].abandoned)
            return "a per-step sweep was abandoned after steps had been taken: the residual may hold updates of effects that were "
                   "dropped; replace the chain state (gmrm_sampler_load) before going on";
    return nullptr;
}

// bayes.cpp:348-358: add the previous mu back, (it == 1) initial sigmae, draw the new mu
int gmrm_sampler_draw_mu(gmrm_sampler* s, int it, double* mu_drawn) {
    if (!s || !mu_drawn) return fail(GMRM_EINVAL, "null argument");
    gmrm_ctx* ctx = s->ctx;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        if (int r = gmrm_offset_eps(ctx, t, c.mu)) return r;
        if (it == 1)
            if (int r = gmrm_eps_sigma(ctx, t, &c.sigmae)) return r;
        const double nonas = (double)ctx->tr[t].nonas;
        mu_drawn[t] = gm::norm(c.dist_d, c.epssum / nonas, c.sigmae / nonas);      // phenotype.cpp:279-282
    }
    return GMRM_OK;
}

// bayes.cpp:358-367 with the adopted mu: the residual loses mu, the markers are shuffled, the counts start at zero; no launch
int gmrm_sampler_begin_parts(gmrm_sampler* s, const double* mu_use) {
    if (!s || !mu_use) return fail(GMRM_EINVAL, "null argument");
    gmrm_ctx* ctx = s->ctx;
    if (const char* why = busy_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_begin_parts: ") + why);
    if (const char* why = open_parts_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_begin_parts: ") + why);
    if (const char* why = abandoned_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_begin_parts: ") + why);
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        c.mu = mu_use[t];
        if (int r = gmrm_offset_eps(ctx, t, -c.mu)) return r;
        if (s->shuffle && !c.preshuffled) gm::shuffle(s->mimic_hydra ? c.dist_d : c.dist_m, c.midx.data(), ctx->M);   // phenotype.cpp:314-323
        c.preshuffled = false;
        std::fill(c.m0.begin(), c.m0.end(), 0);
        std::fill(c.cass.begin(), c.cass.end(), 0);
        c.n_updates = 0; c.n_batches = 0; c.sweep_ms = 0.0; c.n_planned = 0; c.n_stale = 0; c.n_fastb = 0; c.n_cross = 0; c.n_scrt = 0; c.n_scr = 0;
    }
    return GMRM_OK;
}

// The marker loop over positions [first, first + count) of the visit order is launched (asynchronous).  With several
// shards the residual is remembered first: what this part adds to it is what the shards exchange behind it.
int gmrm_sampler_launch_part(gmrm_sampler* s, int first, int count) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    gmrm_ctx* ctx = s->ctx;
    if (first < 0 || count < 0 || (long long)first + count > ctx->M) return fail(GMRM_EINVAL, "part of the sweep outside [0, M)");
    if (const char* why = busy_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_launch_part: ") + why);
    if (s->nranks > 1)
        for (int t = 0; t < ctx->T; t++)
            if (int r = gmrm_eps_snapshot(ctx, t)) return r;
    if (count == 0 && ctx->M > 0) return GMRM_OK;                // (a shard whose block is shorter than the others': nothing to do in this part)
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        gmrm_sweep_in in{};
        in.G = s->G; in.K = s->K;
        in.order = c.midx.data();
        in.sigmag = c.sigmag.data();
        in.pi_est = c.pi_est.data();
        in.cva = s->cva.data();
        in.sigmae = c.sigmae;
        std::memcpy(in.rng_state, c.dist_d.mt, sizeof(in.rng_state));
        in.rng_index = c.dist_d.idx;
        in.first = first; in.count = count;
        if (int r = gmrm_sweep_launch(ctx, t, &in)) return r;
    }
    return GMRM_OK;
}

// waits for the part in flight: the RNG stream and the counters move on, the component counts are the sweep's so far
static int finish_one(gmrm_sampler* s, int t) {
    gmrm_ctx* ctx = s->ctx;
    Chain& c = s->ch[t];
    gmrm_sweep_out out{};
    out.cass = c.cass.data();
    std::memcpy(out.rng_state, c.dist_d.mt, sizeof(out.rng_state));   // kept if M == 0
    out.rng_index = c.dist_d.idx;
    const bool empty = ctx->M == 0;
    if (int r = gmrm_sweep_finish(ctx, t, &out)) return r;
    if (!empty) {
        std::memcpy(c.dist_d.mt, out.rng_state, sizeof(out.rng_state));
        c.dist_d.idx = out.rng_index;
    }
    c.n_updates += out.n_updates; c.n_batches += out.n_batches; c.sweep_ms += out.device_ms;
    c.n_planned += out.n_planned_stops; c.n_stale += out.n_stale_dots; c.n_fastb += out.n_fast_batches; c.n_cross += out.n_crossed_stops; c.n_scrt += out.n_screen_tries; c.n_scr += out.n_screened_passes;
    return GMRM_OK;
}
int gmrm_sampler_finish_part(gmrm_sampler* s) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    int rc = GMRM_OK;
    for (int t = 0; t < s->ctx->T; t++)
        if (s->ctx->tr[t].in_flight)
            if (int r = finish_one(s, t)) rc = r;
    return rc;
}

// bayes.cpp:358-367 with the adopted mu, then the marker loop is launched (asynchronous); nothing else
int gmrm_sampler_launch_sweep(gmrm_sampler* s, const double* mu_use) {
    if (int r = gmrm_sampler_begin_parts(s, mu_use)) return r;
    return gmrm_sampler_launch_part(s, 0, s->ctx->M);
}

// The marker loops are running on the GPU (the order was copied at launch).  The next iteration's
// shuffle draws from dist_m only, a stream nothing else reads (phenotype.cpp:314-323): do it now, on
// the idle host, instead of in front of the next launch (~5 ms per million markers).  Not with
// --mimic-hydra, where the shuffle shares dist_d with the hyper-parameter draws of this iteration.
int gmrm_sampler_preshuffle(gmrm_sampler* s) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    gmrm_ctx* ctx = s->ctx;
    if (s->shuffle && !s->mimic_hydra)
        for (int t = 0; t < ctx->T; t++) {
            Chain& c = s->ch[t];
            if (c.preshuffled) continue;
            gm::shuffle(c.dist_m, c.midx.data(), ctx->M);
            c.preshuffled = true;
        }
    return GMRM_OK;
}

// bayes.cpp:358-367 with the adopted mu, the launch, and the next iteration's shuffle behind it (one shard per caller;
// a host that drives several shards launches all of them first: gmrm_sampler_launch_sweep, then gmrm_sampler_preshuffle)
int gmrm_sampler_begin_sweep(gmrm_sampler* s, const double* mu_use) {
    if (int r = gmrm_sampler_launch_sweep(s, mu_use)) return r;
    return gmrm_sampler_preshuffle(s);
}

// ---- the reference's per-step schedule (bayes.cpp:374-553 as several MPI tasks run it) -----------------------
// One marker step is one gmrm_dot (device), the Gibbs draw of bayes.cpp:396-492 on the host with the same
// arithmetic as the sweep kernel (gm::exp_, the draws of gm_rng.h), and -- by the caller, for the changed markers
// of ALL shards in shard order -- gmrm_update_eps_from on every residual replica (bayes.cpp:681-706).  A launch
// and a device-to-host copy per marker: the reference's own communication pattern, kept as the schedule whose
// chain is `mpiexec -n R gmrm`'s, not as a fast path (the sweep kernel is that).

// bayes.cpp:358-367 with this shard's mu; no launch
static int begin_steps_body(gmrm_sampler* s, const double* mu_use);

// Leave a per-step sweep that cannot be completed (an error inside _begin_steps / _step, or in another shard of a
// group): the residual gets this shard's mu back (what the next gmrm_sampler_draw_mu would add: c.mu is cleared), the
// device keeps the effects / components of the last completed sweep, the host copies are dropped.  If no step had been
// taken the chain can go on (the RNG streams have advanced: it is no longer the chain it was).  If steps HAD been taken the
// caller has applied some of their residual updates (gmrm_update_eps_from) and the library cannot know which: the chain is
// marked abandoned and every entry that would sweep on refuses until gmrm_sampler_load replaces the state.
// Idempotent.
int gmrm_sampler_abort_steps(gmrm_sampler* s) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    gmrm_ctx* ctx = s->ctx;
    int rc = GMRM_OK;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        if (!c.stepping) continue;
        c.stepping = false;
        if (c.steps_taken > 0) c.abandoned = true;
        c.steps_taken = 0;
        if (int r = gmrm_offset_eps(ctx, t, c.mu)) rc = r;
        c.mu = 0.0;
    }
    return rc;
}

int gmrm_sampler_begin_steps(gmrm_sampler* s, const double* mu_use) {
    if (!s || !mu_use) return fail(GMRM_EINVAL, "null argument");
    if (const char* why = busy_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_begin_steps: ") + why);
    if (const char* why = open_parts_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_begin_steps: ") + why);
    if (const char* why = abandoned_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_begin_steps: ") + why);
    const int rc = begin_steps_body(s, mu_use);
    if (rc != GMRM_OK) {                                          // phenotypes already switched over go back
        const std::string keep = gmrm_last_error();
        gmrm_sampler_abort_steps(s);
        return fail(rc, keep);
    }
    return GMRM_OK;
}

static int begin_steps_body(gmrm_sampler* s, const double* mu_use) {
    gmrm_ctx* ctx = s->ctx;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        c.mu = mu_use[t];
        c.stepping = true;                                        // from here on an abort has to add mu back
        c.steps_taken = 0;
        if (int r = gmrm_offset_eps(ctx, t, -c.mu)) return r;
        if (s->shuffle && !c.preshuffled) gm::shuffle(s->mimic_hydra ? c.dist_d : c.dist_m, c.midx.data(), ctx->M);
        c.preshuffled = false;
        std::fill(c.m0.begin(), c.m0.end(), 0);
        std::fill(c.cass.begin(), c.cass.end(), 0);
        c.comp.resize(ctx->M); c.acum.resize(ctx->M); c.mave.resize(ctx->M); c.msig.resize(ctx->M);
        if (ctx->M > 0) {
            if (int r = gmrm_get_betas(ctx, t, c.betas.data())) return r;
            if (int r = gmrm_get_comp(ctx, t, c.comp.data())) return r;
            if (int r = gmrm_get_acum(ctx, t, c.acum.data())) return r;
            if (int r = gmrm_get_marker_stats(ctx, t, c.mave.data(), c.msig.data())) return r;
        }
        c.n_updates = 0; c.n_batches = 0; c.n_planned = 0; c.n_stale = 0; c.n_fastb = 0; c.n_cross = 0; c.n_scrt = 0; c.n_scr = 0; c.sweep_ms = 0.0;
    }
    return GMRM_OK;
}

// bayes.cpp:376-492 for step mrki of this shard: *mloc = phenotype 0's mrki-th marker (bayes.cpp:384) and, per
// phenotype, dbeta3[3t..3t+2] = {dbeta, mave, msig} (all zero when the effect did not change or mrki >= M:
// "share_mrk is false").  The residual is NOT touched: the caller applies the updates of every shard.
int gmrm_sampler_step(gmrm_sampler* s, int mrki, int* mloc_out, double* dbeta3) {
    if (!s || !mloc_out || !dbeta3 || mrki < 0) return fail(GMRM_EINVAL, "bad argument");
    gmrm_ctx* ctx = s->ctx;
    const int K = s->K;
    for (int i = 0; i < 3 * ctx->T; i++) dbeta3[i] = 0.0;
    *mloc_out = 0;
    if (mrki >= ctx->M) return GMRM_OK;
    const int mloc = s->ch[0].midx[mrki];
    const int mgrp = s->group_index[ctx->S + mloc];
    *mloc_out = mloc;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        if (!c.stepping) return fail(GMRM_ESTATE, "gmrm_sampler_step outside gmrm_sampler_begin_steps .. _end_steps");
        c.steps_taken++;
        if (c.sigmag[mgrp] == 0.0) {                                               // bayes.cpp:396-400
            c.acum[mloc] = 1.0;
            c.betas[mloc] = 0.0;
            continue;
        }
        const double beta = c.betas[mloc];
        const double nm1 = (double)(ctx->tr[t].nonas - 1);
        const double sige_g = c.sigmae / c.sigmag[mgrp];
        const double sigg_e = 1.0 / sige_g;
        const double inv2sige = 1.0 / (2.0 * c.sigmae);
        double denom[gm::KMAX], muk[gm::KMAX], logl[gm::KMAX];
        muk[0] = 0.0; denom[0] = 0.0;
        for (int i = 1; i < K; i++) {
            const double cvai = 1.0 / s->cva[mgrp * K + i];                        // options.cpp:282
            denom[i] = (double)(ctx->N - 1) + sige_g * cvai;
        }
        double num = 0.0;
        if (int r = gmrm_dot(ctx, t, mloc, c.mave[mloc], c.msig[mloc], &num)) return r;
        num += beta * nm1;
        for (int i = 1; i < K; i++) muk[i] = num / denom[i];
        for (int i = 0; i < K; i++) {                                              // bayes.cpp:426-432
            logl[i] = std::log(c.pi_est[mgrp * K + i]);
            if (i > 0) logl[i] = logl[i] + (-0.5 * std::log(sigg_e * nm1 * s->cva[mgrp * K + i] + 1.0) + muk[i] * num * inv2sige);
        }
        const double prob = gm::unif(c.dist_d);
        bool zero_acum = false;                                                    // bayes.cpp:437-445
        double tmp1 = 0.0;
        for (int i = 0; i < K; i++) {
            if (std::fabs(logl[i] - logl[0]) > 700.0) zero_acum = true;
            tmp1 += gm::exp_(logl[i] - logl[0]);
        }
        double acum = zero_acum ? 0.0 : 1.0 / tmp1;
        double dbeta = beta;
        for (int i = 0; i < K; i++) {                                              // bayes.cpp:450-477
            if (prob <= acum || i == K - 1) {
                c.betas[mloc] = i == 0 ? 0.0 : gm::norm(c.dist_d, muk[i], c.sigmae / denom[i]);
                c.cass[mgrp * K + i] += 1;
                c.comp[mloc] = i;
                break;
            }
            bool zero_inc = false;
            for (int j = i + 1; j < K; j++)
                if (std::fabs(logl[j] - logl[i + 1]) > 700.0) zero_inc = true;
            if (!zero_inc) {
                double esum = 0.0;
                for (int k = 0; k < K; k++) esum += gm::exp_(logl[k] - logl[i + 1]);
                acum = acum + 1.0 / esum;
            }
        }
        c.acum[mloc] = acum;
        dbeta -= c.betas[mloc];
        if (std::fabs(dbeta) > 0.0) {                                              // bayes.cpp:483-488
            dbeta3[3 * t + 0] = dbeta;
            dbeta3[3 * t + 1] = c.mave[mloc];
            dbeta3[3 * t + 2] = c.msig[mloc];
            c.n_updates += 1;
        }
    }
    return GMRM_OK;
}

// effects, components and acum go back to the device; local cass and beta_sqn (bayes.cpp:565-568)
int gmrm_sampler_end_steps(gmrm_sampler* s, int* cass, double* beta_sqn) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    gmrm_ctx* ctx = s->ctx;
    const int G = s->G, K = s->K;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        if (!c.stepping) return fail(GMRM_ESTATE, "gmrm_sampler_end_steps without gmrm_sampler_begin_steps");
        c.stepping = false;
        c.steps_taken = 0;
        if (ctx->M > 0) {
            if (int r = gmrm_set_betas(ctx, t, c.betas.data())) return r;
            if (int r = gmrm_set_comp(ctx, t, c.comp.data())) return r;
            if (int r = gmrm_set_acum(ctx, t, c.acum.data())) return r;
        }
        std::fill(c.beta_sqn.begin(), c.beta_sqn.end(), 0.0);
        for (int i = 0; i < ctx->M; i++)
            c.beta_sqn[s->group_index[ctx->S + i]] += c.betas[i] * c.betas[i];
        if (cass) std::memcpy(cass + (size_t)t * G * K, c.cass.data(), sizeof(int) * (size_t)G * K);
        if (beta_sqn) std::memcpy(beta_sqn + (size_t)t * G, c.beta_sqn.data(), sizeof(double) * (size_t)G);
    }
    return GMRM_OK;
}

// waits for the marker loops; local cass and beta_sqn (bayes.cpp:565-568)
int gmrm_sampler_end_sweep(gmrm_sampler* s, int* cass, double* beta_sqn) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    gmrm_ctx* ctx = s->ctx;
    const int G = s->G, K = s->K;
    int rc = GMRM_OK;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        if (ctx->tr[t].in_flight)                       // (a sweep in parts has finished its last part already)
            if (int r = finish_one(s, t)) { rc = r; continue; }
        if (ctx->tr[t].part_next != 0) {                // the last part has not run: the new effects are not current yet
            rc = fail(GMRM_ESTATE, "gmrm_sampler_end_sweep: a sweep in parts is open (launch and finish its remaining parts first)");
            continue;
        }
        static const bool prof = std::getenv("GMRM_HOST_PROF") != nullptr;
        const auto ta = std::chrono::steady_clock::now();
        if (ctx->M > 0)
            if (int r = gmrm_get_betas(ctx, t, c.betas.data())) { rc = r; continue; }
        const auto tb = std::chrono::steady_clock::now();
        std::fill(c.beta_sqn.begin(), c.beta_sqn.end(), 0.0);
        // (bayes.cpp:562-566 adds beta^2 of every marker in marker order; a zero effect adds +0.0, which leaves a sum of
        // non-negative terms as it is -- skipping it is the same sum without a dependent f64 add per marker: 2.4 ms -> 0.3 ms
        // per million markers, of which ~7 k are in the model)
        const double* bp = c.betas.data();
        const int* gp = s->group_index.data() + ctx->S;
        const int M8 = ctx->M & ~7;
        for (int i0 = 0; i0 < M8; i0 += 8) {                 // eight effects at a time: all zero (as bits, sign aside) in most places
            uint64_t w[8];
            std::memcpy(w, bp + i0, sizeof(w));
            if (((w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) << 1) == 0) continue;
            for (int i = i0; i < i0 + 8; i++)
                if (bp[i] != 0.0) c.beta_sqn[gp[i]] += bp[i] * bp[i];
        }
        for (int i = M8; i < ctx->M; i++)
            if (bp[i] != 0.0) c.beta_sqn[gp[i]] += bp[i] * bp[i];
        if (prof)
            std::fprintf(stderr, "[host prof end_sweep] effects to the host %.0f us, beta_sqn loop %.0f us\n",
                         std::chrono::duration<double, std::micro>(tb - ta).count(),
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tb).count());
        if (cass) std::memcpy(cass + (size_t)t * G * K, c.cass.data(), sizeof(int) * (size_t)G * K);
        if (beta_sqn) std::memcpy(beta_sqn + (size_t)t * G, c.beta_sqn.data(), sizeof(double) * (size_t)G);
    }
    return rc;
}

// bayes.cpp:590-651 with the (all-reduced) cass and beta_sqn
int gmrm_sampler_epilogue(gmrm_sampler* s, const int* cass, const double* beta_sqn) {
    if (!s || !cass || !beta_sqn) return fail(GMRM_EINVAL, "null argument");
    gmrm_ctx* ctx = s->ctx;
    const int G = s->G, K = s->K;
    for (int t = 0; t < ctx->T; t++) {
        Chain& c = s->ch[t];
        std::memcpy(c.cass.data(), cass + (size_t)t * G * K, sizeof(int) * (size_t)G * K);
        std::memcpy(c.beta_sqn.data(), beta_sqn + (size_t)t * G, sizeof(double) * (size_t)G);
        for (int g = 0; g < G; g++) {
            if (s->mtotgrp[g] == 0) continue;
            c.m0[g] = s->mtotgrp[g] - c.cass[g * K + 0];
            int cass_sum = 0;
            for (int k = 0; k < K; k++) cass_sum += c.cass[g * K + k];
            if (c.m0[g] == 0 || cass_sum == 0) { c.sigmag[g] = 0.0; continue; }
            const double m0 = (double)c.m0[g];
            c.sigmag[g] = gm::inv_scaled_chisq(c.dist_d, V0G + m0, (c.beta_sqn[g] * m0 + V0G * S02G) / (V0G + m0));
            double sum = 0.0;                                                      // phenotype.cpp:227-237
            for (int i = 0; i < K; i++) {
                const double val = gm::rgamma(c.dist_d, (double)c.cass[g * K + i] + 1.0, 1.0);
                c.pi_est[g * K + i] = val;
                sum += val;
            }
            for (int i = 0; i < K; i++) c.pi_est[g * K + i] = c.pi_est[g * K + i] / sum;
        }
        double e_sqn = 0.0;
        if (int r = gmrm_sumsqr(ctx, t, &e_sqn)) return r;                          // bayes.cpp:631
        c.sigmae = gm::inv_scaled_chisq(c.dist_d, V0E + (double)ctx->N, (e_sqn + V0E * S02E) / (V0E + (double)ctx->N));
    }
    return GMRM_OK;
}

int gmrm_sampler_adopt(gmrm_sampler* s, int t, const double* sigmag, const double* pi_est, double sigmae) {
    if (!s || !sigmag || !pi_est) return fail(GMRM_EINVAL, "null argument");
    if (t < 0 || t >= s->ctx->T) return fail(GMRM_EINVAL, "phenotype index out of range");
    Chain& c = s->ch[t];
    c.sigmag.assign(sigmag, sigmag + s->G);
    c.pi_est.assign(pi_est, pi_est + (size_t)s->G * s->K);
    c.sigmae = sigmae;
    return GMRM_OK;
}

int gmrm_sampler_iterate(gmrm_sampler* s, int it) {
    if (!s) return fail(GMRM_EINVAL, "null sampler");
    if (s->nranks != 1) return fail(GMRM_ESTATE, "gmrm_sampler_iterate is the single-shard form; use the split calls");
    const int T = s->ctx->T;
    std::vector<double> mu(T);
    static const bool prof = std::getenv("GMRM_HOST_PROF") != nullptr;     // where the host's share of an iteration goes (stderr)
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t0 = now();
    if (int r = gmrm_sampler_draw_mu(s, it, mu.data())) return r;
    const auto t1 = now();
    if (int r = gmrm_sampler_launch_sweep(s, mu.data())) return r;
    const auto t2 = now();
    if (int r = gmrm_sampler_preshuffle(s)) return r;
    const auto t3 = now();
    std::vector<int> cass((size_t)T * s->G * s->K);
    std::vector<double> bsq((size_t)T * s->G);
    if (int r = gmrm_sampler_end_sweep(s, cass.data(), bsq.data())) return r;
    const auto t4 = now();
    const int rc = gmrm_sampler_epilogue(s, cass.data(), bsq.data());
    if (prof)
        std::fprintf(stderr, "[host prof it %d] draw_mu %.0f us, launch (offset, order upload, kernel launch) %.0f, next shuffle %.0f (beside the kernel), "
                             "end_sweep %.0f (kernel %.0f ms inside), epilogue %.0f\n", it, us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4),
                     (double)s->ch[0].sweep_ms, us(t4, now()));
    return rc;
}

int gmrm_sampler_get(gmrm_sampler* s, int t, gmrm_hyper* out) {
    if (!s || !out) return fail(GMRM_EINVAL, "null argument");
    if (t < 0 || t >= s->ctx->T) return fail(GMRM_EINVAL, "phenotype index out of range");
    const Chain& c = s->ch[t];
    std::memset(out, 0, sizeof(*out));
    out->sigmae = c.sigmae; out->mu = c.mu;
    int m0s = 0;
    for (int g = 0; g < s->G; g++) m0s += c.m0[g];                                  // phenotype.cpp:98-103
    out->m0_sum = m0s;
    for (int g = 0; g < s->G; g++) out->sigmag[g] = c.sigmag[g];
    for (int i = 0; i < s->G * s->K; i++) out->pi_est[i] = c.pi_est[i];
    out->n_updates = c.n_updates; out->n_batches = c.n_batches; out->sweep_device_ms = c.sweep_ms;
    out->n_planned_stops = c.n_planned; out->n_stale_dots = c.n_stale; out->n_fast_batches = c.n_fastb; out->n_crossed_stops = c.n_cross; out->n_screen_tries = c.n_scrt; out->n_screened_passes = c.n_scr;
    return GMRM_OK;
}

// ---- checkpoint / restart (upstream has none -- its outputs are deleted at start, bayes.cpp:323) --
// Everything the next iteration reads: the residual, effects and components (device), the visit order that the
// next shuffle permutes in place (phenotype.cpp:314-323), hyper-parameters, both RNG streams.  Little-endian
// binary: magic, dimensions, iteration, then per phenotype the fields in the order written below.
int gmrm_sampler_save(gmrm_sampler* s, const char* path, int it) {
    if (!s || !path) return fail(GMRM_EINVAL, "null argument");
    gmrm_ctx* ctx = s->ctx;
    if (const char* why = busy_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_save: ") + why);
    if (const char* why = open_parts_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_save: ") + why);
    if (const char* why = abandoned_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_save: ") + why);
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(GMRM_EIO, std::string("cannot write checkpoint ") + tmp);
    const int hdr[12] = {ctx->N, ctx->M, ctx->Mt, ctx->S, ctx->T, s->G, s->K, it, (int)s->seed, s->rank, s->nranks,
                         s->shuffle * 2 + s->mimic_hydra};
    bool ok = put(f, CKP_MAGIC, 8) && put(f, hdr, 12);
    std::vector<double> eps(4 * ctx->mbytes), betas((size_t)std::max(1, ctx->M));
    std::vector<int> comp((size_t)std::max(1, ctx->M));
    int rc = GMRM_OK;
    for (int t = 0; t < ctx->T && ok && rc == GMRM_OK; t++) {
        const Chain& c = s->ch[t];
        if ((rc = gmrm_download_eps(ctx, t, eps.data())) != GMRM_OK) break;
        if (ctx->M > 0) {
            if ((rc = gmrm_get_betas(ctx, t, betas.data())) != GMRM_OK) break;
            if ((rc = gmrm_get_comp(ctx, t, comp.data())) != GMRM_OK) break;
        }
        const int flags[1] = {c.preshuffled ? 1 : 0};
        const double sc[3] = {c.sigmae, c.mu, c.epssum};
        ok = put(f, sc, 3) && put(f, flags, 1) && put(f, c.dist_m.mt, 624) && put(f, &c.dist_m.idx, 1) &&
             put(f, c.dist_d.mt, 624) && put(f, &c.dist_d.idx, 1) && put(f, c.midx.data(), (size_t)ctx->M) &&
             put(f, c.m0.data(), (size_t)s->G) && put(f, c.cass.data(), (size_t)s->G * s->K) &&
             put(f, c.sigmag.data(), (size_t)s->G) && put(f, c.pi_est.data(), (size_t)s->G * s->K) &&
             put(f, betas.data(), (size_t)ctx->M) && put(f, comp.data(), (size_t)ctx->M) && put(f, eps.data(), eps.size());
    }
    ok = (std::fclose(f) == 0) && ok;
    if (rc != GMRM_OK) { std::remove(tmp.c_str()); return rc; }
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return fail(GMRM_EIO, std::string("short write on checkpoint ") + path); }
    return GMRM_OK;
}

int gmrm_sampler_load(gmrm_sampler* s, const char* path, int* it_out) {
    if (!s || !path || !it_out) return fail(GMRM_EINVAL, "null argument");
    gmrm_ctx* ctx = s->ctx;
    if (const char* why = busy_reason(s)) return fail(GMRM_ESTATE, std::string("gmrm_sampler_load: ") + why);
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(GMRM_EIO, std::string("cannot open checkpoint ") + path);
    char magic[8];
    int hdr[12];
    bool ok = get(f, magic, 8) && get(f, hdr, 12) && std::memcmp(magic, CKP_MAGIC, 8) == 0;
    const int want[12] = {ctx->N, ctx->M, ctx->Mt, ctx->S, ctx->T, s->G, s->K, 0, (int)s->seed, s->rank, s->nranks,
                          s->shuffle * 2 + s->mimic_hydra};
    for (int i = 0; ok && i < 12; i++)
        if (i != 7 && hdr[i] != want[i]) { std::fclose(f); return fail(GMRM_EINVAL, "checkpoint was written for other dimensions / options / seed"); }
    if (!ok) { std::fclose(f); return fail(GMRM_EIO, std::string("not a gmrm checkpoint: ") + path); }
    std::vector<double> eps(4 * ctx->mbytes), betas((size_t)std::max(1, ctx->M));
    std::vector<int> comp((size_t)std::max(1, ctx->M));
    int rc = GMRM_OK;
    for (int t = 0; t < ctx->T && ok && rc == GMRM_OK; t++) {
        Chain& c = s->ch[t];
        int flags[1];
        double sc[3];
        ok = get(f, sc, 3) && get(f, flags, 1) && get(f, c.dist_m.mt, 624) && get(f, &c.dist_m.idx, 1) &&
             get(f, c.dist_d.mt, 624) && get(f, &c.dist_d.idx, 1) && get(f, c.midx.data(), (size_t)ctx->M) &&
             get(f, c.m0.data(), (size_t)s->G) && get(f, c.cass.data(), (size_t)s->G * s->K) &&
             get(f, c.sigmag.data(), (size_t)s->G) && get(f, c.pi_est.data(), (size_t)s->G * s->K) &&
             get(f, betas.data(), (size_t)ctx->M) && get(f, comp.data(), (size_t)ctx->M) && get(f, eps.data(), eps.size());
        if (!ok) break;
        {   // a damaged or foreign file must not reach the kernel: the visit order is a permutation of the block's markers
            // (the kernel indexes the genotype block with it), components are below K, RNG positions within a block
            std::vector<uint8_t> seen((size_t)std::max(1, ctx->M), 0);
            bool good = c.dist_m.idx >= 0 && c.dist_m.idx <= 624 && c.dist_d.idx >= 0 && c.dist_d.idx <= 624 &&
                        std::isfinite(sc[0]) && sc[0] > 0.0 && std::isfinite(sc[1]);
            for (int i = 0; good && i < ctx->M; i++) {
                const int m = c.midx[i];
                if (m < 0 || m >= ctx->M || seen[(size_t)m]) good = false; else seen[(size_t)m] = 1;
                if (comp[i] < 0 || comp[i] >= s->K || !std::isfinite(betas[i])) good = false;
            }
            if (!good) { rc = fail(GMRM_EINVAL, std::string("checkpoint ") + path + " holds an inconsistent chain state (visit order / components / RNG position)"); break; }
        }
        c.sigmae = sc[0]; c.mu = sc[1]; c.epssum = sc[2]; c.preshuffled = flags[0] != 0;
        c.abandoned = false;                                       // the whole chain state is replaced: residual, effects, components,
        ctx->tr[t].part_next = 0; ctx->tr[t].part_last = true;     // visit order, streams -- also what an open sweep in parts left
        if ((rc = gmrm_upload_eps(ctx, t, eps.data())) != GMRM_OK) break;
        if (ctx->M > 0) {
            if ((rc = gmrm_set_betas(ctx, t, betas.data())) != GMRM_OK) break;
            if ((rc = gmrm_set_comp(ctx, t, comp.data())) != GMRM_OK) break;
            c.betas = betas;
        }
    }
    if (ok && rc == GMRM_OK && std::fgetc(f) != EOF) { std::fclose(f); return fail(GMRM_EINVAL, std::string("checkpoint ") + path + " is longer than its header says"); }
    std::fclose(f);
    if (rc != GMRM_OK) return rc;
    if (!ok) return fail(GMRM_EIO, std::string("truncated checkpoint ") + path);
    *it_out = hdr[7];
    return GMRM_OK;
}

// xfiles.cpp:17-42
int gmrm_sampler_csv_line(gmrm_sampler* s, int t, int it, char* buf, size_t len) {
    if (!s || !buf) return fail(GMRM_EINVAL, "null argument");
    if (t < 0 || t >= s->ctx->T) return fail(GMRM_EINVAL, "phenotype index out of range");
    const Chain& c = s->ch[t];
    const int G = s->G, K = s->K;
    size_t n = 0;
    auto put = [&](int w) { if (w > 0) n += (size_t)w; };
    put(std::snprintf(buf + n, n < len ? len - n : 0, "%5d, %4d", it, G));
    double sigmag_sum = 0.0;
    for (int i = 0; i < G; i++) put(std::snprintf(buf + n, n < len ? len - n : 0, ", %20.15f", c.sigmag[i]));
    for (int i = 0; i < G; i++) sigmag_sum += c.sigmag[i];
    int m0s = 0;
    for (int g = 0; g < G; g++) m0s += c.m0[g];
    put(std::snprintf(buf + n, n < len ? len - n : 0, ", %20.15f, %20.15f, %7d, %4d, %2d", c.sigmae,
                      sigmag_sum / (c.sigmae + sigmag_sum), m0s, G, K));
    for (int i = 0; i < G * K; i++) put(std::snprintf(buf + n, n < len ? len - n : 0, ", %20.15f", c.pi_est[i]));
    put(std::snprintf(buf + n, n < len ? len - n : 0, "\n"));
    if (n >= len) return fail(GMRM_EINVAL, "csv buffer too small");
    return (int)n;
}

}  // extern "C"

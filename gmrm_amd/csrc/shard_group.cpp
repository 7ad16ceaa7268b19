// shard_group.cpp -- several marker shards (one context + sampler per GPU) driven by ONE host
// process: the sweep-synchronous schedule of DESIGN.md "Multi-GPU" behind a C ABI, for hosts that
// are not Python (bin/gmrm_hip --devices 0,1,...).  gmrm_amd/dist.py is the same schedule with one
// process per GPU under torch.distributed.
//
// Reference: the MPI calls inside Bayes::process (src/bayes.cpp:495-553 per marker step,
// :575-588 per sweep).  Here, per iteration:
//   1. every shard draws mu on its own stream; shard 0's draw is used            (bayes.cpp:348-351)
//   2. every shard launches its marker loop (persistent kernel, asynchronous)
//   3. per phenotype the pre-rounded residual changes of all shards are summed:
//      ONE all-reduce of 2 x 4 ceil(N/4) doubles (RCCL ncclAllReduce over xGMI; exact sums, so the
//      result does not depend on RCCL's reduction order)
//   4. cass summed, beta_sqn summed in shard order                                (bayes.cpp:575-588)
//   5. every shard runs the hyper-parameter draws, then adopts shard 0's          (bayes.cpp:626,638,649)
// RCCL is opened with dlopen (librccl.so) only when asked for: nothing else in the library needs
// it, and a Python host already carries its own copy.  Without it (or when two shards share a
// device, as in the one-GPU tests) the exchange is staged through host memory.
#include "../../include/gmrm_hip.h"
#include "gm_host.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using gm::fail;

namespace {
// the few RCCL entry points used, by their public C signatures (rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;                 // ncclSuccess == 0
constexpr int kNcclFloat64 = 8;           // ncclDataType_t: ncclDouble / ncclFloat64
constexpr int kNcclSum = 0;               // ncclRedOp_t: ncclSum
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool open() {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(h, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
    }
};
}  // namespace

struct gmrm_group {
    int n = 0, T = 0, G = 0, K = 0;
    size_t nq = 0;                            // doubles per exchange buffer: 2 * 4 * ceil(N/4)
    std::vector<gmrm_ctx*> ctx;
    std::vector<gmrm_sampler*> smp;
    std::vector<double*> q;                   // per shard, on its device
    std::vector<hipStream_t> st;
    std::vector<ncclComm_t> comm;
    Rccl rccl;
    bool use_rccl = false;
    std::vector<double> hq, hsum;             // host staging
    double exchange_ms = 0.0;
    // shards in ascending device order: the order in which their sweeps are launched (each launch takes its device's
    // inter-process lock, capi.cpp: two hosts that take the devices in one global order cannot wait for each other crosswise)
    std::vector<int> by_device;
    const std::vector<int>& launch_order() {
        if ((int)by_device.size() != n) {
            by_device.resize(n);
            for (int r = 0; r < n; r++) by_device[r] = r;
            std::stable_sort(by_device.begin(), by_device.end(), [&](int a, int b) { return ctx[a]->device < ctx[b]->device; });
        }
        return by_device;
    }
};

#define HIPG(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(GMRM_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

static int group_create_body(gmrm_group* g, int n, gmrm_ctx** ctxs, gmrm_sampler** smps, int G, int K, int want_rccl);

extern "C" int gmrm_group_create(gmrm_group** out, int n, gmrm_ctx** ctxs, gmrm_sampler** smps, int G, int K, int want_rccl) {
    if (!out || !ctxs || !smps || n < 1) return fail(GMRM_EINVAL, "bad argument");
    *out = nullptr;
    gmrm_group* g = new gmrm_group();
    const int rc = group_create_body(g, n, ctxs, smps, G, K, want_rccl);
    if (rc != GMRM_OK) {                                      // release the buffers / streams made before the failure
        const std::string keep = gmrm_last_error();
        gmrm_group_destroy(g);
        fail(rc, keep);
        return rc;
    }
    *out = g;
    return GMRM_OK;
}

static int group_create_body(gmrm_group* g, int n, gmrm_ctx** ctxs, gmrm_sampler** smps, int G, int K, int want_rccl) {
    g->n = n; g->G = G; g->K = K;
    g->ctx.assign(ctxs, ctxs + n);
    g->smp.assign(smps, smps + n);
    g->T = ctxs[0]->T;
    g->nq = 2 * 4 * ctxs[0]->mbytes;
    bool distinct = true;
    for (int r = 0; r < n; r++) {
        if (ctxs[r]->N != ctxs[0]->N || ctxs[r]->T != g->T) return fail(GMRM_EINVAL, "shards disagree on N or T");
        for (int s = 0; s < r; s++) if (ctxs[s]->device == ctxs[r]->device) distinct = false;
    }
    g->q.assign(n, nullptr);
    g->st.assign(n, nullptr);
    for (int r = 0; r < n; r++) {
        HIPG(hipSetDevice(ctxs[r]->device));
        HIPG(hipMalloc(reinterpret_cast<void**>(&g->q[r]), g->nq * sizeof(double)));
        HIPG(hipStreamCreateWithFlags(&g->st[r], hipStreamNonBlocking));
    }
    if (n > 1 && want_rccl && distinct && g->rccl.open()) {
        std::vector<int> devs(n);
        for (int r = 0; r < n; r++) devs[r] = ctxs[r]->device;
        g->comm.assign(n, nullptr);
        if (g->rccl.CommInitAll(g->comm.data(), n, devs.data()) == 0) g->use_rccl = true;
        else g->comm.clear();
    }
    if (!g->use_rccl) { g->hq.resize(g->nq); g->hsum.resize(g->nq); }
    return GMRM_OK;
}

extern "C" int gmrm_group_uses_rccl(const gmrm_group* g) { return g && g->use_rccl ? 1 : 0; }

extern "C" int gmrm_group_destroy(gmrm_group* g) {
    if (!g) return GMRM_OK;
    for (int r = 0; r < g->n && r < (int)g->ctx.size(); r++) {
        (void)hipSetDevice(g->ctx[r]->device);
        if (g->use_rccl && r < (int)g->comm.size() && g->comm[r]) g->rccl.CommDestroy(g->comm[r]);
        if (r < (int)g->q.size() && g->q[r]) (void)hipFree(g->q[r]);
        if (r < (int)g->st.size() && g->st[r]) (void)hipStreamDestroy(g->st[r]);
    }
    if (g->rccl.h) dlclose(g->rccl.h);
    delete g;
    return GMRM_OK;
}

// sum the n exchange buffers (same length, one per shard) and leave the total in each
static int allreduce_q(gmrm_group* g) {
    const int n = g->n;
    if (g->use_rccl) {
        if (g->rccl.GroupStart() != 0) return fail(GMRM_EHIP, "ncclGroupStart failed");
        for (int r = 0; r < n; r++) {
            HIPG(hipSetDevice(g->ctx[r]->device));
            const ncclResult_t rc = g->rccl.AllReduce(g->q[r], g->q[r], g->nq, kNcclFloat64, kNcclSum, g->comm[r], g->st[r]);
            if (rc != 0) return fail(GMRM_EHIP, std::string("ncclAllReduce: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(rc) : "error"));
        }
        if (g->rccl.GroupEnd() != 0) return fail(GMRM_EHIP, "ncclGroupEnd failed");
        for (int r = 0; r < n; r++) {
            HIPG(hipSetDevice(g->ctx[r]->device));
            HIPG(hipStreamSynchronize(g->st[r]));
        }
        return GMRM_OK;
    }
    std::fill(g->hsum.begin(), g->hsum.end(), 0.0);
    for (int r = 0; r < n; r++) {                               // exact bins: any order gives the same bits
        HIPG(hipSetDevice(g->ctx[r]->device));
        HIPG(hipMemcpy(g->hq.data(), g->q[r], g->nq * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < g->nq; i++) g->hsum[i] += g->hq[i];
    }
    for (int r = 0; r < n; r++) {
        HIPG(hipSetDevice(g->ctx[r]->device));
        HIPG(hipMemcpy(g->q[r], g->hsum.data(), g->nq * sizeof(double), hipMemcpyHostToDevice));
    }
    return GMRM_OK;
}

extern "C" int gmrm_group_iterate(gmrm_group* g, int it) {
    if (!g) return fail(GMRM_EINVAL, "null group");
    const int n = g->n, T = g->T, G = g->G, K = g->K;
    std::vector<double> mu0(T), mu(T);
    for (int r = 0; r < n; r++)
        if (int rc = gmrm_sampler_draw_mu(g->smp[r], it, r == 0 ? mu0.data() : mu.data())) return rc;
    // Every shard's marker loop is launched before anything else happens on the host (a shard's launch used to wait for
    // the previous shard's next-iteration shuffle: ~0.6 ms per 125 000 markers, several ms of skew at eight shards); the
    // shuffles and the end-of-sweep downloads (effects, counts, RNG state: blocking copies per device) then run on one
    // host thread per shard.
    for (int r : g->launch_order())
        if (int rc = gmrm_sampler_launch_sweep(g->smp[r], mu0.data())) return rc;     // launches; returns at once
    std::vector<int> cass((size_t)T * G * K, 0);
    std::vector<double> bsq((size_t)T * G, 0.0);
    std::vector<std::vector<int>> c1(n, std::vector<int>((size_t)T * G * K));
    std::vector<std::vector<double>> b1(n, std::vector<double>((size_t)T * G));
    std::vector<int> rcs(n, GMRM_OK);
    std::vector<std::string> errs(n);
    auto finish = [&](int r) {
        int rc = gmrm_sampler_preshuffle(g->smp[r]);
        const int rc2 = gmrm_sampler_end_sweep(g->smp[r], c1[r].data(), b1[r].data());   // always: the sweep must be collected
        if (rc == GMRM_OK) rc = rc2;
        rcs[r] = rc;
        if (rc != GMRM_OK) errs[r] = gmrm_last_error();                                 // the message is per thread
    };
    if (n == 1) finish(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < n; r++) th.emplace_back(finish, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n; r++)
        if (rcs[r] != GMRM_OK) return fail(rcs[r], errs[r]);
    for (int r = 0; r < n; r++) {
        for (size_t i = 0; i < cass.size(); i++) cass[i] += c1[r][i];
        for (size_t i = 0; i < bsq.size(); i++) bsq[i] += b1[r][i];                     // shard order, as a sequential MPI_SUM would
    }
    if (n > 1) {
        for (int t = 0; t < T; t++) {
            for (int r = 0; r < n; r++)
                if (int rc = gmrm_eps_delta_export(g->ctx[r], t, g->q[r])) return rc;
            if (int rc = allreduce_q(g)) return rc;
            for (int r = 0; r < n; r++)
                if (int rc = gmrm_eps_delta_import(g->ctx[r], t, g->q[r])) return rc;
        }
    }
    for (int r = 0; r < n; r++)
        if (int rc = gmrm_sampler_epilogue(g->smp[r], cass.data(), bsq.data())) return rc;
    for (int t = 0; t < T; t++) {
        gmrm_hyper h;
        if (int rc = gmrm_sampler_get(g->smp[0], t, &h)) return rc;
        for (int r = 1; r < n; r++)
            if (int rc = gmrm_sampler_adopt(g->smp[r], t, h.sigmag, h.pi_est, h.sigmae)) return rc;
    }
    return GMRM_OK;
}

// gmrm_group_iterate with the residual exchange every k marker positions instead of once per sweep (`--sync-every k`,
// 1 < k < M): part p = positions [p k, (p + 1) k) of every shard's own visit order, all shards'
// parts in flight together, the replicas reconciled behind every part (the same exact all-reduce of the deltas).
extern "C" int gmrm_group_iterate_parts(gmrm_group* g, int it, int k) {
    if (!g) return fail(GMRM_EINVAL, "null group");
    if (k < 1) return fail(GMRM_EINVAL, "gmrm_group_iterate_parts: k must be positive");
    const int n = g->n, T = g->T, G = g->G, K = g->K;
    std::vector<double> mu0(T), mu(T);
    for (int r = 0; r < n; r++)
        if (int rc = gmrm_sampler_draw_mu(g->smp[r], it, r == 0 ? mu0.data() : mu.data())) return rc;
    int Mm = 0;
    for (int r = 0; r < n; r++) {
        if (int rc = gmrm_sampler_begin_parts(g->smp[r], mu0.data())) return rc;
        Mm = std::max(Mm, g->ctx[r]->M);
    }
    int rc = GMRM_OK;
    for (int first = 0; first < Mm && rc == GMRM_OK; first += k) {
        for (int r : g->launch_order()) {                                               // (ascending device order: see launch_order)
            if (rc != GMRM_OK) break;
            const int Mr = g->ctx[r]->M, f = std::min(first, Mr);
            rc = gmrm_sampler_launch_part(g->smp[r], f, std::min(k, Mr - f));           // launches; returns at once
        }
        std::string keep = rc != GMRM_OK ? gmrm_last_error() : "";
        if (first == 0 && rc == GMRM_OK)                                                // the next iteration's shuffles, beside the first parts
            for (int r = 0; r < n && rc == GMRM_OK; r++) {
                rc = gmrm_sampler_preshuffle(g->smp[r]);
                if (rc != GMRM_OK) keep = gmrm_last_error();
            }
        for (int r = 0; r < n; r++) {                                                   // always: what is in flight must be collected
            const int rc2 = gmrm_sampler_finish_part(g->smp[r]);
            if (rc == GMRM_OK && rc2 != GMRM_OK) { rc = rc2; keep = gmrm_last_error(); }
        }
        if (rc != GMRM_OK) return fail(rc, keep);
        if (n > 1)
            for (int t = 0; t < T; t++) {
                for (int r = 0; r < n; r++)
                    if (int rc3 = gmrm_eps_delta_export(g->ctx[r], t, g->q[r])) return rc3;
                if (int rc3 = allreduce_q(g)) return rc3;
                for (int r = 0; r < n; r++)
                    if (int rc3 = gmrm_eps_delta_import(g->ctx[r], t, g->q[r])) return rc3;
            }
    }
    std::vector<int> cass((size_t)T * G * K, 0), c1((size_t)T * G * K);
    std::vector<double> bsq((size_t)T * G, 0.0), b1((size_t)T * G);
    for (int r = 0; r < n; r++) {
        if (int rc3 = gmrm_sampler_end_sweep(g->smp[r], c1.data(), b1.data())) return rc3;
        for (size_t i = 0; i < cass.size(); i++) cass[i] += c1[i];
        for (size_t i = 0; i < bsq.size(); i++) bsq[i] += b1[i];                        // shard order, as a sequential MPI_SUM would
    }
    for (int r = 0; r < n; r++)
        if (int rc3 = gmrm_sampler_epilogue(g->smp[r], cass.data(), bsq.data())) return rc3;
    for (int t = 0; t < T; t++) {
        gmrm_hyper h;
        if (int rc3 = gmrm_sampler_get(g->smp[0], t, &h)) return rc3;
        for (int r = 1; r < n; r++)
            if (int rc3 = gmrm_sampler_adopt(g->smp[r], t, h.sigmag, h.pi_est, h.sigmae)) return rc3;
    }
    return GMRM_OK;
}

// The reference's own schedule (bayes.cpp:340-651 with several MPI tasks), one marker step at a time:
//   1. every shard draws AND uses its own mu                                       (bayes.cpp:348-358)
//   2. step mrki = 0 .. max M - 1: every shard draws the effect of its mrki-th marker against its own residual
//      replica (gmrm_sampler_step); then every replica applies the changed markers of all shards in shard
//      order -- what MPI_Allgather / MPI_Allgatherv + update_epsilon do                  (bayes.cpp:495-553, 681-706)
//   3. cass summed, beta_sqn summed in shard order; hyper-parameter draws; shard 0's adopted   (bayes.cpp:575-651)
// The chain is that of `mpiexec -n <shards> gmrm` (per-shard seeds, bayes.cpp:796-803).  Every step costs a
// kernel launch and a device-to-host copy per shard plus one launch per changed marker and replica: the
// reference's communication pattern, not a fast path -- gmrm_group_iterate is the schedule built for speed.
static int group_iterate_steps_body(gmrm_group* g, int it);
extern "C" int gmrm_group_iterate_steps(gmrm_group* g, int it) {
    if (!g) return fail(GMRM_EINVAL, "null group");
    const int rc = group_iterate_steps_body(g, it);
    if (rc != GMRM_OK) {                      // no shard stays inside a half-done per-step sweep (gmrm_sampler_abort_steps)
        const std::string keep = gmrm_last_error();
        for (int r = 0; r < g->n; r++) gmrm_sampler_abort_steps(g->smp[r]);
        return fail(rc, keep);
    }
    return GMRM_OK;
}
static int group_iterate_steps_body(gmrm_group* g, int it) {
    const int n = g->n, T = g->T, G = g->G, K = g->K;
    std::vector<double> mu(T);
    int Mm = 0;
    for (int r = 0; r < n; r++) {
        if (int rc = gmrm_sampler_draw_mu(g->smp[r], it, mu.data())) return rc;
        if (int rc = gmrm_sampler_begin_steps(g->smp[r], mu.data())) return rc;
        if (g->ctx[r]->M > Mm) Mm = g->ctx[r]->M;
    }
    std::vector<double> d3((size_t)n * T * 3);
    std::vector<int> mloc(n);
    for (int mrki = 0; mrki < Mm; mrki++) {
        for (int r = 0; r < n; r++)
            if (int rc = gmrm_sampler_step(g->smp[r], mrki, &mloc[r], &d3[(size_t)r * T * 3])) return rc;
        for (int dst = 0; dst < n; dst++)
            for (int r = 0; r < n; r++)                                                   // shard order (bayes.cpp:688)
                for (int t = 0; t < T; t++) {
                    const double* d = &d3[((size_t)r * T + t) * 3];
                    if (d[0] != 0.0)                                                      // bayes.cpp:698
                        if (int rc = gmrm_update_eps_from(g->ctx[dst], t, g->ctx[r], mloc[r], d)) return rc;
                }
    }
    std::vector<int> cass((size_t)T * G * K, 0), c1((size_t)T * G * K);
    std::vector<double> bsq((size_t)T * G, 0.0), b1((size_t)T * G);
    for (int r = 0; r < n; r++) {
        if (int rc = gmrm_sampler_end_steps(g->smp[r], c1.data(), b1.data())) return rc;
        for (size_t i = 0; i < cass.size(); i++) cass[i] += c1[i];
        for (size_t i = 0; i < bsq.size(); i++) bsq[i] += b1[i];
    }
    for (int r = 0; r < n; r++)
        if (int rc = gmrm_sampler_epilogue(g->smp[r], cass.data(), bsq.data())) return rc;
    for (int t = 0; t < T; t++) {
        gmrm_hyper h;
        if (int rc = gmrm_sampler_get(g->smp[0], t, &h)) return rc;
        for (int r = 1; r < n; r++)
            if (int rc = gmrm_sampler_adopt(g->smp[r], t, h.sigmag, h.pi_est, h.sigmae)) return rc;
    }
    return GMRM_OK;
}

// One-rank exercise of the dlopen'ed RCCL entry points (signatures, enum values, stream use) on
// `device`: all-reduce of a small f64 buffer in place.  The multi-GPU exchange itself needs several
// devices; this is what a one-GPU box can check of it.
extern "C" int gmrm_rccl_selftest(int device) {
    Rccl r;
    if (!r.open()) return fail(GMRM_ENODEV, "librccl.so could not be opened");
    int rc = GMRM_OK;
    ncclComm_t comm = nullptr;
    double* d = nullptr;
    hipStream_t st = nullptr;
    const int n = 1000;
    std::vector<double> h(n), back(n);
    for (int i = 0; i < n; i++) h[i] = 0.5 * i - 3.0;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = fail(GMRM_EHIP, "hipSetDevice"); break; }
        if (r.CommInitAll(&comm, 1, &device) != 0) { rc = fail(GMRM_EHIP, "ncclCommInitAll failed"); break; }
        if (hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(double)) != hipSuccess) { rc = fail(GMRM_EHIP, "hipMalloc"); break; }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = fail(GMRM_EHIP, "hipStreamCreate"); break; }
        if (hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(GMRM_EHIP, "hipMemcpy"); break; }
        if (r.GroupStart() != 0 || r.AllReduce(d, d, n, kNcclFloat64, kNcclSum, comm, st) != 0 || r.GroupEnd() != 0) { rc = fail(GMRM_EHIP, "ncclAllReduce failed"); break; }
        if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(GMRM_EHIP, "hipStreamSynchronize"); break; }
        if (hipMemcpy(back.data(), d, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GMRM_EHIP, "hipMemcpy"); break; }
        if (std::memcmp(h.data(), back.data(), n * sizeof(double)) != 0) rc = fail(GMRM_EKERNEL, "one-rank ncclAllReduce changed the data");
    } while (0);
    if (st) (void)hipStreamDestroy(st);
    if (d) (void)hipFree(d);
    if (comm) r.CommDestroy(comm);
    dlclose(r.h);
    return rc;
}

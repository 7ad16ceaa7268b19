// capi.cpp -- context management and the C ABI of libgmrm_hip (include/gmrm_hip.h).
// Host-side glue only: device buffers, streams, launches.  No CPU compute fallback: every
// compute entry needs a HIP device and fails with GMRM_ENODEV / GMRM_EHIP otherwise.
#include "../../include/gmrm_hip.h"
#include "gm_common.h"
#include "gm_internal.h"
#include "gm_host.h"

#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace gm {

// Phenotypes sweep side by side as persistent launches on streams of their own, and a persistent kernel holds its
// hardware queue until it ends.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of
// them shared with the null stream): with four chains the fourth queued behind the third (kernel trace: three sweeps
// together, then one).  Ask for eight before the runtime initialises, unless the user has set it.
// (It has no effect when the HIP runtime was initialised before this library was loaded: gmrm_ctx_geometry reports the
// value the environment holds, and the Python loader warns when torch had initialised HIP first.)
__attribute__((constructor)) static void want_hw_queues() { ::setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
static int hip_fail(hipError_t e, const char* what) {
    return fail(GMRM_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

// Tuning / diagnostic knobs from the environment: a value outside the accepted range is reported (once per name and
// process) instead of being dropped silently.
static bool env_int(const char* name, int lo, int hi, int* dst) {
    const char* e = std::getenv(name);
    if (!e) return false;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi) {
        static std::mutex mu;
        static std::map<std::string, bool> said;
        std::lock_guard<std::mutex> lk(mu);
        if (!said[name]) {
            std::fprintf(stderr, "WARNING: libgmrm_hip: %s=%s ignored (expected an integer in [%d, %d])\n", name, e, lo, hi);
            said[name] = true;
        }
        return false;
    }
    *dst = (int)v;
    return true;
}

template <class T> static hipError_t dalloc(T** p, size_t n) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemset(*p, 0, n * sizeof(T));
}

// ---- co-residency bookkeeping ---------------------------------------------------------------
// The sweep kernel's workgroups wait for each other inside the launch, so every workgroup of every
// sweep that is running on a device must be resident at the same time.  Launches are counted per
// DEVICE (two contexts may share one, e.g. `--devices 0,0`): a launch that would not fit beside the
// sweeps already in flight there first waits for them (they then simply run one after another).
struct InFlight { const gmrm_ctx* ctx; int t; hipStream_t stream; int wgs; };
static std::mutex g_dev_mu;
static std::map<int, std::vector<InFlight>> g_dev_inflight;

static int device_inflight_wgs(int device, const gmrm_ctx* c, int own_extra) {
    // a context runs at most `conc` of its own launches at once (the others queue behind them on the
    // same streams); other contexts' launches are counted in full
    int own = own_extra, others = 0;
    for (const InFlight& f : g_dev_inflight[device]) {
        if (f.ctx == c) own++; else others += f.wgs;
    }
    return others + std::min(own, c->conc) * c->W;
}

// The bookkeeping above is per PROCESS.  Two processes that sweep on one device cannot see each other's grids; if the
// two could not be co-resident their workgroups would interleave and both sweeps would end in the spin timeout.  Every
// sweep therefore holds an advisory lock on a per-device file from launch to finish (flock: released by the kernel if
// the process dies): EXCLUSIVE when it needs more than half of the device's resident workgroups, SHARED otherwise -- a
// small sweep waits for another process's large one and the other way round; two small ones of different processes run
// side by side.  (Three or more processes whose small sweeps together exceed the device are NOT caught: the lock knows
// two sizes, not a count.)  One descriptor per device and process, counted: contexts of one process share it (they are
// ordered by the bookkeeping above) and the process holds the stronger of the modes its sweeps asked for.
// The wait is bounded (GMRM_DEVICE_LOCK_TIMEOUT_MS, default 10 minutes) and happens BEFORE g_dev_mu is taken, so that a
// process waiting for one device does not stop its own launches on another; a host that sweeps on several devices takes
// them in ascending device order (shard_group.cpp), so two such hosts cannot wait for each other crosswise.
struct DevLock { int fd = -1; int holders = 0; bool ex = false; bool warned = false; std::mutex mu; std::string path; };
static std::mutex g_lock_map_mu;
static std::map<int, DevLock> g_dev_lock;                     // nodes are stable; each guarded by its own mu

static int devlock_acquire(int device, bool want_ex) {        // g_dev_mu NOT held
    DevLock* L;
    { std::lock_guard<std::mutex> lk(g_lock_map_mu); L = &g_dev_lock[device]; }
    std::lock_guard<std::mutex> lk(L->mu);
    if (L->fd < 0) {
        char bus[64] = "unknown";
        (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device);
        for (char* p = bus; *p; p++) if (*p == ':' || *p == '.' || *p == '/') *p = '_';
        const char* dir = std::getenv("GMRM_LOCK_DIR");
        L->path = std::string(dir ? dir : "/tmp") + "/gmrm_hip_" + bus + ".lock";
        // read-only is enough for flock, so a file another user created can be shared; never follow a link planted there
        L->fd = ::open(L->path.c_str(), O_CREAT | O_RDONLY | O_CLOEXEC | O_NOFOLLOW, 0666);
        if (L->fd >= 0) (void)::fchmod(L->fd, 0666);          // (fails unless we own it: then its owner has done so)
    }
    if (L->fd < 0) {                                          // no lock file, no guard across processes: say so once
        if (!L->warned) {
            std::fprintf(stderr, "WARNING: libgmrm_hip: cannot open the device lock file %s (%s); sweeps of OTHER processes on this device are not "
                                 "kept apart (set GMRM_LOCK_DIR to a directory every user can write).\n", L->path.c_str(), std::strerror(errno));
            L->warned = true;
        }
        L->holders++;
        return GMRM_OK;
    }
    if (L->holders > 0 && (L->ex || !want_ex)) { L->holders++; return GMRM_OK; }     // held in a mode that covers this sweep
    int limit_ms = 600000;
    if (const char* e = std::getenv("GMRM_DEVICE_LOCK_TIMEOUT_MS")) { const int v = std::atoi(e); if (v >= 1) limit_ms = v; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (::flock(L->fd, (want_ex ? LOCK_EX : LOCK_SH) | LOCK_NB) == 0) break;
        if (errno != EWOULDBLOCK && errno != EINTR)
            return fail(GMRM_ESTATE, std::string("flock(") + L->path + "): " + std::strerror(errno));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(limit_ms))
            return fail(GMRM_ESTATE, "device " + std::to_string(device) + " is held by another process's sweep for more than " +
                                     std::to_string(limit_ms) + " ms (lock file " + L->path + ", GMRM_DEVICE_LOCK_TIMEOUT_MS)");
        ::usleep(200);
    }
    L->ex = L->ex || want_ex;
    L->holders++;
    return GMRM_OK;
}
static void devlock_release(int device) {
    DevLock* L;
    { std::lock_guard<std::mutex> lk(g_lock_map_mu); L = &g_dev_lock[device]; }
    std::lock_guard<std::mutex> lk(L->mu);
    if (L->holders > 0 && --L->holders == 0) {
        if (L->fd >= 0) (void)::flock(L->fd, LOCK_UN);
        L->ex = false;
    }
}

int ctx_check_t(const gmrm_ctx* c, int t) {
    if (!c) return fail(GMRM_EINVAL, "null context");
    if (t < 0 || t >= c->T) return fail(GMRM_EINVAL, "phenotype index out of range");
    return GMRM_OK;
}

}  // namespace gm

using namespace gm;

extern "C" {

const char* gmrm_last_error(void) { return g_err.c_str(); }
int gmrm_abi_version(void) { return 1; }
int gmrm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ctx_create_body(gmrm_ctx* c, int device, int N, int M, int Mt, int S, int T) {
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->device = device; c->N = N; c->M = M; c->Mt = Mt; c->S = S; c->T = T;
    c->mbytes = ((size_t)N + 3) / 4;                          // bayes.cpp:776
    // column stride: a multiple of the L2 line (128 B), so that the 256 R-byte slice a sweep workgroup reads of every column
    // is made of whole lines -- with a 16-byte multiple a slice straddled a line that the neighbouring workgroup (another
    // XCD, another L2) fetched as well (GMRM_STRIDE_ALIGN=16 restores that for A/B runs)
    {
        size_t al = 128;
        int v = 0;
        if (env_int("GMRM_STRIDE_ALIGN", 16, 256, &v)) {
            if (v == 16 || v == 64 || v == 128 || v == 256) al = (size_t)v;
            else std::fprintf(stderr, "WARNING: libgmrm_hip: GMRM_STRIDE_ALIGN=%d ignored (16, 64, 128 or 256)\n", v);
        }
        c->stride = (c->mbytes + al - 1) / al * al;
    }
    c->num_cu = prop.multiProcessorCount;
    // Phenotypes are independent chains.  `conc` of them sweep side by side (each persistent launch
    // needs one CU per workgroup): the largest count for which a chain still fits num_cu / conc
    // workgroups (wider slices per workgroup, R up to 4) and that divides T.  Chain t runs on stream t % conc, so the
    // rest queue behind them.  Measured at 500k x 1M: two chains side by side (R = 4, 123 workgroups
    // each) deliver 1.28x the updates/s of one chain at a time (R = 2, 245 workgroups).
    c->R = -1;
    c->conc = 1;
    for (int cc = T; cc >= 1; cc--) {
        if (T % cc != 0) continue;                            // equal groups only (a lone straggler at a wide R costs more than it gains)
        int w = 0;
        const int r = sweep_pick_R(c->stride, c->num_cu / cc, &w);
        if (r > 0) { c->R = r; c->W = w; c->conc = cc; break; }
    }
    c->concurrent = c->conc == T;
    if (c->R < 0) {
        const long long lim = (long long)std::min(c->num_cu, SW_TPB) * SW_TPB * 4 * 4;
        return fail(GMRM_EINVAL, "N too large for the resident-residual sweep kernel on this device: the limit is "
                                 + std::to_string(lim) + " individuals (compute units x 256 threads x 4 bytes x 4 individuals per byte)");
    }
    if (const char* e = std::getenv("GMRM_SWEEP_R")) {        // diagnostic override of the bytes-per-thread choice
        const int r = std::atoi(e);
        if ((r == 1 || r == 2 || r == 4) && (size_t)r * SW_TPB * 256 >= c->stride && r >= c->R) {
            c->R = r; c->W = (int)((c->stride + (size_t)SW_TPB * r - 1) / ((size_t)SW_TPB * r));
        }
    }
    c->Wpad = (c->W + 15) / 16 * 16;
    // Co-residency: every workgroup of the `conc` launches must be resident at once (they wait for
    // each other).  Ask the runtime how many the device holds instead of assuming one per CU.
    {
        int per_cu = 0;
        HIPCHK(sweep_occupancy(c->R, &per_cu));
        c->max_resident_wg = per_cu * c->num_cu;
        if (c->W * c->conc > c->max_resident_wg)
            return fail(GMRM_EINVAL, "sweep geometry cannot be co-resident: " + std::to_string(c->W) + " workgroups x " +
                                     std::to_string(c->conc) + " chains > " + std::to_string(c->max_resident_wg) +
                                     " resident workgroups (occupancy query x compute units)");
    }
    env_int("GMRM_NB_FACTOR16", 8, 256, &c->nb_factor16);
    env_int("GMRM_CROSS_FRAC16", 1, 16, &c->cross_frac16);
    env_int("GMRM_LONG_CROSS", 0, 2, &c->long_cross);
    env_int("GMRM_BATCH_CAP", 16, 240, &c->batch_cap);
    env_int("GMRM_LONG_CROSS_FRAC16", 1, 16, &c->long_cross_frac16);
    if (const char* e = std::getenv("GMRM_CROSS_DENSITY")) {
        const double v = std::atof(e);
        if (v >= 0.0 && v <= 1.0) c->cross_density = v;
        else std::fprintf(stderr, "WARNING: libgmrm_hip: GMRM_CROSS_DENSITY=%s ignored (expected a fraction in [0, 1])\n", e);
    }
    // the sampling screen (sweep.hip, walk_piece) is tried when the recent run length is at least this many sixteenths of a
    // marker; 0: in every pass (tests put the screened branch under the oracle that way), 1000000: never
    env_int("GMRM_SCREEN_MIN_RUN16", 0, 1000000, &c->screen_min_run16);
    env_int("GMRM_SPIN_TIMEOUT_MS", 1, 60000, &c->spin_timeout_ms);

    hipError_t e = hipSuccess;
    const size_t bedbytes = (size_t)(M > 0 ? M : 1) * c->stride;
    e = hipMalloc(reinterpret_cast<void**>(&c->bed), bedbytes);
    if (e != hipSuccess) return fail(GMRM_ENOMEM, std::string("hipMalloc(bed): ") + hipGetErrorString(e));
    HIPCHK(hipMemset(c->bed, 0, bedbytes));
    HIPCHK(dalloc(&c->group, (size_t)(M > 0 ? M : 1)));
    c->tr.resize(T);
    const size_t n4 = 4 * c->stride, Mm = (size_t)(M > 0 ? M : 1);
    for (int t = 0; t < T; t++) {
        Trait& tr = c->tr[t];
        HIPCHK(dalloc(&tr.eps, n4));
        HIPCHK(dalloc(&tr.eps_start, n4));
        HIPCHK(dalloc(&tr.namask2, c->stride));
        HIPCHK(dalloc(&tr.mave, Mm));
        HIPCHK(dalloc(&tr.msig, Mm));
        HIPCHK(dalloc(&tr.nomiss, Mm));
        HIPCHK(dalloc(&tr.betas[0], Mm));
        HIPCHK(dalloc(&tr.betas[1], Mm));
        HIPCHK(dalloc(&tr.comp, Mm));
        HIPCHK(dalloc(&tr.acum, Mm));
        HIPCHK(dalloc(&tr.order, Mm));
        HIPCHK(dalloc(&tr.o_g, Mm)); HIPCHK(dalloc(&tr.o_beta, Mm)); HIPCHK(dalloc(&tr.o_mave, Mm)); HIPCHK(dalloc(&tr.o_msig, Mm)); HIPCHK(dalloc(&tr.o_nm, Mm));
        HIPCHK(dalloc(&tr.tab, (size_t)GMAX * (1 + 3 * KMAX)));
        HIPCHK(dalloc(&tr.rng_state, (size_t)624));
        HIPCHK(dalloc(&tr.rng_index, (size_t)4));
        HIPCHK(dalloc(&tr.cass, (size_t)GMAX * KMAX));
        HIPCHK(dalloc(&tr.stats, (size_t)40));
        HIPCHK(dalloc(&tr.err, (size_t)4));
        HIPCHK(dalloc(&tr.P, (size_t)4 * SW_VMAX * c->Wpad));       // 2 generations x two 8-byte granules per value
        HIPCHK(dalloc(&tr.Tt, (size_t)4 * SW_VMAX));
        HIPCHK(dalloc(&tr.cnt, (size_t)96));
        HIPCHK(dalloc(&tr.scratch, (size_t)8));
        HIPCHK(hipStreamCreateWithFlags(&tr.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&tr.ev0));
        HIPCHK(hipEventCreate(&tr.ev1));
    }
    // The zero-fills above run on the null stream and may still be in flight; every later
    // operation runs on non-blocking streams, which do not wait for the null stream.
    HIPCHK(hipDeviceSynchronize());
    return GMRM_OK;
}

int gmrm_ctx_create(gmrm_ctx** out, int device, int N, int M, int Mt, int S, int T) {
    if (!out) return fail(GMRM_EINVAL, "out is null");
    *out = nullptr;
    if (N < 2 || M < 0 || Mt < M || S < 0 || S + M > Mt || T < 1 || T > 64)
        return fail(GMRM_EINVAL, "bad dimensions");
    if (((long long)N + 3) / 4 > (1ll << gm::MAX_LOG2_N) / 4)
        return fail(GMRM_EINVAL, "N exceeds 2^22 individuals (limit of the exact-summation bins)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GMRM_ENODEV, "no HIP device visible: libgmrm_hip has no CPU path");
    if (device < 0 || device >= ndev) return fail(GMRM_EINVAL, "device index out of range");
    gmrm_ctx* c = new gmrm_ctx();
    c->device = device;
    const int rc = ctx_create_body(c, device, N, M, Mt, S, T);
    if (rc != GMRM_OK) {                                      // release whatever was allocated before the failure
        const std::string keep = g_err;
        gmrm_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return GMRM_OK;
}

int gmrm_ctx_geometry(const gmrm_ctx* c, gmrm_geometry* out) {
    if (!c || !out) return fail(GMRM_EINVAL, "null argument");
    out->R = c->R; out->W = c->W; out->conc = c->conc; out->num_cu = c->num_cu; out->max_resident_wg = c->max_resident_wg;
    out->hw_queues = 0;
    if (const char* e = std::getenv("GPU_MAX_HW_QUEUES")) out->hw_queues = std::atoi(e);
    return GMRM_OK;
}

int gmrm_ctx_destroy(gmrm_ctx* c) {
    if (!c) return GMRM_OK;
    hipSetDevice(c->device);
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto& v = g_dev_inflight[c->device];
        v.erase(std::remove_if(v.begin(), v.end(), [&](const InFlight& f) { return f.ctx == c; }), v.end());
    }
    for (auto& tr : c->tr)
        if (tr.holds_devlock) { devlock_release(c->device); tr.holds_devlock = false; }
    for (auto& tr : c->tr) {
        if (tr.stream) hipStreamSynchronize(tr.stream);
        hipFree(tr.eps); hipFree(tr.eps_start); hipFree(tr.namask2); hipFree(tr.mave); hipFree(tr.msig); hipFree(tr.nomiss);
        hipFree(tr.betas[0]); hipFree(tr.betas[1]); hipFree(tr.comp); hipFree(tr.acum); hipFree(tr.order);
        hipFree(tr.o_g); hipFree(tr.o_beta); hipFree(tr.o_mave); hipFree(tr.o_msig); hipFree(tr.o_nm);
        hipFree(tr.tab); hipFree(tr.rng_state); hipFree(tr.rng_index); hipFree(tr.cass); hipFree(tr.stats);
        hipFree(tr.err); hipFree(tr.P); hipFree(tr.Tt); hipFree(tr.cnt); hipFree(tr.scratch);
        if (tr.ev0) hipEventDestroy(tr.ev0);
        if (tr.ev1) hipEventDestroy(tr.ev1);
        if (tr.stream) hipStreamDestroy(tr.stream);
    }
    hipFree(c->bed); hipFree(c->group); hipFree(c->colbuf);
    delete c;
    return GMRM_OK;
}

int gmrm_ctx_sync(gmrm_ctx* c) {
    if (!c) return fail(GMRM_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->device));
    for (auto& tr : c->tr) HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}

int gmrm_upload_bed(gmrm_ctx* c, const uint8_t* cols, size_t first, size_t n) {
    if (!c || !cols) return fail(GMRM_EINVAL, "null argument");
    if (first + n > (size_t)c->M) return fail(GMRM_EINVAL, "marker range outside this context's block");
    if (n == 0) return GMRM_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy2D(c->bed + first * c->stride, c->stride, cols, c->mbytes, c->mbytes, n, hipMemcpyHostToDevice));
    HIPCHK(launch_recode(c->bed + first * c->stride, n * c->stride, 0, c->tr[0].stream));      // .bed code -> device code (gm_common.h)
    HIPCHK(hipStreamSynchronize(c->tr[0].stream));
    c->have_bed = true;
    for (auto& tr : c->tr) tr.have_stats = false;
    return GMRM_OK;
}

int gmrm_download_bed(gmrm_ctx* c, uint8_t* cols, size_t first, size_t n) {
    if (!c || !cols) return fail(GMRM_EINVAL, "null argument");
    if (first + n > (size_t)c->M) return fail(GMRM_EINVAL, "marker range outside this context's block");
    if (n == 0) return GMRM_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy2D(cols, c->mbytes, c->bed + first * c->stride, c->stride, c->mbytes, n, hipMemcpyDeviceToHost));
    {   // device code -> .bed code, byte by byte through a table
        uint8_t lut[256];
        for (int v = 0; v < 256; v++) lut[v] = (uint8_t)gm::dcode_to_bed((uint32_t)v);
        const size_t tot = n * c->mbytes;
        for (size_t i = 0; i < tot; i++) cols[i] = lut[cols[i]];   // (a bijection on bytes: pad bits come back as they were uploaded)
    }
    return GMRM_OK;
}

int gmrm_synth_bed_ld(gmrm_ctx* c, uint64_t seed, double maf, double miss_rate, int ld_block, double ld_keep) {
    if (!c) return fail(GMRM_EINVAL, "null context");
    if (!(maf > 0.0 && maf < 1.0) || !(miss_rate >= 0.0 && miss_rate < 1.0)) return fail(GMRM_EINVAL, "bad maf / miss_rate");
    if (ld_block < 0 || ld_block > 4096 || !(ld_keep >= 0.0 && ld_keep < 1.0)) return fail(GMRM_EINVAL, "bad ld_block (0..4096) / ld_keep [0, 1)");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_synth(c->bed, c->stride, c->N, c->M, c->S, seed, maf, miss_rate, ld_block, ld_keep, c->tr[0].stream));
    HIPCHK(hipStreamSynchronize(c->tr[0].stream));
    c->have_bed = true;
    for (auto& tr : c->tr) tr.have_stats = false;
    return GMRM_OK;
}
int gmrm_synth_bed(gmrm_ctx* c, uint64_t seed, double maf, double miss_rate) { return gmrm_synth_bed_ld(c, seed, maf, miss_rate, 0, 0.0); }

int gmrm_phen_prepare(const double* y, const uint8_t* isna, int N, double* epsilon, uint8_t* mask4, int* nonas_out) {
    if (!y || !isna || !epsilon || !mask4 || !nonas_out || N < 2) return fail(GMRM_EINVAL, "bad argument");
    const int im4 = (N % 4 == 0) ? N / 4 : N / 4 + 1;           // phenotype.cpp:22
    int nonas = 0;
    double sum = 0.0;
    for (int i = 0; i < im4; i++) mask4[i] = 0x0F;               // phenotype.cpp:601
    for (int i = 0; i < N; i++) {
        if (isna[i]) mask4[i / 4] &= (uint8_t)~(1u << (i % 4));  // phenotype.cpp:614
        else { nonas++; sum += y[i]; }
    }
    if (N % 4 != 0)                                              // phenotype.cpp:633-637
        for (int i = N % 4; i < 4; i++) mask4[N / 4] &= (uint8_t)~(1u << i);
    if (nonas < 2) return fail(GMRM_EINVAL, "fewer than two non-NA phenotype values");
    const double avg = sum / (double)nonas;                      // phenotype.cpp:647-667
    double sqn = 0.0;
    for (int i = 0; i < N; i++) {
        if (isna[i]) epsilon[i] = 0.0;
        else { epsilon[i] = y[i] - avg; sqn += epsilon[i] * epsilon[i]; }
    }
    sqn = std::sqrt((double)(nonas - 1) / sqn);
    for (int i = 0; i < N; i++) epsilon[i] *= sqn;
    for (int i = N; i < 4 * im4; i++) epsilon[i] = 0.0;          // the reference leaves the tail uninitialised
    *nonas_out = nonas;
    return GMRM_OK;
}

int gmrm_upload_trait(gmrm_ctx* c, int t, const double* eps, const uint8_t* mask4, int nonas) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!eps || !mask4) return fail(GMRM_EINVAL, "null argument");
    if (nonas < 2 || nonas > c->N) return fail(GMRM_EINVAL, "nonas out of range");
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    // 2-bit NA mask in the genotype layout: field 11 = phenotype present (na_lut == 1.0,
    // src/na_lut.hpp:3-68), 00 = NA or beyond N
    std::vector<uint8_t> m2(c->stride, 0);
    for (size_t b = 0; b < c->mbytes; b++) {
        uint8_t v = 0;
        for (int k = 0; k < 4; k++)
            if ((mask4[b] >> k) & 1) v |= (uint8_t)(3u << (2 * k));
        m2[b] = v;
    }
    HIPCHK(hipMemcpy(tr.namask2, m2.data(), c->stride, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(tr.eps, 0, 4 * c->stride * sizeof(double)));
    {   // the residual lives on the grid 2^-44 (gm_common.h): rounded here, exact from then on
        std::vector<double> q(4 * c->mbytes);
        for (size_t i = 0; i < q.size(); i++) q[i] = gm::grid(eps[i]);
        HIPCHK(hipMemcpy(tr.eps, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    tr.nonas = nonas;
    tr.have_trait = true;
    tr.have_stats = false;
    tr.poisoned = false;
    return GMRM_OK;
}

int gmrm_download_eps(gmrm_ctx* c, int t, double* eps) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!eps) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->tr[t].stream));
    HIPCHK(hipMemcpy(eps, c->tr[t].eps, 4 * c->mbytes * sizeof(double), hipMemcpyDeviceToHost));
    return GMRM_OK;
}

int gmrm_upload_eps(gmrm_ctx* c, int t, const double* eps) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!eps) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->tr[t].stream));
    std::vector<double> q(4 * c->mbytes);                      // values read back from this library are on the grid already
    for (size_t i = 0; i < q.size(); i++) q[i] = gm::grid(eps[i]);
    HIPCHK(hipMemcpy(c->tr[t].eps, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice));
    return GMRM_OK;
}

static int need_trait(gmrm_ctx* c, int t, bool bed) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!c->tr[t].have_trait) return fail(GMRM_ESTATE, "phenotype not uploaded (gmrm_upload_trait)");
    if (bed && !c->have_bed) return fail(GMRM_ESTATE, "genotypes not uploaded (gmrm_upload_bed / gmrm_synth_bed)");
    return GMRM_OK;
}

int gmrm_marker_stats(gmrm_ctx* c, int t) {
    if (int r = need_trait(c, t, true)) return r;
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(launch_marker_stats(c->bed, tr.namask2, c->stride, c->M, tr.nonas, tr.mave, tr.msig, tr.nomiss, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    {   // one flag per block: can the sweep use the 2-value exchange layout?
        std::vector<uint8_t> nm((size_t)c->M);
        if (c->M > 0) HIPCHK(hipMemcpy(nm.data(), tr.nomiss, (size_t)c->M, hipMemcpyDeviceToHost));
        size_t clean = 0;
        for (uint8_t v : nm) clean += v ? 1 : 0;
        tr.miss_mode = clean == nm.size() ? 0 : (clean == 0 ? 2 : 1);
        tr.n_dirty = (long long)(nm.size() - clean);
    }
    tr.have_stats = true;
    return GMRM_OK;
}

int gmrm_get_marker_stats(gmrm_ctx* c, int t, double* mave, double* msig) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!c->tr[t].have_stats) return fail(GMRM_ESTATE, "marker statistics not computed");
    HIPCHK(hipSetDevice(c->device));
    if (mave) HIPCHK(hipMemcpy(mave, c->tr[t].mave, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost));
    if (msig) HIPCHK(hipMemcpy(msig, c->tr[t].msig, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost));
    return GMRM_OK;
}

int gmrm_set_marker_stats(gmrm_ctx* c, int t, const double* mave, const double* msig) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!mave || !msig) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->tr[t].mave, mave, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->tr[t].msig, msig, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(c->tr[t].nomiss, 0, (size_t)(c->M > 0 ? c->M : 1)));   // unknown: general exchange layout
    HIPCHK(hipDeviceSynchronize());
    c->tr[t].miss_mode = 2;
    c->tr[t].have_stats = true;
    return GMRM_OK;
}

int gmrm_dot(gmrm_ctx* c, int t, int mloc, double mu, double sigma_inv, double* num) {
    if (int r = need_trait(c, t, true)) return r;
    if (mloc < 0 || mloc >= c->M || !num) return fail(GMRM_EINVAL, "marker index out of range");
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(hipMemsetAsync(tr.scratch, 0, 4 * sizeof(double), tr.stream));
    HIPCHK(launch_dot(c->bed + (size_t)mloc * c->stride, tr.namask2, tr.eps, c->stride, tr.scratch, tr.stream));
    double h[4];
    HIPCHK(hipMemcpyAsync(h, tr.scratch, sizeof(h), hipMemcpyDeviceToHost, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    const double dpa = h[0] + h[1];
    const double dpb = h[2] + h[3];
    *num = sigma_inv * (dpa - mu * dpb);                       // bayes.cpp:765
    return GMRM_OK;
}

int gmrm_update_eps(gmrm_ctx* c, int t, int mloc, const double* dbeta) {
    if (int r = need_trait(c, t, true)) return r;
    if (mloc < 0 || mloc >= c->M || !dbeta) return fail(GMRM_EINVAL, "marker index out of range");
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    double alpha_, beta_;                                      // phenotype.cpp:328-329,385-388 on the residual's grid
    gm::update_values(dbeta[0], dbeta[1], dbeta[2], alpha_, beta_);
    const double v0 = beta_, v1 = beta_ + alpha_, v2 = v1 + alpha_, v3 = 0.0;   // per device code: a = 0, 1, 2, missing
    HIPCHK(launch_update(tr.eps, c->bed + (size_t)mloc * c->stride, tr.namask2, c->stride, v0, v1, v2, v3, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}

// bayes.cpp:681-706 as one MPI task sees it: the changed marker of ANOTHER task (its column arrived through
// MPI_Allgatherv, bayes.cpp:537-541) applied to this task's residual replica.  `src` holds the column (marker
// mloc of its block); with src == c this is gmrm_update_eps.  The column is copied device to device into a
// staging buffer of c when the two contexts sit on different devices.
int gmrm_update_eps_from(gmrm_ctx* c, int t, gmrm_ctx* src, int mloc, const double* dbeta) {
    if (int r = need_trait(c, t, true)) return r;
    if (!src || !dbeta) return fail(GMRM_EINVAL, "null argument");
    if (src == c) return gmrm_update_eps(c, t, mloc, dbeta);
    if (!src->have_bed || mloc < 0 || mloc >= src->M) return fail(GMRM_EINVAL, "marker index out of range");
    if (src->N != c->N || src->stride != c->stride) return fail(GMRM_EINVAL, "the two contexts hold different individuals");
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    const uint8_t* col = src->bed + (size_t)mloc * src->stride;
    if (src->device != c->device) {
        if (!c->colbuf) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->colbuf), c->stride));
        HIPCHK(hipMemcpyPeerAsync(c->colbuf, c->device, col, src->device, c->stride, tr.stream));
        col = c->colbuf;
    }
    double alpha_, beta_;                                      // phenotype.cpp:328-329,385-388 on the residual's grid
    gm::update_values(dbeta[0], dbeta[1], dbeta[2], alpha_, beta_);
    const double v0 = beta_, v1 = beta_ + alpha_, v2 = v1 + alpha_, v3 = 0.0;   // per device code: a = 0, 1, 2, missing
    HIPCHK(launch_update(tr.eps, col, tr.namask2, c->stride, v0, v1, v2, v3, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}

int gmrm_offset_eps(gmrm_ctx* c, int t, double offset) {
    if (int r = need_trait(c, t, false)) return r;
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(launch_offset(tr.eps, tr.namask2, c->stride, gm::grid(offset), tr.stream));      // the residual stays on its grid
    HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}

static int sumsq_common(gmrm_ctx* c, int t, bool masked, size_t n, double* out, double* maxabs) {
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(hipMemsetAsync(tr.scratch, 0, 4 * sizeof(double), tr.stream));
    HIPCHK(launch_sumsq(tr.eps, masked ? tr.namask2 : nullptr, n, tr.scratch, tr.scratch + 2, tr.stream));
    double h[3];
    HIPCHK(hipMemcpyAsync(h, tr.scratch, sizeof(h), hipMemcpyDeviceToHost, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    *out = h[0] + h[1];
    if (maxabs) *maxabs = h[2];
    return GMRM_OK;
}

int gmrm_sumsqr(gmrm_ctx* c, int t, double* out) {
    if (int r = need_trait(c, t, false)) return r;
    if (!out) return fail(GMRM_EINVAL, "null argument");
    double mx = 0.0;
    if (int r = sumsq_common(c, t, false, (size_t)c->N, out, &mx)) return r;     // phenotype.cpp:257: i < N
    if (!(mx < gm::EPS_ABS_LIMIT))
        return fail(GMRM_EKERNEL, "|residual| reached 2^8: outside the range of the exact-summation bins");
    return GMRM_OK;
}

int gmrm_eps_sigma(gmrm_ctx* c, int t, double* sigmae) {
    if (int r = need_trait(c, t, false)) return r;
    if (!sigmae) return fail(GMRM_EINVAL, "null argument");
    double s = 0.0;
    if (int r = sumsq_common(c, t, true, 4 * c->mbytes, &s, nullptr)) return r;  // phenotype.cpp:453-457
    *sigmae = s / (double)c->tr[t].nonas * 0.5;
    return GMRM_OK;
}

// ---- Bayes::predict building blocks (src/bayes.cpp:16-284) -------------------------------------
int gmrm_predict_g(gmrm_ctx* c, int t, const double* beta_local, double* g) {
    if (int r = need_trait(c, t, true)) return r;
    if (!beta_local || !g) return fail(GMRM_EINVAL, "null argument");
    Trait& tr = c->tr[t];
    if (!tr.have_stats) return fail(GMRM_ESTATE, "marker statistics not computed (gmrm_marker_stats)");
    HIPCHK(hipSetDevice(c->device));
    double *d_beta = nullptr, *d_g = nullptr;
    void* d_ws = nullptr;
    int rc = GMRM_OK;
    hipError_t e;
    if ((e = dalloc(&d_beta, (size_t)std::max(1, c->M))) != hipSuccess) return hip_fail(e, "hipMalloc");
    if ((e = dalloc(&d_g, 4 * c->stride)) != hipSuccess) { (void)hipFree(d_beta); return hip_fail(e, "hipMalloc"); }
    do {
        if ((e = hipMemset(d_g, 0, 4 * c->stride * sizeof(double))) != hipSuccess) break;
        if ((e = hipMemcpy(d_beta, beta_local, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipDeviceSynchronize()) != hipSuccess) break;
        // The contraction over markers runs on the matrix cores, exact sum rounded once (ops.hip, k_pg_mfma); a block with
        // missing genotypes among the phenotyped individuals (the marker statistics' flags) takes the variant with the
        // indicator planes.  GMRM_PREDICT_LUT=1 forces the in-order f64 kernel.
        if (!std::getenv("GMRM_PREDICT_LUT")) {
            if ((e = hipMalloc(&d_ws, predict_workspace_bytes(c->stride, c->M))) != hipSuccess) break;
            if ((e = launch_predict_g_mfma(c->bed, tr.namask2, c->stride, c->M, tr.mave, tr.msig, d_beta, d_g, d_ws, tr.miss_mode != 0, tr.stream)) != hipSuccess) break;
        } else if ((e = launch_predict_g(c->bed, tr.namask2, c->stride, c->M, tr.mave, tr.msig, d_beta, d_g, tr.stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(tr.stream)) != hipSuccess) break;
        e = hipMemcpy(g, d_g, (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost);
    } while (0);
    if (e != hipSuccess) rc = hip_fail(e, "gmrm_predict_g");
    (void)hipFree(d_beta); (void)hipFree(d_g);
    if (d_ws) (void)hipFree(d_ws);
    return rc;
}

int gmrm_assoc(gmrm_ctx* c, int t, const double* yk, double* xtx, double* xty) {
    if (int r = need_trait(c, t, true)) return r;
    if (!xtx || !xty) return fail(GMRM_EINVAL, "null argument");
    Trait& tr = c->tr[t];
    HIPCHK(hipSetDevice(c->device));
    double *d_y = nullptr, *d_xx = nullptr, *d_xy = nullptr;
    void* d_ws = nullptr;
    int rc = GMRM_OK;
    hipError_t e = hipSuccess;
    do {
        if ((e = hipMalloc(&d_ws, assoc_workspace_bytes(c->stride))) != hipSuccess) break;
        if ((e = dalloc(&d_xx, (size_t)std::max(1, c->M))) != hipSuccess) break;
        if ((e = dalloc(&d_xy, (size_t)std::max(1, c->M))) != hipSuccess) break;
        const double* ysrc = tr.eps;                                   // default: the residual as it stands
        if (yk) {
            if ((e = dalloc(&d_y, 4 * c->stride)) != hipSuccess) break;
            if ((e = hipMemset(d_y, 0, 4 * c->stride * sizeof(double))) != hipSuccess) break;
            if ((e = hipMemcpy(d_y, yk, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) break;
            ysrc = d_y;
        }
        if ((e = hipDeviceSynchronize()) != hipSuccess) break;
        if ((e = launch_assoc(c->bed, tr.namask2, c->stride, c->M, ysrc, d_xx, d_xy, d_ws, tr.stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(tr.stream)) != hipSuccess) break;
        if (c->M > 0) {
            if ((e = hipMemcpy(xtx, d_xx, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) break;
            e = hipMemcpy(xty, d_xy, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost);
        }
    } while (0);
    if (e != hipSuccess) rc = hip_fail(e, "gmrm_assoc");
    if (d_ws) (void)hipFree(d_ws);
    if (d_y) (void)hipFree(d_y);
    if (d_xx) (void)hipFree(d_xx);
    if (d_xy) (void)hipFree(d_xy);
    return rc;
}

int gmrm_set_groups(gmrm_ctx* c, const int* group_local) {
    if (!c || !group_local) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->group, group_local, (size_t)c->M * sizeof(int), hipMemcpyHostToDevice));
    c->have_groups = true;
    return GMRM_OK;
}

int gmrm_sweep_launch(gmrm_ctx* c, int t, const gmrm_sweep_in* in) {
    if (int r = need_trait(c, t, true)) return r;
    if (!in || !in->order || !in->sigmag || !in->pi_est || !in->cva) return fail(GMRM_EINVAL, "null argument");
    Trait& tr = c->tr[t];
    if (!tr.have_stats) return fail(GMRM_ESTATE, "marker statistics not computed (gmrm_marker_stats)");
    if (!c->have_groups) return fail(GMRM_ESTATE, "marker groups not set (gmrm_set_groups)");
    if (tr.in_flight) return fail(GMRM_ESTATE, "a sweep of this phenotype is already in flight");
    if (tr.poisoned)
        return fail(GMRM_ESTATE, "an earlier sweep of this phenotype failed inside the kernel: its per-marker outputs are "
                                 "partly written; upload the phenotype again (gmrm_upload_trait) before sweeping");
    const int G = in->G, K = in->K;
    if (G < 1 || G > GMAX || K < 2 || K > KMAX) return fail(GMRM_EINVAL, "G or K outside the supported range (G<=64, 2<=K<=8)");
    if (in->rng_index < 0 || in->rng_index > 624) return fail(GMRM_EINVAL, "rng_index out of range");
    if (!(in->sigmae > 0.0)) return fail(GMRM_EINVAL, "sigmae must be positive");
    // a part of the sweep: positions [first, first + count) of the order (count == 0: all of it)
    const int first = in->count == 0 ? 0 : in->first, count = in->count == 0 ? c->M : in->count;
    if (in->count != 0 && (in->first < 0 || in->count < 0 || (long long)in->first + in->count > c->M))
        return fail(GMRM_EINVAL, "part of the sweep outside [0, M)");
    if (count >= (1 << 24) - 2)
        return fail(GMRM_EINVAL, "a sweep launch covers at most 16 777 213 markers (the exchange tags count the rounds of a launch modulo 2^24): "
                                 "sweep a larger block in parts (gmrm_sweep_in.first / count, gmrm_sampler_launch_part)");
    if (first != tr.part_next && !(first == 0))
        return fail(GMRM_ESTATE, "parts of a sweep must be launched in order of position (expected first = " + std::to_string(tr.part_next) + ")");
    HIPCHK(hipSetDevice(c->device));
    tr.G = G; tr.K = K;
    if (c->M == 0) { tr.in_flight = true; tr.empty = true; tr.part_last = true; tr.part_next = 0; return GMRM_OK; }
    tr.empty = false;
    const bool part_last = first + count == c->M;      // (the bookkeeping of a sweep in parts moves once the launch is enqueued: a
    const int part_next = part_last ? 0 : first + count;   //  part whose uploads or launch fail can be launched again)

    // per-group tables of the Gibbs step, evaluated exactly as bayes.cpp:403-432 writes them
    std::vector<double> tab((size_t)G * (1 + 3 * K), 0.0);
    double* sg = tab.data();
    double* denom = sg + G;
    double* logpi = denom + (size_t)G * K;
    double* mhl = logpi + (size_t)G * K;
    const double nm1 = (double)(tr.nonas - 1);
    for (int g = 0; g < G; g++) {
        sg[g] = in->sigmag[g];
        if (sg[g] == 0.0) continue;                            // bayes.cpp:396: group is skipped
        const double sige_g = in->sigmae / sg[g];
        const double sigg_e = 1.0 / sige_g;
        for (int k = 0; k < K; k++) {
            logpi[g * K + k] = std::log(in->pi_est[g * K + k]);
            if (k > 0) {
                const double cvai = 1.0 / in->cva[g * K + k];  // options.cpp:282
                denom[g * K + k] = (double)(c->N - 1) + sige_g * cvai;
                mhl[g * K + k] = -0.5 * std::log(sigg_e * nm1 * in->cva[g * K + k] + 1.0);
            }
        }
    }
    // blocking copies: the sources (a local table, the caller's sweep_in) need not outlive this call
    HIPCHK(hipMemcpy(tr.tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    if (first == 0) HIPCHK(hipMemcpy(tr.order, in->order, (size_t)c->M * sizeof(int), hipMemcpyHostToDevice));   // (later parts: the same order)
    HIPCHK(hipMemcpy(tr.rng_state, in->rng_state, 624 * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(tr.rng_index, &in->rng_index, sizeof(int), hipMemcpyHostToDevice));
    // every polled word starts at zero in every launch (tags count from 1 inside the launch)
    HIPCHK(hipMemsetAsync(tr.cnt, 0, 96 * sizeof(unsigned), tr.stream));
    HIPCHK(hipMemsetAsync(tr.P, 0, (size_t)4 * SW_VMAX * c->Wpad * sizeof(double), tr.stream));
    HIPCHK(hipMemsetAsync(tr.Tt, 0, (size_t)4 * SW_VMAX * sizeof(double), tr.stream));
    HIPCHK(hipMemsetAsync(tr.err, 0, 4 * sizeof(int), tr.stream));
    if (first == 0) HIPCHK(hipMemsetAsync(tr.cass, 0, (size_t)GMAX * KMAX * sizeof(int), tr.stream));   // every workgroup adds its share; the parts of a sweep add up
    HIPCHK(hipMemsetAsync(tr.stats, 0, 40 * sizeof(long long), tr.stream));

    SweepArgs a{};
    a.N = c->N; a.M = count; a.W = c->W; a.Wpad = c->Wpad; a.G = G; a.K = K;
    a.stride = c->stride;
    a.bed = c->bed; a.namask2 = tr.namask2; a.order = tr.order + first; a.group = c->group;
    a.mave = tr.mave; a.msig = tr.msig; a.nomiss = tr.nomiss;
    a.betas_in = tr.betas[tr.cur]; a.betas_out = tr.betas[tr.cur ^ 1];
    a.o_g = tr.o_g + first; a.o_beta = tr.o_beta + first; a.o_mave = tr.o_mave + first; a.o_msig = tr.o_msig + first; a.o_nm = tr.o_nm + first;
    a.comp = tr.comp; a.acum = tr.acum; a.eps = tr.eps;
    a.sigmag = tr.tab; a.denom = tr.tab + G; a.logpi = tr.tab + G + (size_t)G * K; a.mhl = tr.tab + G + 2 * (size_t)G * K;
    a.sigmae = in->sigmae;
    a.inv2sige = 1.0 / (2.0 * in->sigmae);                     // bayes.cpp:406
    a.nm1 = nm1;
    a.rng_state = tr.rng_state; a.rng_index = tr.rng_index;
    a.cass = tr.cass; a.stats = tr.stats; a.err = tr.err;
    a.P = tr.P; a.Tt = tr.Tt; a.cnt = tr.cnt;
    a.batch_init = c->batch_init;
    a.bcap = c->batch_cap;
    a.direct_pub = std::getenv("GMRM_NO_DIRECT_PUBLISH") ? 0 : 1;   // A/B knob
    // wavefront 0 waits before its first look at the totals (sweep.hip): measured optima on 500k x 1M (245 workgroups): 45 units of
    // 64 clocks for the long-batch kernel, 15 for the others; two chains of 123 workgroups: 25; 50k x 100k (49 workgroups): 20 and 0
    // (profiles/r04_ab_totals_delay.txt; GMRM_TOTALS_DELAY / GMRM_TOTALS_DELAY2 override)
    a.totals_delay = c->W >= 200 ? 45 : (c->W >= 100 ? 25 : 20); a.totals_delay2 = c->W >= 100 ? 15 : 0;
    if (const char* e = std::getenv("GMRM_TOTALS_DELAY")) a.totals_delay = std::atoi(e) & 127;
    if (const char* e = std::getenv("GMRM_TOTALS_DELAY2")) a.totals_delay2 = std::atoi(e) & 127;
    a.tile_trim = std::getenv("GMRM_NO_TILE_TRIM") ? 0 : 1;         // A/B knob (sweep.hip, the batch at the top of a round)
    a.trace = nullptr;
    if (std::getenv("GMRM_SWEEP_TRACE")) {                  // diagnostic build only
        if (!tr.trace) HIPCHK(hipMalloc(reinterpret_cast<void**>(&tr.trace), (size_t)256 * 64 * 8 * 8));
        HIPCHK(hipMemset(tr.trace, 0, (size_t)256 * 64 * 8 * 8));
        HIPCHK(hipDeviceSynchronize());
        a.trace = tr.trace;
    }
    a.nb_factor16 = c->nb_factor16;
    a.screen_min_run16 = c->screen_min_run16;
    a.miss_mode = tr.miss_mode;
    if (std::getenv("GMRM_FORCE_MIXED")) a.miss_mode = 1;     // diagnostic: run any block through the per-marker-layout kernel
    a.reduce4 = std::getenv("GMRM_REDUCE_W0") ? 0 : 1;      // (only read by builds with -DGM_PACK_ROWS=0: the short-batch kernels' reduce role on wavefront 0 alone; measured c2 22.7 / 23.6 ms, c6 148 / 154, c5 equal)
    // The walk may cross markers whose effect was non-zero: when no marker of the block has a missing genotype among the
    // phenotyped individuals (mode 0: mave * nonas is then the integer sum of a marker's genotype values; flags and means
    // come from gmrm_marker_stats -- values set through gmrm_set_marker_stats leave miss_mode at 2), or when every marker
    // is treated as having some (mode 2: everything the patch needs is counted inside the kernel).  The kernel with that code
    // is ~3 % slower in rounds that cross nothing, and a crossing costs about half a round, so it is launched only when
    // enough markers are in the model (their number is known: the previous sweep's component counts) -- measured on
    // 500k x 1M: 7.8 % in the model 690 -> 470 ms per sweep, 1.7 % 202 -> 181, 0.36 % (stationary) 132 -> 136.  The chain
    // is the same either way, bit for bit.  GMRM_NO_CROSS=1 / GMRM_FORCE_CROSS=1: A/B knobs.
    a.cross = 0;
    a.long_cross = 0;
    // Some markers with missing genotypes (mode 1): when they are few (at most one in 30: a batch of 240 then seldom holds more than the
    // eight whose Z terms the kernel gathers) and the model is sparse, the sweep runs on the long-batch kernel too (sweep.hip: every
    // tile on the one-MFMA-set pass, the dirty markers' Z terms gathered from the digit planes).  GMRM_NO_LONG_MIXED=1: A/B knob.
    a.long_mixed = 0;
    if (a.miss_mode == 1 && !std::getenv("GMRM_NO_LONG_MIXED")) {
        const bool dense = (double)tr.in_model >= c->cross_density * (double)c->M;
        if (!dense && tr.n_dirty * 30 <= (long long)c->M) a.long_mixed = 1;
    }
    if ((a.miss_mode == 0 || a.miss_mode == 2) && !std::getenv("GMRM_NO_CROSS")) {      // (the mixed layout, mode 1, has no such kernel)
        const bool dense = (double)tr.in_model >= c->cross_density * (double)c->M;      // (as of the last completed sweep)
        if (dense || std::getenv("GMRM_FORCE_CROSS")) a.cross = c->cross_frac16;
        // Sparse models in the layout without missing genotypes (the stationary sweeps): the long-batch kernel that crosses stops --
        // batches of up to 240 markers, the walk on four wavefronts, the exact sums exchanged (sweep.hip, long_cont_body).  Exact
        // and tested, but NOT the default: on 500k x 1M it takes 12 % fewer rounds and 10 % more time (a crossing costs about as
        // much as the round it saves while the tile window holds 20 tiles: profiles/r04_ab_long_batch_crossing.txt).
        // GMRM_LONG_CROSS=1: in sparse models; 2: in dense models as well.
        if (a.miss_mode == 0 && c->long_cross > 0 && (!dense || c->long_cross > 1)) { a.cross = c->long_cross_frac16; a.long_cross = 1; }
    }
    a.spin_ticks = (unsigned long long)c->spin_timeout_ms * 100000ull;      // s_memrealtime ticks (100 MHz)
    // Phenotypes that do not fit side by side share stream 0 and run one after another.
    hipStream_t st = c->tr[t % c->conc].stream;       // conc chains side by side, the others queue behind them
    if (t % c->conc != t) HIPCHK(hipStreamSynchronize(tr.stream));           // uploads above done
    // Co-residency across contexts that share this device: wait for sweeps that would not fit beside this one.  The
    // mutex is held from the check to the enqueue, so that an entry becomes visible together with its kernel (two host
    // threads launching on one device cannot both see "room"), and nothing is registered when the launch fails.
    int grid = c->W;
    if (const char* e = std::getenv("GMRM_FAULT_DROP_WG")) {                 // test hook: launch one workgroup short, so the
        if (std::atoi(e) > 0 && grid > 1) grid -= 1;                         // grid-wide wait can never complete (timeout path)
    }
    // another PROCESS on this device: see devlock_acquire (may wait, bounded; before the mutex)
    const bool guarded = !std::getenv("GMRM_NO_DEVICE_LOCK");
    if (guarded)
        if (int r = devlock_acquire(c->device, 2 * c->W * c->conc > c->max_resident_wg)) return r;
    {
        std::unique_lock<std::mutex> lk(g_dev_mu);
        auto& v = g_dev_inflight[c->device];
        while (device_inflight_wgs(c->device, c, 1) > c->max_resident_wg) {
            auto it = std::find_if(v.begin(), v.end(), [&](const InFlight& f) { return f.ctx != c; });
            if (it == v.end()) break;                   // only this context's own launches: they queue on its streams
            const hipStream_t other = it->stream;
            v.erase(it);                                // its owner's gmrm_sweep_finish still synchronises the stream
            const hipError_t se = hipStreamSynchronize(other);   // (the kernel it waits for needs nothing from this host thread)
            if (se != hipSuccess) { lk.unlock(); if (guarded) devlock_release(c->device); return hip_fail(se, "hipStreamSynchronize(other sweep)"); }
        }
        // the sampling step's inputs of this part in visit order (ops.hip, k_order_inputs: ~30 us per million markers), then the sweep
        hipError_t le = launch_order_inputs(tr.order + first, count, c->group, tr.betas[tr.cur], tr.mave, tr.msig, tr.nomiss,
                                            tr.o_g + first, tr.o_beta + first, tr.o_mave + first, tr.o_msig + first, tr.o_nm + first, st);
        if (le == hipSuccess) le = hipEventRecord(tr.ev0, st);
        if (le == hipSuccess) le = launch_sweep(a, c->R, st, grid);
        if (le == hipSuccess) le = hipEventRecord(tr.ev1, st);
        if (le != hipSuccess) {
            lk.unlock();
            if (guarded) devlock_release(c->device);
            return hip_fail(le, "sweep launch");
        }
        tr.holds_devlock = guarded;
        v.push_back(InFlight{c, t, st, c->W});
    }
    tr.part_last = part_last;
    tr.part_next = part_next;
    tr.launch_stream = st;
    tr.in_flight = true;
    return GMRM_OK;
}

int gmrm_sweep_finish(gmrm_ctx* c, int t, gmrm_sweep_out* out) {
    if (int r = ctx_check_t(c, t)) return r;
    Trait& tr = c->tr[t];
    if (!tr.in_flight) return fail(GMRM_ESTATE, "no sweep in flight for this phenotype");
    tr.in_flight = false;
    if (tr.empty) {
        if (out) { out->n_updates = 0; out->n_batches = 0; out->device_ms = 0.0; out->n_planned_stops = 0; out->n_stale_dots = 0; out->n_fast_batches = 0; out->n_crossed_stops = 0; out->n_screen_tries = 0; out->n_screened_passes = 0;
                   if (out->cass) std::memset(out->cass, 0, sizeof(int) * (size_t)tr.G * tr.K); }
        return GMRM_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    const hipError_t sync_err = hipStreamSynchronize(tr.launch_stream);
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto& v = g_dev_inflight[c->device];
        auto it = std::find_if(v.begin(), v.end(), [&](const InFlight& f) { return f.ctx == c && f.t == t; });
        if (it != v.end()) v.erase(it);
    }
    if (tr.holds_devlock) { devlock_release(c->device); tr.holds_devlock = false; }
    if (sync_err != hipSuccess) { tr.poisoned = true; return hip_fail(sync_err, "hipStreamSynchronize(sweep)"); }
    int err[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(err, tr.err, sizeof(err), hipMemcpyDeviceToHost));
    if (err[0] != 0) {
        // the kernel left before storing the residual, the RNG state and the new effects' buffer was not
        // adopted (tr.cur unchanged) -- but comp / acum are partly overwritten: the chain state is unusable
        tr.poisoned = true;
        if (err[0] == 1) return fail(GMRM_EKERNEL, "sweep kernel: a grid-wide wait timed out (workgroups not co-resident?)");
        if (err[0] == 2) return fail(GMRM_EKERNEL, "sweep kernel: RNG window exhausted inside one batch");
        if (err[0] == 3) return fail(GMRM_EKERNEL, "sweep kernel: dynamic LDS does not start at offset 0");
        if (err[0] == 4) return fail(GMRM_EKERNEL, "sweep kernel: |residual| reached 2^8 inside the sweep: outside the range of the exact-summation bins");
        return fail(GMRM_EKERNEL, "sweep kernel: unknown error code " + std::to_string(err[0]));
    }
    if (tr.part_last) tr.cur ^= 1;                  // (a part that does not end the sweep: the old effects stay current)
    // component counts of this sweep (so far, if it runs in parts); markers in the model afterwards = those not in
    // component 0 (the next launch's choice of kernel)
    std::vector<int> hc((size_t)tr.G * tr.K);
    HIPCHK(hipMemcpy(hc.data(), tr.cass, sizeof(int) * hc.size(), hipMemcpyDeviceToHost));
    if (tr.part_last) {
        tr.in_model = 0;
        for (int g = 0; g < tr.G; g++)
            for (int k = 1; k < tr.K; k++) tr.in_model += hc[(size_t)g * tr.K + k];
    }
    if (out) {
        long long st[40];
        HIPCHK(hipMemcpy(st, tr.stats, sizeof(st), hipMemcpyDeviceToHost));
        out->n_updates = st[0]; out->n_batches = st[1]; out->n_planned_stops = st[29]; out->n_stale_dots = st[30];
        out->n_fast_batches = st[31]; out->n_crossed_stops = st[32]; out->n_screen_tries = st[33]; out->n_screened_passes = st[34];
        if (const char* path = std::getenv("GMRM_SWEEP_TRACE")) {
            if (tr.trace) {
                std::vector<unsigned long long> h((size_t)256 * 64 * 8);
                HIPCHK(hipMemcpy(h.data(), tr.trace, h.size() * 8, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(path, "ab")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
            }
        }
        if (std::getenv("GMRM_SWEEP_PROF")) {                  // diagnostic build only (-DGM_SWEEP_PROF)
            static const char* nm[8] = {"promote+reduce", "dots", "pf_issue", "pf_commit", "wait_totals", "post_barrier", "update", "sample"};
            for (int w = 0; w < 2; w++) {
                std::fprintf(stderr, "[sweep prof wg %s] batches %lld (discarded %lld):", w ? "W/2" : "0", st[1], st[3]);
                for (int i = 0; i < 8; i++)
                    std::fprintf(stderr, " %s %.2fus", nm[i], st[1] ? st[(w ? 12 : 4) + i] * 0.01 / (double)st[1] : 0.0);
                std::fprintf(stderr, "\n");
            }
            std::fprintf(stderr, "[sweep prof sampler wg W/2] inputs %.2fus decide0 %.2fus search %.2fus commit %.2fus\n",
                         st[20] * 0.01 / (double)st[1], st[21] * 0.01 / (double)st[1], st[22] * 0.01 / (double)st[1],
                         st[23] * 0.01 / (double)st[1]);
            std::fprintf(stderr, "[sweep prof phase A wg W/2] loop top -> barrier %.2fus scan+inputs %.2fus tiles %.2fus barrier %.2fus publish %.2fus (max batch %lld)\n",
                         st[24] * 0.01 / (double)st[1], st[25] * 0.01 / (double)st[1], st[26] * 0.01 / (double)st[1],
                         st[27] * 0.01 / (double)st[1], st[28] * 0.01 / (double)st[1], st[2]);
            std::fprintf(stderr, "[sweep prof loop top wg W/2] meta commit %.2fus batch bookkeeping %.2fus wait for tile loads %.2fus (the rest of 'loop top -> barrier' is the barrier)\n",
                         st[35] * 0.01 / (double)st[1], st[36] * 0.01 / (double)st[1], st[37] * 0.01 / (double)st[1]);
            std::fprintf(stderr, "[sweep prof polls wg W/2] looks per round: %.2f at the row of partial sums (reduce role, wavefront 0), %.2f at the totals\n",
                         st[38] / (double)st[1], st[39] / (double)st[1]);
        }
        if (out->cass) std::memcpy(out->cass, hc.data(), sizeof(int) * (size_t)tr.G * tr.K);
        HIPCHK(hipMemcpy(out->rng_state, tr.rng_state, 624 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&out->rng_index, tr.rng_index, sizeof(int), hipMemcpyDeviceToHost));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, tr.ev0, tr.ev1));
        out->device_ms = ms;
    }
    return GMRM_OK;
}

int gmrm_get_betas(gmrm_ctx* c, int t, double* betas) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!betas) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(betas, c->tr[t].betas[c->tr[t].cur], (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost));
    return GMRM_OK;
}
int gmrm_set_betas(gmrm_ctx* c, int t, const double* betas) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!betas) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->tr[t].betas[c->tr[t].cur], betas, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice));
    long long nz = 0;
    for (int i = 0; i < c->M; i++) nz += betas[i] != 0.0 ? 1 : 0;
    c->tr[t].in_model = nz;
    return GMRM_OK;
}
int gmrm_get_comp(gmrm_ctx* c, int t, int* comp) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!comp) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(comp, c->tr[t].comp, (size_t)c->M * sizeof(int), hipMemcpyDeviceToHost));
    return GMRM_OK;
}
int gmrm_set_comp(gmrm_ctx* c, int t, const int* comp) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!comp) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->tr[t].comp, comp, (size_t)c->M * sizeof(int), hipMemcpyHostToDevice));
    return GMRM_OK;
}
int gmrm_set_acum(gmrm_ctx* c, int t, const double* acum) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!acum) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->tr[t].acum, acum, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice));
    return GMRM_OK;
}
int gmrm_get_acum(gmrm_ctx* c, int t, double* acum) {
    if (int r = ctx_check_t(c, t)) return r;
    if (!acum) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(acum, c->tr[t].acum, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost));
    return GMRM_OK;
}

int gmrm_selftest_math(int device, int op, const double* x, double* y, int n) {
    if (!x || !y || n < 1 || op < 0 || op > 4 || (op == 3 && n > 65536)) return fail(GMRM_EINVAL, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GMRM_ENODEV, "no HIP device visible");
    HIPCHK(hipSetDevice(device));
    const size_t nout = (size_t)n * (op == 4 ? 2 : 1);
    double *dx = nullptr, *dy = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dx), (size_t)n * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dy), nout * sizeof(double)));
    HIPCHK(hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(launch_selftest(op, dx, dy, n, nullptr));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(y, dy, nout * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dy);
    return GMRM_OK;
}

int gmrm_eps_snapshot(gmrm_ctx* c, int t) {
    if (int r = need_trait(c, t, false)) return r;
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(hipMemcpyAsync(tr.eps_start, tr.eps, 4 * c->stride * sizeof(double), hipMemcpyDeviceToDevice, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}
int gmrm_eps_delta_export(gmrm_ctx* c, int t, double* dev_q) {
    if (int r = need_trait(c, t, false)) return r;
    if (!dev_q) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(launch_delta_export(tr.eps, tr.eps_start, dev_q, 4 * c->mbytes, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}
int gmrm_eps_delta_import(gmrm_ctx* c, int t, const double* dev_q) {
    if (int r = need_trait(c, t, false)) return r;
    if (!dev_q) return fail(GMRM_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    Trait& tr = c->tr[t];
    HIPCHK(launch_delta_import(tr.eps, tr.eps_start, dev_q, 4 * c->mbytes, tr.stream));
    HIPCHK(hipStreamSynchronize(tr.stream));
    return GMRM_OK;
}

}  // extern "C"

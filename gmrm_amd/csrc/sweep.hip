// sweep.hip -- the marker loop of Bayes::process (reference src/bayes.cpp:375-553) for one
// phenotype on one GPU as ONE persistent kernel launch:
//   Bayes::dot_product          src/bayes.cpp:709-770      -> phase A (all workgroups)
//   Gibbs step                  src/bayes.cpp:396-492      -> sample_batch (wavefront 0)
//   Phenotype::update_epsilon   src/phenotype.cpp:326-393  -> phase C (all workgroups)
//
// Layout.  The residual never leaves the chip during a sweep: workgroup w / thread t owns
// R consecutive bytes of every genotype column (4R individuals) and keeps their residual
// eps_i and its two pre-rounded parts (q1_i, q2_i) in VGPRs.  A marker's dot product is then
// 4 partial sums per thread (sum a*q1, a*q2, b*q1, b*q2), all exact (gm_common.h), reduced by
// wavefront shuffles -> LDS -> one value per workgroup -> cross-workgroup.
//
// Schedule.  The chain is sequential (marker j+1 needs the residual after marker j), and a
// grid-wide exchange costs microseconds on an 8-XCD part, so markers are processed in
// speculative batches: the dots of the next nb markers of the shuffled order are computed
// against the current residual in one pass; wavefront 0 then walks the batch in order and
// stops at the first marker whose effect changes (dbeta != 0), because every later dot in
// the batch is then stale.  The residual update is applied and the next batch starts after
// that marker.  Results are exactly those of the one-marker-at-a-time loop.
//
// Exchange per batch (placement-independent, gfx950: private L2 per XCD):
//   1. every workgroup stores its nb*4 partials to P[v][wg] with sc1 (write-through) stores,
//      drains them, then one lane bumps cnt1 (agent-scope atomic);
//   2. workgroup v (v < nb*4) waits for cnt1, reads row v with sc1 loads, reduces it, stores
//      the total Tt[v] (sc1), bumps cnt2;
//   3. every workgroup waits for cnt2, reads the totals (sc1 loads) and runs the SAME sampling
//      step on the SAME RNG stream (kept in LDS) -- redundant, hence no broadcast hop.
// Every spin is bounded (wall-clock timeout -> error word -> all workgroups leave).
#include "gm_common.h"
#include "gm_rng.h"
#include "gm_internal.h"

namespace gm {

// ---- LDS carve (bytes, all multiples of 16) --------------------------------------------
constexpr int L_LUT  = 0;                       // double2[4]   (a,b) per 2-bit code
constexpr int L_VAL  = 64;                      // double[4]    update table of the stopping marker
constexpr int L_CTL  = 96;                      // int[16]      control words
constexpr int L_M    = 160;                     // int[64]      marker ids of the batch
constexpr int L_RNG0 = 416;                     // uint32[624]  current MT block (untempered)
constexpr int L_RNG1 = L_RNG0 + 2496;           // uint32[624]  next MT block
constexpr int L_CASS = L_RNG1 + 2496;           // int[GMAX*KMAX]
constexpr int L_WSUM = L_CASS + GMAX * KMAX * 4;   // double[4][SW_VMAX]
constexpr int L_RED  = L_WSUM + 4 * SW_VMAX * 8;   // double[4]
constexpr int L_END  = L_RED + 64;
// Request > 80 KiB so that exactly one workgroup fits per CU (the hand-off forms used here
// are the ones measured at one workgroup per CU).
constexpr int L_TOTAL = 84 * 1024;
static_assert(L_END <= L_TOTAL, "LDS carve");

enum { C_NDONE = 0, C_UPD, C_MUPD, C_NBNEXT, C_CURSOR, C_OK, C_EMA, C_RNGERR };

size_t sweep_lds_bytes() { return L_TOTAL; }

// Every word another workgroup reads or writes inside the launch is accessed through a
// GLOBAL (address space 1) agent-scope atomic: global_load/store ... sc1, never flat_.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define GM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), GM_RLX_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const gu64*)p, GM_RLX_AGENT));
}
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load((const gu32*)p, GM_RLX_AGENT); }
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, GM_RLX_AGENT); }
__device__ __forceinline__ void add_u32(unsigned* p, unsigned v) { __hip_atomic_fetch_add((gu32*)p, v, GM_RLX_AGENT); }
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The MT stream as the sampling wavefront sees it: two consecutive 624-word blocks in LDS.
struct LdsStream {
    const uint32_t* s0;
    const uint32_t* s1;
    int cursor;
    int* err;
    __device__ __forceinline__ uint32_t peek(int p) const {
        return mt_temper(p < 624 ? s0[p] : s1[(p - 624) < 624 ? (p - 624) : 623]);
    }
    __device__ __forceinline__ uint32_t u32() {
        if (cursor >= 1248) { *err = 1; return 0u; }     // window exhausted: reported, never silent
        return peek(cursor++);
    }
};

// One full MT block step by the whole workgroup: S0 <- S1, S1 <- twist(S1).
__device__ void block_advance(uint32_t* s0, uint32_t* s1, int* ctl, bool copy) {
    const int tid = threadIdx.x;
    if (copy) {
        for (int i = tid; i < 624; i += SW_TPB) s0[i] = s1[i];
        __syncthreads();
    }
    for (int i = tid; i < 227; i += SW_TPB) s1[i] = mt_twist1(s0[i], s0[i + 1], s0[i + 397]);
    __syncthreads();
    for (int i = 227 + tid; i < 454; i += SW_TPB) s1[i] = mt_twist1(s0[i], s0[i + 1], s1[i - 227]);
    __syncthreads();
    for (int i = 454 + tid; i < 623; i += SW_TPB) s1[i] = mt_twist1(s0[i], s0[i + 1], s1[i - 227]);
    __syncthreads();
    if (tid == 0) {
        s1[623] = mt_twist1(s0[623], s1[0], s1[396]);
        if (copy) ctl[C_CURSOR] -= 624;
    }
    __syncthreads();
}

// Bounded wait until *p >= target (monotonic counter).  All threads call; returns false on
// timeout or if another workgroup raised the abort word.
__device__ bool wait_ge(unsigned* p, unsigned target, unsigned* abort_word, int* ctl) {
    __syncthreads();                                  // readers of the previous verdict are done
    if (threadIdx.x == 0) {
        bool ok = true;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while (ld_u32(p) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 127u) == 0u) {
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > 400000000ull;   // 4 s @ 100 MHz
                if (late || ld_u32(abort_word) != 0u) {
                    st_u32(abort_word, 1u);
                    ok = false;
                    break;
                }
            }
        }
        ctl[C_OK] = ok ? 1 : 0;
    }
    __syncthreads();
    return ctl[C_OK] != 0;
}

// bayes.cpp:403-477 for one marker, given num (the dot product + beta*(nonas-1)) and the
// uniform draw: returns the chosen component, the acum value and muk/denom of that component.
template <int K>
__device__ __forceinline__ void decide(double num, double prob, const double* __restrict__ denom_g,
                                       const double* __restrict__ logpi_g, const double* __restrict__ mhl_g,
                                       double inv2sige, int& kc, double& acum_v, double& muk_c, double& denom_c) {
    double muk[K], logl[K];
    muk[0] = 0.0;
    logl[0] = logpi_g[0];
#pragma unroll
    for (int i = 1; i < K; i++) {
        muk[i] = num / denom_g[i];
        logl[i] = logpi_g[i] + (mhl_g[i] + muk[i] * num * inv2sige);
    }
    bool zero_acum = false;
    double tmp1 = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const double d = logl[i] - logl[0];
        if (fabs(d) > 700.0) zero_acum = true;
        tmp1 += exp_(d);
    }
    double acum = zero_acum ? 0.0 : 1.0 / tmp1;
    kc = K - 1;
    bool done = false;
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (!done) {
            if (prob <= acum || i == K - 1) {
                kc = i;
                done = true;
            } else {
                bool zero_inc = false;
#pragma unroll
                for (int j = i + 1; j < K; j++)
                    if (fabs(logl[j] - logl[i + 1 < K ? i + 1 : K - 1]) > 700.0) zero_inc = true;
                if (!zero_inc) {
                    double esum = 0.0;
#pragma unroll
                    for (int k = 0; k < K; k++) esum += exp_(logl[k] - logl[i + 1 < K ? i + 1 : K - 1]);
                    acum = acum + 1.0 / esum;
                }
            }
        }
    }
    acum_v = acum;
    muk_c = 0.0;
    denom_c = 1.0;
#pragma unroll
    for (int i = 1; i < K; i++)
        if (i == kc) { muk_c = muk[i]; denom_c = denom_g[i]; }
}

// The Gibbs step for a whole batch, run by wavefront 0 of EVERY workgroup on identical
// inputs.  Lane j handles batch position j; the walk stops at the first lane whose effect
// may change.  Only workgroup 0 writes per-marker outputs.
template <int K>
__device__ __noinline__ void sample_batch(const SweepArgs& a, int nb, char* smem, bool writer) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    const int* s_m = reinterpret_cast<const int*>(smem + L_M);
    double* s_val = reinterpret_cast<double*>(smem + L_VAL);
    int* s_cass = reinterpret_cast<int*>(smem + L_CASS);
    LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1),
                 ctl[C_CURSOR], &ctl[C_RNGERR]};

    const bool act = lane < nb;
    const int m = act ? s_m[lane] : 0;
    const int g = act ? a.group[m] : 0;
    const double beta_old = act ? a.betas_in[m] : 0.0;
    const bool sig0 = act && (a.sigmag[g] == 0.0);              // bayes.cpp:396-400
    const bool use = act && !sig0;
    const unsigned long long use_mask = __ballot(use);
    const int prefix = __popcll(use_mask & ((1ull << lane) - 1ull));
    const int cursor0 = rs.cursor;
    const double prob = unif_from_word(rs.peek(cursor0 + prefix));   // bayes.cpp:435

    int kc = 0;
    double acum_v = 1.0, muk_c = 0.0, denom_c = 1.0;
    if (use) {
        const double t0 = ld_sc1(&a.Tt[4 * lane + 0]), t1 = ld_sc1(&a.Tt[4 * lane + 1]);
        const double t2 = ld_sc1(&a.Tt[4 * lane + 2]), t3 = ld_sc1(&a.Tt[4 * lane + 3]);
        const double dpa = t0 + t1, dpb = t2 + t3;
        double num = a.msig[m] * (dpa - a.mave[m] * dpb);           // bayes.cpp:765
        num += beta_old * a.nm1;                                     // bayes.cpp:421
        decide<K>(num, prob, a.denom + g * K, a.logpi + g * K, a.mhl + g * K, a.inv2sige, kc, acum_v, muk_c, denom_c);
    }
    const bool stop = use && (kc > 0 || beta_old != 0.0);
    const unsigned long long stop_mask = __ballot(stop);
    const int s = stop_mask ? (__ffsll((long long)stop_mask) - 1) : nb;
    const int n_done = s < nb ? s + 1 : nb;

    if (act && lane < n_done && lane != s) {
        if (sig0) {
            if (writer) { a.acum[m] = 1.0; a.betas_out[m] = 0.0; }
        } else {                                                     // component 0, effect stays 0
            if (writer) {
                a.acum[m] = acum_v; a.betas_out[m] = 0.0; a.comp[m] = 0;
                atomicAdd(&s_cass[g * K + 0], 1);
            }
        }
    }
    if (s < nb && lane == s) {                                       // the stopping marker
        rs.cursor = cursor0 + prefix + 1;
        double beta_new = 0.0;
        if (kc > 0) beta_new = norm(rs, muk_c, a.sigmae / denom_c);  // bayes.cpp:455
        const double dbeta = beta_old - beta_new;                    // bayes.cpp:479
        int upd = 0;
        if (fabs(dbeta) > 0.0) {                                     // bayes.cpp:483, phenotype.cpp:328-329,388
            upd = 1;
            const double bs_ = dbeta * a.msig[m];
            const double mdb = -a.mave[m];
            s_val[0] = (mdb * 1.0 + 2.0) * bs_;
            s_val[1] = (mdb * 0.0 + 0.0) * bs_;
            s_val[2] = (mdb * 1.0 + 1.0) * bs_;
            s_val[3] = (mdb * 1.0 + 0.0) * bs_;
            ctl[C_MUPD] = m;
        }
        if (writer) {
            a.acum[m] = acum_v; a.betas_out[m] = beta_new; a.comp[m] = kc;
            atomicAdd(&s_cass[g * K + kc], 1);
        }
        ctl[C_UPD] = upd;
        ctl[C_CURSOR] = rs.cursor;
        ctl[C_NDONE] = n_done;
    }
    if (s >= nb && lane == 0) {
        ctl[C_UPD] = 0;
        ctl[C_CURSOR] = cursor0 + __popcll(use_mask);
        ctl[C_NDONE] = n_done;
    }
    if (lane == 0) {                                                 // next batch size: ~2x the recent run length
        const int run = s < nb ? s + 1 : 2 * nb;
        const int ema = (3 * ctl[C_EMA] + 16 * run) / 4;            // fixed point, 1/16 marker
        ctl[C_EMA] = ema;
        int nxt = ((2 * ema / 16) + SW_GB - 1) / SW_GB * SW_GB;
        nxt = nxt < SW_GB ? SW_GB : (nxt > SW_BMAX ? SW_BMAX : nxt);
        ctl[C_NBNEXT] = nxt;
    }
}

template <int R> struct Slice {
    static constexpr int NI = 4 * R;                 // individuals per thread
    static constexpr int NW = R >= 4 ? R / 4 : 1;    // 32-bit words per thread per column
};

template <int R>
__device__ __forceinline__ void load_words(const uint8_t* p, uint32_t (&w)[Slice<R>::NW]) {
    if constexpr (R == 1) w[0] = *p;
    else if constexpr (R == 2) w[0] = *reinterpret_cast<const uint16_t*>(p);
    else if constexpr (R == 4) w[0] = *reinterpret_cast<const uint32_t*>(p);
    else if constexpr (R == 8) { const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y; }
    else { const uint4 v = *reinterpret_cast<const uint4*>(p); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
}

// Diagnostic build only (-DGM_SWEEP_PROF): thread 0 of every workgroup accumulates wall-clock
// ticks (100 MHz) per phase; workgroups 0 and W/2 write them to stats[4..]/stats[12..].
#ifdef GM_SWEEP_PROF
#define PROF(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
                                     prof[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PROF(i) do { } while (0)
#endif

template <int R>
__global__ __launch_bounds__(SW_TPB, 1) void k_sweep(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NI = Slice<R>::NI, NW = Slice<R>::NW;
    constexpr int IPW = NI / NW;                     // individuals per word (<= 16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int W = a.W, K = a.K;

    double2* lut = reinterpret_cast<double2*>(smem + L_LUT);
    const double* s_val = reinterpret_cast<const double*>(smem + L_VAL);
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    int* s_m = reinterpret_cast<int*>(smem + L_M);
    uint32_t* s_rng0 = reinterpret_cast<uint32_t*>(smem + L_RNG0);
    uint32_t* s_rng1 = reinterpret_cast<uint32_t*>(smem + L_RNG1);
    int* s_cass = reinterpret_cast<int*>(smem + L_CASS);
    double* s_wsum = reinterpret_cast<double*>(smem + L_WSUM);
    double* s_red = reinterpret_cast<double*>(smem + L_RED);
    unsigned* cnt1 = a.cnt;
    unsigned* cnt2 = a.cnt + 32;
    unsigned* abort_word = a.cnt + 64;

    if (tid < 4) lut[tid] = make_double2(code_a(tid), code_b(tid));
    for (int i = tid; i < 624; i += SW_TPB) s_rng0[i] = a.rng_state[i];
    for (int i = tid; i < a.G * K; i += SW_TPB) s_cass[i] = 0;
    if (tid == 0) {
        ctl[C_CURSOR] = *a.rng_index;
        ctl[C_RNGERR] = 0;
        ctl[C_EMA] = 16 * a.batch_init / 2;
        ctl[C_NBNEXT] = a.batch_init;
        ctl[C_OK] = 1;
    }
    __syncthreads();
    block_advance(s_rng0, s_rng1, ctl, false);       // S1 = twist(S0)

    // ---- this thread's slice of the residual ------------------------------------------
    const size_t b0 = ((size_t)wg * SW_TPB + tid) * R;
    const bool valid = b0 < a.stride;
    double eps[NI], q1[NI], q2[NI];
    uint32_t nam[NW];
    if (valid) {
        load_words<R>(a.namask2 + b0, nam);
#pragma unroll
        for (int i = 0; i < NI; i++) eps[i] = a.eps[4 * b0 + i];
    } else {
#pragma unroll
        for (int w = 0; w < NW; w++) nam[w] = 0u;
#pragma unroll
        for (int i = 0; i < NI; i++) eps[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NI; i++) split2(eps[i], q1[i], q2[i]);
    // codes of NA / out-of-range individuals are forced to 01 (a = b = 0, update value 0)
    uint32_t keep[NW], force[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) { keep[w] = nam[w]; force[w] = ~nam[w] & 0x55555555u; }
    if constexpr (R < 4) {                           // unused high fields of the single word
        constexpr uint32_t used = (R == 1) ? 0xFFu : 0xFFFFu;
        keep[0] &= used; force[0] = (~nam[0] & 0x55555555u & used);
    }

    int pos = 0;
    unsigned gen = 0, tgt2 = 0;
    long long n_upd = 0, n_batch = 0;
#ifdef GM_SWEEP_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memrealtime();
#endif
    int max_nb = 0;
    bool ok = true;

    while (pos < a.M) {
        if (ctl[C_CURSOR] >= 624) block_advance(s_rng0, s_rng1, ctl, true);
        int nb = ctl[C_NBNEXT];
        if (nb > a.M - pos) nb = a.M - pos;
        max_nb = nb > max_nb ? nb : max_nb;
        __syncthreads();
        if (tid < nb) s_m[tid] = a.order[pos + tid];
        __syncthreads();

        PROF(0);   // batch prologue (MT advance, order fetch)
        // ---- phase A: partial dot products of the batch --------------------------------
        for (int g0 = 0; g0 < nb; g0 += SW_GB) {
            uint32_t wd[SW_GB][NW];
#pragma unroll
            for (int gm = 0; gm < SW_GB; gm++) {
                const int j = g0 + gm;
                if (valid && j < nb) {
                    load_words<R>(a.bed + (size_t)s_m[j] * a.stride + b0, wd[gm]);
#pragma unroll
                    for (int w = 0; w < NW; w++) wd[gm][w] = (wd[gm][w] & keep[w]) | force[w];
                } else {
#pragma unroll
                    for (int w = 0; w < NW; w++) wd[gm][w] = 0x55555555u;
                }
            }
            double acc[SW_GB * 4];
#pragma unroll
            for (int gm = 0; gm < SW_GB; gm++) {
                double sa1 = 0.0, sa2 = 0.0, sb1 = 0.0, sb2 = 0.0;
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const uint32_t c = (wd[gm][i / IPW] >> (2 * (i % IPW))) & 3u;
                    const double2 ab = lut[c];
                    sa1 = fma_(ab.x, q1[i], sa1); sa2 = fma_(ab.x, q2[i], sa2);
                    sb1 = fma_(ab.y, q1[i], sb1); sb2 = fma_(ab.y, q2[i], sb2);
                }
                acc[gm * 4 + 0] = sa1; acc[gm * 4 + 1] = sa2; acc[gm * 4 + 2] = sb1; acc[gm * 4 + 3] = sb2;
            }
            // 32 values x 64 lanes -> lane l holds value (l >> 1), summed over the wavefront
#pragma unroll
            for (int half = 16, mask = 32; half >= 1; half >>= 1, mask >>= 1) {
                const bool upper = (lane & mask) != 0;
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const double send = upper ? acc[i] : acc[i + half];
                    const double keepv = upper ? acc[i + half] : acc[i];
                    acc[i] = keepv + __shfl_xor(send, mask, 64);
                }
            }
            acc[0] += __shfl_xor(acc[0], 1, 64);
            if ((lane & 1) == 0) s_wsum[wave * SW_VMAX + g0 * 4 + (lane >> 1)] = acc[0];
        }
        __syncthreads();
        PROF(1);   // phase A: dots
        const int nv = nb * 4;
        if (tid < nv) {
            const double tot = s_wsum[tid] + s_wsum[SW_VMAX + tid] + s_wsum[2 * SW_VMAX + tid] + s_wsum[3 * SW_VMAX + tid];
            st_sc1(&a.P[(size_t)tid * a.Wpad + wg], tot);
        }
        drain_vm();
        __syncthreads();
        if (tid == 0) add_u32(cnt1, 1u);
        PROF(2);   // publish partials + arrive

        // ---- reduce role: workgroup v sums row v over all workgroups ---------------------
        if (wg < nv) {
            if (!wait_ge(cnt1, (unsigned)W * (gen + 1u), abort_word, ctl)) { ok = false; break; }
            for (int v = wg; v < nv; v += W) {
                double x = 0.0;
                for (int w = tid; w < W; w += SW_TPB) x += ld_sc1(&a.P[(size_t)v * a.Wpad + w]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                if (lane == 0) s_red[wave] = x;
                __syncthreads();
                if (tid == 0) st_sc1(&a.Tt[v], s_red[0] + s_red[1] + s_red[2] + s_red[3]);
                __syncthreads();
            }
            drain_vm();
            if (tid == 0) add_u32(cnt2, 1u);
        }
        PROF(3);   // reduce role (incl. waiting for all arrivals)
        tgt2 += (unsigned)(nv < W ? nv : W);
        if (!wait_ge(cnt2, tgt2, abort_word, ctl)) { ok = false; break; }
        PROF(4);   // wait for the totals

        // ---- sampling step (wavefront 0, every workgroup, identical inputs) -------------
        if (wave == 0) {
            switch (K) {
                case 2: sample_batch<2>(a, nb, smem, wg == 0); break;
                case 3: sample_batch<3>(a, nb, smem, wg == 0); break;
                case 4: sample_batch<4>(a, nb, smem, wg == 0); break;
                case 5: sample_batch<5>(a, nb, smem, wg == 0); break;
                case 6: sample_batch<6>(a, nb, smem, wg == 0); break;
                case 7: sample_batch<7>(a, nb, smem, wg == 0); break;
                default: sample_batch<8>(a, nb, smem, wg == 0); break;
            }
        }
        __syncthreads();
        PROF(5);   // sampling step
        if (ctl[C_RNGERR]) { ok = false; break; }

        // ---- phase C: residual update of the stopping marker ----------------------------
        if (ctl[C_UPD]) {
            n_upd++;
            if (valid) {
                uint32_t wd[NW];
                load_words<R>(a.bed + (size_t)ctl[C_MUPD] * a.stride + b0, wd);
#pragma unroll
                for (int w = 0; w < NW; w++) wd[w] = (wd[w] & keep[w]) | force[w];
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const uint32_t c = (wd[i / IPW] >> (2 * (i % IPW))) & 3u;
                    eps[i] += s_val[c];
                    split2(eps[i], q1[i], q2[i]);
                }
            }
        }
        pos += ctl[C_NDONE];
        gen++;
        n_batch++;
        PROF(6);   // residual update
    }

    if (!ok) {
        if (tid == 0) {
            st_u32(abort_word, 1u);
            atomicMax(a.err, ctl[C_RNGERR] ? 2 : 1);
        }
        return;
    }
    __syncthreads();
    if (ctl[C_CURSOR] >= 624) block_advance(s_rng0, s_rng1, ctl, true);
    if (valid) {
#pragma unroll
        for (int i = 0; i < NI; i++) a.eps[4 * b0 + i] = eps[i];
    }
    if (wg == 0) {
        for (int i = tid; i < 624; i += SW_TPB) a.rng_state[i] = s_rng0[i];
        for (int i = tid; i < a.G * K; i += SW_TPB) a.cass[i] = s_cass[i];
        if (tid == 0) {
            *a.rng_index = ctl[C_CURSOR];
            a.stats[0] = n_upd; a.stats[1] = n_batch; a.stats[2] = max_nb; a.stats[3] = 0;
        }
    }
#ifdef GM_SWEEP_PROF
    if (tid == 0 && (wg == 0 || wg == W / 2))
        for (int i = 0; i < 8; i++) a.stats[(wg == 0 ? 4 : 12) + i] = (long long)prof[i];
#endif
}

// Bytes per thread: the smallest R in {1,2,4,8,16} whose grid fits max_wg workgroups.
int sweep_pick_R(size_t stride, int max_wg, int* W_out) {
    for (int R = 1; R <= 4; R *= 2) {
        const size_t per_wg = (size_t)SW_TPB * R;
        const size_t W = (stride + per_wg - 1) / per_wg;
        if (W <= (size_t)max_wg) { *W_out = (int)W; return R; }
    }
    return -1;
}

hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st) {
    const dim3 grid(a.W), block(SW_TPB);
    hipError_t e = hipSuccess;
    switch (R) {
        case 1:
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<1>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_sweep<1>, grid, block, L_TOTAL, st, a);
            break;
        case 2:
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<2>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_sweep<2>, grid, block, L_TOTAL, st, a);
            break;
        case 4:
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<4>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_sweep<4>, grid, block, L_TOTAL, st, a);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gm

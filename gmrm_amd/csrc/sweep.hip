// sweep.hip -- the marker loop of Bayes::process (reference src/bayes.cpp:375-553) for one
// phenotype on one GPU as ONE persistent kernel launch:
//   Bayes::dot_product          src/bayes.cpp:709-770      -> phase A (all workgroups)
//   Gibbs step                  src/bayes.cpp:396-492      -> sample_batch (wavefront 0)
//   Phenotype::update_epsilon   src/phenotype.cpp:326-393  -> phase C (all workgroups)
//
// Layout.  The residual never leaves the chip during a sweep: workgroup w / thread t owns
// R consecutive bytes of every genotype column (4R individuals) and keeps their residual
// eps_i and its two pre-rounded parts (q1_i, q2_i) in VGPRs.  A marker's dot product is then
// 4 (or 2) partial sums per thread, all exact (gm_common.h), reduced by
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
// Pipelining.  While the totals of batch g are in flight, every wavefront already computes and
// publishes the dots of batch g+1, assuming g ends without a residual update; g+1 is then
// promoted, otherwise discarded and a fresh batch starts after the stopping marker.  Speculation
// is switched on only when the recent run length makes P(no update) >~ 1/2.
//
// Genotype stream.  The visit order is known for the whole sweep, so each workgroup keeps a
// 256-position ring of its 256*R-byte column slices in LDS.  Wavefronts 1-2 fetch the slices
// of upcoming positions (coalesced 2R-byte loads, NA mask applied once) one iteration ahead and
// park them in the ring just before the next fetch is issued; phase A and the residual update
// read the ring, never HBM.  Wavefront 3 does the same for the per-marker inputs of the sampling
// step (marker id, group, previous effect, mave, msig) in a 128-position ring.
//
// Exchange per batch (placement-independent, gfx950: private L2 per XCD).  "The data is the
// flag": every exchanged double travels as two 8-byte granules {tag = generation + 1,
// 32 data bits}, each written by ONE sc1 (write-through) store and read by sc1 loads until
// the tag matches -- no counters, no fences.  Generation g uses buffer g & 1.
//   1. every workgroup stores its partial sums (4 per marker, or 2 per marker + 2 per batch in
//      the no-missing-genotype layout) to P[g&1][v][wg];
//   2. workgroup v polls row v, reduces it (exact sums: any order), stores the total Tt[g&1][v];
//   3. wavefront 0 of EVERY workgroup polls the totals and runs the SAME sampling step on the
//      SAME RNG stream (kept in LDS) -- redundant, hence no broadcast hop.
// Every spin is bounded (wall-clock timeout -> error word -> all workgroups leave).
#include "gm_common.h"
#include "gm_rng.h"
#include "gm_internal.h"

namespace gm {

// ---- LDS carve (bytes, all multiples of 16) --------------------------------------------
constexpr int ring_pos(int R) { return R == 4 ? 128 : 256; }   // ring capacity in order positions
constexpr int bmax(int R) { return R == 4 ? 32 : 64; }          // markers per batch (two batches + look-ahead fit the ring)
constexpr int PFN      = 24;                    // positions prefetched per batch per loader thread (register-resident; larger spills)
constexpr int L_LUT  = 0;                       // (spare, 64 B)
constexpr int L_VAL  = 64;                      // double[4]    update table of the stopping marker
constexpr int L_CTL  = 96;                      // int[16]      control words
constexpr int L_M    = 160;                     // spare (diagnostic stamps live at +64)
constexpr int L_RNG0 = 416;                     // uint32[624]  current MT block (untempered)
constexpr int L_RNG1 = L_RNG0 + 2496;           // uint32[624]  next MT block
constexpr int L_CASS = L_RNG1 + 2496;           // int[GMAX*KMAX]
constexpr int L_WSUM = L_CASS + GMAX * KMAX * 4;   // double[4][SW_VMAX]
constexpr int L_RED  = L_WSUM + 4 * SW_VMAX * 8;   // double[4]
constexpr int L_TAB  = L_RED + 64;                 // double[GMAX*(1+3*KMAX)] per-group tables
constexpr int META_POS = 128;                   // per-marker inputs of the sampling step, ring over order positions
constexpr int TAB_LDS = 576;                    // group tables up to 576 doubles live in LDS, larger ones stay in HBM/L2
constexpr int L_META = L_TAB + TAB_LDS * 8;     // int m[128], int g[128], double beta[128], mave[128], msig[128]
constexpr int L_BLUT = L_META + META_POS * 32;  // double[256][4]: a of the 4 genotypes of a byte (rows swizzled), fast layout
constexpr int L_RING = L_BLUT + 256 * 32;       // uint8[ring_pos(R)][SW_TPB*R]
static_assert(L_RING % 16 == 0, "LDS carve");
// Request > 80 KiB so that exactly one workgroup fits per CU.
constexpr int L_MIN = 84 * 1024;
constexpr int lds_total(int R) { return (L_RING + ring_pos(R) * SW_TPB * R) > L_MIN ? (L_RING + ring_pos(R) * SW_TPB * R) : L_MIN; }
static_assert(lds_total(4) <= 160 * 1024, "LDS budget");

enum { C_NDONE = 0, C_UPD, C_SUPD, C_NBNEXT, C_CURSOR, C_EMA, C_RNGERR, C_BAD };

size_t sweep_lds_bytes() { return (size_t)lds_total(4); }

// Every word another workgroup reads or writes inside the launch is accessed through a
// GLOBAL (address space 1) agent-scope atomic: global_load/store ... sc1, never flat_.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define GM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned long long ld_g(const unsigned long long* p) { return __hip_atomic_load((const gu64*)p, GM_RLX_AGENT); }
__device__ __forceinline__ void st_g(unsigned long long* p, unsigned long long v) { __hip_atomic_store((gu64*)p, v, GM_RLX_AGENT); }
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load((const gu32*)p, GM_RLX_AGENT); }
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, GM_RLX_AGENT); }

// one double as two tagged granules
__device__ __forceinline__ void put_value(unsigned long long* g, unsigned tag, double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    st_g(g, ((unsigned long long)tag << 32) | (u & 0xffffffffull));
    st_g(g + 1, ((unsigned long long)tag << 32) | (u >> 32));
}
__device__ __forceinline__ bool get_value(const unsigned long long* g, unsigned tag, double& v) {
    const unsigned long long g0 = ld_g(g), g1 = ld_g(g + 1);
    v = __longlong_as_double((long long)((g0 & 0xffffffffull) | (g1 << 32)));
    return (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
}

// ---- cross-lane helpers for the wavefront reductions (gfx950: v_permlane{16,32}_swap, DPP) ----
__device__ __forceinline__ unsigned lo32(double x) { return (unsigned)(unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ unsigned hi32(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ double mk64(unsigned lo, unsigned hi) {
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// a' = [a.lanes0-31, b.lanes0-31], b' = [a.lanes32-63, b.lanes32-63]
__device__ __forceinline__ void swap32(double& a, double& b) {
    const auto l = __builtin_amdgcn_permlane32_swap(lo32(a), lo32(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi32(a), hi32(b), false, false);
    a = mk64(l[0], h[0]); b = mk64(l[1], h[1]);
}
// rows of 16 lanes: a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3]
__device__ __forceinline__ void swap16(double& a, double& b) {
    const auto l = __builtin_amdgcn_permlane16_swap(lo32(a), lo32(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi32(a), hi32(b), false, false);
    a = mk64(l[0], h[0]); b = mk64(l[1], h[1]);
}
template <int CTRL> __device__ __forceinline__ double dpp64(double x) {
    const int l = __builtin_amdgcn_update_dpp(0, (int)lo32(x), CTRL, 0xf, 0xf, false);
    const int h = __builtin_amdgcn_update_dpp(0, (int)hi32(x), CTRL, 0xf, 0xf, false);
    return mk64((unsigned)l, (unsigned)h);
}
constexpr int DPP_ROW_MIRROR = 0x140;        // lane ^ 15 within a row of 16
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // lane ^ 7
constexpr int DPP_QUAD_3210 = 0x1B;          // lane ^ 3
constexpr int DPP_QUAD_1032 = 0xB1;          // lane ^ 1

// Genotype decode without a table read (an LDS read per individual costs ~70 exposed cycles at
// one wavefront per SIMD): a = {2,0,1,0}[c], b = {1,0,1,1}[c] as IEEE doubles built from the code
// bits -- high word 0x40000000 - (h << 20) masked by !l for a, 0x3FF00000 masked by !(l & !h) for b.
// The values are those of the reference's dotp_lut_a / dotp_lut_b rows (src/dotp_lut.hpp).
__device__ __forceinline__ double code_a_bits(uint32_t w, uint32_t nw, int i) {
    const uint32_t h = (w >> (2 * i + 1)) & 1u;
    const uint32_t nl = (uint32_t)((int)(nw << (31 - 2 * i)) >> 31);        // all ones when the low bit is 0
    return mk64(0u, (0x40000000u - (h << 20)) & nl);
}
__device__ __forceinline__ double code_b_bits(uint32_t present, int i) {   // present: bit 2i set unless code == 01
    const uint32_t pm = (uint32_t)((int)(present << (31 - 2 * i)) >> 31);
    return mk64(0u, 0x3FF00000u & pm);
}

// The reference's 256 x 4 dotp_lut_a rows (4 genotypes per byte), staged in LDS for the fast
// layout: one row = 32 bytes = two ds_read_b128 per genotype byte instead of ~20 VALU decode
// operations.  A row's bank group is row & 7; rows are placed at e ^ swz(e) so that the skewed
// byte distribution of real genotypes spreads over all 8 groups.
__device__ __forceinline__ uint32_t blut_row(uint32_t e) { return e ^ ((e >> 3) & 7u) ^ ((e >> 6) & 3u); }

// 32 per-lane values -> lane l holds value (l >> 1) summed over the 64 lanes.  Each step pairs
// lanes that agree on every earlier selector bit (masks 32, 16, 15, 7, 3, then 1), so the sums
// telescope exactly like an xor butterfly; the sums are exact, so the pairing order is free.
__device__ __forceinline__ double reduce32(double (&acc)[32], int lane) {
#pragma unroll
    for (int i = 0; i < 16; i++) { swap32(acc[i], acc[i + 16]); acc[i] = acc[i] + acc[i + 16]; }
#pragma unroll
    for (int i = 0; i < 8; i++) { swap16(acc[i], acc[i + 8]); acc[i] = acc[i] + acc[i + 8]; }
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double send = up ? acc[i] : acc[i + 4], keep = up ? acc[i + 4] : acc[i];
            acc[i] = keep + dpp64<DPP_ROW_MIRROR>(send);
        }
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const double send = up ? acc[i] : acc[i + 2], keep = up ? acc[i + 2] : acc[i];
            acc[i] = keep + dpp64<DPP_ROW_HALF_MIRROR>(send);
        }
    }
    {
        const bool up = (lane & 2) != 0;
        const double send = up ? acc[0] : acc[1], keep = up ? acc[1] : acc[0];
        acc[0] = keep + dpp64<DPP_QUAD_3210>(send);
    }
    return acc[0] + dpp64<DPP_QUAD_1032>(acc[0]);
}
// two per-lane values -> lanes 0-31 hold sum(a), lanes 32-63 hold sum(b)
__device__ __forceinline__ double reduce2(double a, double b) {
    swap32(a, b);
    double x = a + b;
    x += __shfl_xor(x, 16, 64);
    x += dpp64<DPP_ROW_MIRROR>(x);
    x += dpp64<DPP_ROW_HALF_MIRROR>(x);
    x += dpp64<DPP_QUAD_3210>(x);
    x += dpp64<DPP_QUAD_1032>(x);
    return x;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which
// would make the loader wavefronts wait here for their in-flight genotype prefetches.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Spin budget shared by every poll loop: give up after ~4 s of wall clock or when another
// workgroup has raised the abort word.
struct Spin {
    unsigned long long t0;
    unsigned n;
    __device__ __forceinline__ void start() { t0 = __builtin_amdgcn_s_memrealtime(); n = 0; }
    __device__ __forceinline__ bool expired(unsigned* abort_word) {
#ifndef GM_POLL_SLEEP
#define GM_POLL_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(GM_POLL_SLEEP);
        if ((++n & 63u) != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull || ld_u32(abort_word) != 0u) {
            st_u32(abort_word, 1u);
            return true;
        }
        return false;
    }
};

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

// bayes.cpp:403-445 for one marker, given num (the dot product + beta*(nonas-1)): muk, logl and
// the probability of component 0 (the first value of "acum").  Evaluated by every lane.
template <int K>
__device__ __forceinline__ double decide0(double num, const double* denom_g, const double* logpi_g,
                                          const double* mhl_g, double inv2sige, double (&muk)[K], double (&logl)[K]) {
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
        tmp1 += (d == 0.0) ? 1.0 : exp_(d);          // exp_(0) is exactly 1
    }
    return zero_acum ? 0.0 : 1.0 / tmp1;
}
// bayes.cpp:450-477: the component search.  Only the lane that stops the walk needs it (a lane
// whose draw exceeds acum0 ends with a component > 0, i.e. it is the stopping lane).
template <int K>
__device__ __forceinline__ void decide_rest(double prob, double acum0, const double (&logl)[K], int& kc, double& acum_v) {
    double acum = acum0;
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
                    for (int k = 0; k < K; k++) {
                        const double d = logl[k] - logl[i + 1 < K ? i + 1 : K - 1];
                        esum += (d == 0.0) ? 1.0 : exp_(d);
                    }
                    acum = acum + 1.0 / esum;
                }
            }
        }
    }
    acum_v = acum;
}

// What lane j of the sampling wavefront needs about batch position j; fetched at batch
// start (these loads do not depend on the dots) so they are in registers when the totals land.
struct LaneIn {
    int m, g;
    double beta_old, mave, msig;
};
// Exchange layouts.  General: 4 values per marker (sum a*q1, a*q2, b*q1, b*q2).  Fast (no marker
// of the batch has a missing genotype among the phenotyped individuals, so b == 1 wherever the
// residual is non-zero): 2 values per marker (sum a*q1, a*q2) + 2 per batch (sum q1, sum q2 over
// all individuals = the b-sums of every such marker).
struct SampleOut {                 // global outputs, written by workgroup 0 only
    double* acum;
    double* betas_out;
    int* comp;
};

// The Gibbs step for a whole batch, run by wavefront 0 of EVERY workgroup on identical
// inputs.  Lane j handles batch position j; the walk stops at the first lane whose effect
// may change.  Returns false on a poll timeout.
#ifdef GM_SWEEP_PROF
#define SSTAMP(i) do { if (lane == 0) { unsigned long long* sp_ = reinterpret_cast<unsigned long long*>(smem + L_M + 64); \
                       const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); sp_[i] += t_ - sp_[7]; sp_[7] = t_; } } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

template <int K>
__device__ __forceinline__ void sample_batch_body(int nb, int bmax_, int gran, int nbf16, int G, char* smem, const double* tab, const LaneIn in, double t0, double t1,
                                          double t2, double t3, double sigmae, double inv2sige, double nm1,
                                          const SampleOut out, bool writer) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    double* s_val = reinterpret_cast<double*>(smem + L_VAL);
    int* s_cass = reinterpret_cast<int*>(smem + L_CASS);
    LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1),
                 ctl[C_CURSOR], &ctl[C_RNGERR]};

#ifdef GM_SWEEP_PROF
    if (lane == 0) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[7] = __builtin_amdgcn_s_memrealtime();
#endif
    const bool act = lane < nb;
    const int m = in.m, g = in.g;
    const double beta_old = in.beta_old;
    const bool sig0 = act && (tab[g] == 0.0);                   // bayes.cpp:396-400
    const bool use = act && !sig0;
    const unsigned long long use_mask = __ballot(use);
    const int prefix = __popcll(use_mask & ((1ull << lane) - 1ull));
    const int cursor0 = rs.cursor;
    const double prob = unif_from_word(rs.peek(cursor0 + prefix));   // bayes.cpp:435

    SSTAMP(0);   // inputs, RNG peek
    int kc = 0;
    double acum_v = 1.0, muk_c = 0.0, denom_c = 1.0;
    double muk[K], logl[K];
#pragma unroll
    for (int i = 0; i < K; i++) { muk[i] = 0.0; logl[i] = 0.0; }
    const double* denom_g = tab + G + g * K;
    if (use) {
        const double dpa = t0 + t1, dpb = t2 + t3;
        double num = in.msig * (dpa - in.mave * dpb);               // bayes.cpp:765
        num += beta_old * nm1;                                       // bayes.cpp:421
        acum_v = decide0<K>(num, denom_g, tab + G + G * K + g * K, tab + G + 2 * G * K + g * K, inv2sige, muk, logl);
    }
    SSTAMP(1);   // decide0
    // a lane whose draw exceeds acum0 ends in a component > 0 (bayes.cpp:451,476): it stops the walk
    const bool stop = use && (!(prob <= acum_v) || beta_old != 0.0);
    const unsigned long long stop_mask = __ballot(stop);
    const int s = stop_mask ? (__ffsll((long long)stop_mask) - 1) : nb;
    const int n_done = s < nb ? s + 1 : nb;
    if (s < nb && lane == s && !(prob <= acum_v)) {                 // the component search, one lane
        decide_rest<K>(prob, acum_v, logl, kc, acum_v);
#pragma unroll
        for (int i = 1; i < K; i++)
            if (i == kc) { muk_c = muk[i]; denom_c = denom_g[i]; }
    }

    SSTAMP(2);   // component search
    if (act && lane < n_done && lane != s) {
        if (sig0) {
            if (writer) { out.acum[m] = 1.0; out.betas_out[m] = 0.0; }
        } else if (writer) {                                         // component 0, effect stays 0
            out.acum[m] = acum_v; out.betas_out[m] = 0.0; out.comp[m] = 0;
            atomicAdd(&s_cass[g * K + 0], 1);
        }
    }
    if (s < nb && lane == s) {                                       // the stopping marker
        rs.cursor = cursor0 + prefix + 1;
        double beta_new = 0.0;
        if (kc > 0) beta_new = norm(rs, muk_c, sigmae / denom_c);    // bayes.cpp:455
        const double dbeta = beta_old - beta_new;                    // bayes.cpp:479
        int upd = 0;
        if (fabs(dbeta) > 0.0) {                                     // bayes.cpp:483, phenotype.cpp:328-329,388
            upd = 1;
            const double bs_ = dbeta * in.msig;
            const double mdb = -in.mave;
            s_val[0] = (mdb * 1.0 + 2.0) * bs_;
            s_val[1] = (mdb * 0.0 + 0.0) * bs_;
            s_val[2] = (mdb * 1.0 + 1.0) * bs_;
            s_val[3] = (mdb * 1.0 + 0.0) * bs_;
        }
        if (writer) {
            out.acum[m] = acum_v; out.betas_out[m] = beta_new; out.comp[m] = kc;
            atomicAdd(&s_cass[g * K + kc], 1);
        }
        ctl[C_UPD] = upd;
        ctl[C_SUPD] = s;
        ctl[C_CURSOR] = rs.cursor;
        ctl[C_NDONE] = n_done;
    }
    SSTAMP(3);   // commit + stop lane
    if (s >= nb && lane == 0) {
        ctl[C_UPD] = 0;
        ctl[C_CURSOR] = cursor0 + __popcll(use_mask);
        ctl[C_NDONE] = n_done;
    }
    if (lane == 0) {                                                 // next batch size: ~2x the recent run length
        const int run = s < nb ? s + 1 : 2 * nb;
        const int ema = (3 * ctl[C_EMA] + 16 * run) / 4;            // fixed point, 1/16 marker
        ctl[C_EMA] = ema;
        int nxt = ((nbf16 * ema / 256) + gran - 1) / gran * gran;  // whole register groups only
        nxt = nxt < gran ? gran : (nxt > bmax_ ? bmax_ : nxt);
        ctl[C_NBNEXT] = nxt;
    }
}

// K = 4 (the reference's example mixtures) is inlined into the kernel; other K share out-of-line copies.
template <int K>
__device__ __noinline__ void sample_batch(int nb, int bmax_, int gran, int nbf16, int G, char* smem, const double* tab,
                                          const LaneIn in, double t0, double t1, double t2, double t3, double sigmae,
                                          double inv2sige, double nm1, const SampleOut out, bool writer) {
    sample_batch_body<K>(nb, bmax_, gran, nbf16, G, smem, tab, in, t0, t1, t2, t3, sigmae, inv2sige, nm1, out, writer);
}

// Wavefront 0: lane j polls the totals of batch position j (tagged granules) until they have
// arrived.  Returns false on timeout.
__device__ __forceinline__ bool poll_totals(int nb, bool fast, bool act, const unsigned long long* Ttg, unsigned tag,
                                            double& t0, double& t1, double& t2, double& t3, unsigned* abort_word) {
    const int lane = threadIdx.x & 63;
    Spin sp;
    sp.start();
    bool bad = false;
    // Spin on ONE value (two 8-byte loads per lane per round trip); the other three were stored
    // by neighbouring reducers at about the same time and are normally there on the first look.
    const unsigned long long* g0 = Ttg + 2 * (fast ? 2 * lane + 0 : 4 * lane + 0);
    const unsigned long long* g1 = Ttg + 2 * (fast ? 2 * lane + 1 : 4 * lane + 1);
    const unsigned long long* g2 = Ttg + 2 * (fast ? 2 * nb + 0 : 4 * lane + 2);
    const unsigned long long* g3 = Ttg + 2 * (fast ? 2 * nb + 1 : 4 * lane + 3);
    for (int stage = 0; stage < 2; stage++) {
        for (;;) {
            bool ok = true;
            if (act) {
                if (stage == 0) ok = get_value(g0, tag, t0);
                else { ok = get_value(g1, tag, t1); ok &= get_value(g2, tag, t2); ok &= get_value(g3, tag, t3); }
            }
            if (__all(ok)) break;
            if (sp.expired(abort_word)) { bad = true; break; }
        }
        if (__any(bad)) break;
    }
    return !__any(bad);
}

// ---- per-R storage types ------------------------------------------------------------------
template <int R> struct Slice;
template <> struct Slice<1> { using own_t = uint8_t;  using ld_t = uint16_t; };
template <> struct Slice<2> { using own_t = uint16_t; using ld_t = uint32_t; };
template <> struct Slice<4> { using own_t = uint32_t; using ld_t = unsigned long long; };

// Diagnostic build only (-DGM_SWEEP_PROF): thread 0 of every workgroup accumulates wall-clock
// ticks (100 MHz) per phase; workgroups 0 and W/2 write them to stats[4..]/stats[12..].
#ifdef GM_SWEEP_PROF
#define TRACE(k) do { if (tid == 0 && a.trace && n_batch >= 2000 && n_batch < 2064) \
    a.trace[((size_t)wg * 64 + (size_t)(n_batch - 2000)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRACE(k) do { } while (0)
#endif
#ifdef GM_SWEEP_PROF
#define PROF(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
                                     prof[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PROF(i) do { } while (0)
#endif

template <int R>
__global__ __launch_bounds__(SW_TPB, 1) void k_sweep(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using own_t = typename Slice<R>::own_t;          // this thread's R bytes of a column
    using ld_t = typename Slice<R>::ld_t;            // a loader thread's 2R bytes
    constexpr int NI = 4 * R;                        // individuals per thread
    constexpr ld_t ODD = (ld_t)0x5555555555555555ull;
    constexpr int RPOS = ring_pos(R), BMAX = bmax(R);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int W = a.W, K = a.K, G = a.G;

    const double* s_val = reinterpret_cast<const double*>(smem + L_VAL);
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    uint32_t* s_rng0 = reinterpret_cast<uint32_t*>(smem + L_RNG0);
    uint32_t* s_rng1 = reinterpret_cast<uint32_t*>(smem + L_RNG1);
    int* s_cass = reinterpret_cast<int*>(smem + L_CASS);
    double* s_wsum = reinterpret_cast<double*>(smem + L_WSUM);
    double* s_red = reinterpret_cast<double*>(smem + L_RED);
    double* s_tab = reinterpret_cast<double*>(smem + L_TAB);
    double* blut = reinterpret_cast<double*>(smem + L_BLUT);
    const bool tab_in_lds = G * (1 + 3 * K) <= TAB_LDS;
    const double* tabp = tab_in_lds ? s_tab : a.sigmag;   // sigmag|denom|logpi|mhl, contiguous
    char* ring = smem + L_RING;
    unsigned* abort_word = a.cnt + 64;
    unsigned long long* Pg = reinterpret_cast<unsigned long long*>(a.P);
    unsigned long long* Ttg = reinterpret_cast<unsigned long long*>(a.Tt);

    for (int i = tid; i < 624; i += SW_TPB) s_rng0[i] = a.rng_state[i];
    for (int i = tid; i < G * K; i += SW_TPB) s_cass[i] = 0;
    if (tab_in_lds)
        for (int i = tid; i < G * (1 + 3 * K); i += SW_TPB) s_tab[i] = a.sigmag[i];
    for (int i = tid; i < 1024; i += SW_TPB) blut[blut_row((uint32_t)i >> 2) * 4 + (i & 3)] = code_a(((i >> 2) >> (2 * (i & 3))) & 3);
#ifdef GM_SWEEP_PROF
    if (tid < 8) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[tid] = 0ull;
#endif
    if (tid == 0) {
        ctl[C_CURSOR] = *a.rng_index;
        ctl[C_RNGERR] = 0;
        ctl[C_BAD] = 0;
        ctl[C_EMA] = 16 * a.batch_init / 2;
        ctl[C_NBNEXT] = a.batch_init;
    }
    __syncthreads();
    block_advance(s_rng0, s_rng1, ctl, false);       // S1 = twist(S0)

    // ---- this thread's slice of the residual ------------------------------------------
    const size_t b0 = ((size_t)wg * SW_TPB + tid) * R;
    const bool valid = b0 < a.stride;
    double eps[NI], q1[NI], q2[NI];
    if (valid) {
#pragma unroll
        for (int i = 0; i < NI; i++) eps[i] = a.eps[4 * b0 + i];
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++) eps[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NI; i++) split2(eps[i], q1[i], q2[i]);
    double sq1 = 0.0, sq2 = 0.0;                     // this thread's sum of q1 / q2 (exact)
#pragma unroll
    for (int i = 0; i < NI; i++) { sq1 += q1[i]; sq2 += q2[i]; }

    // ---- loader role (wavefronts 1-2): 2R bytes of every upcoming column ----------------
    // codes of NA / out-of-range individuals are forced to 01 (a = b = 0, update value 0)
    const bool loader = wave == 1 || wave == 2;
    const int lt = loader ? tid - 64 : 0;
    const size_t cb_true = (size_t)wg * SW_TPB * R + (size_t)lt * 2 * R;
    const bool lvalid = loader && cb_true < a.stride;
    const size_t cb = lvalid ? cb_true : 0;           // out-of-range lanes read column byte 0 and mask it away
    ld_t lkeep = 0, lforce = ODD;
    if (lvalid) {
        const ld_t nam = *reinterpret_cast<const ld_t*>(a.namask2 + cb);
        lkeep = nam;
        lforce = (ld_t)(~nam) & ODD;
    }
    ld_t pf[PFN];
    int hi = 0;                                       // ring holds order positions [pos, hi)

    // synchronous ring fill of positions [from, to) (start-up and the rare slow path)
    auto fill = [&](int from, int to) {
        for (int p0 = from; p0 < to; p0 += PFN) {
            const int n = (to - p0) < PFN ? (to - p0) : PFN;
            if (loader) {
#pragma unroll
                for (int i = 0; i < PFN; i++) {            // unconditional loads (clamped index): nothing waits on a select
                    const int pi = p0 + i < a.M ? p0 + i : a.M - 1;
                    pf[i] = *reinterpret_cast<const ld_t*>(a.bed + (size_t)a.order[pi] * a.stride + cb);
                }
#pragma unroll
                for (int i = 0; i < PFN; i++) pf[i] = (pf[i] & lkeep) | lforce;
#pragma unroll
                for (int i = 0; i < PFN; i++)
                    if (i < n) *reinterpret_cast<ld_t*>(ring + (size_t)((p0 + i) & (RPOS - 1)) * (SW_TPB * R) + (size_t)lt * 2 * R) = pf[i];
            }
        }
        __syncthreads();
    };

    // ---- per-marker inputs of the sampling step (marker id, group, previous effect, mave, msig):
    // wavefront 3 fetches them for upcoming order positions (dependent global loads) one
    // iteration ahead and parks them in a 128-position LDS ring, so a restart never waits on them.
    int* mr_m = reinterpret_cast<int*>(smem + L_META);
    int* mr_g = mr_m + META_POS;
    double* mr_beta = reinterpret_cast<double*>(mr_g + META_POS);
    double* mr_mave = mr_beta + META_POS;
    double* mr_msig = mr_mave + META_POS;
    int mhi = 0, npm = 0;                             // meta ring holds positions [pos, mhi)
    int pm_m = 0, pm_g = 0;
    double pm_beta = 0.0, pm_mave = 0.0, pm_msig = 1.0;
    auto meta_issue = [&](int base_pos) {
        int want = base_pos + META_POS < a.M ? base_pos + META_POS : a.M;
        npm = want - mhi;
        if (npm > 64) npm = 64;
        if (npm < 0) npm = 0;
        if (wave == 3) {
            const int pi = mhi + lane < a.M ? mhi + lane : a.M - 1;
            pm_m = a.order[pi];
            pm_g = a.group[pm_m];
            pm_beta = a.betas_in[pm_m];
            pm_mave = a.mave[pm_m];
            pm_msig = a.msig[pm_m];
        }
    };
    auto meta_commit = [&]() {
        if (wave == 3 && lane < npm) {
            const int sl = (mhi + lane) & (META_POS - 1);
            mr_m[sl] = pm_m; mr_g[sl] = pm_g; mr_beta[sl] = pm_beta; mr_mave[sl] = pm_mave; mr_msig[sl] = pm_msig;
        }
        mhi += npm;
        npm = 0;
    };
    auto ensure_meta = [&](int base_pos, int upto) {   // slow path (uniform): meta ring must hold [.., upto)
        if (mhi < upto) meta_commit();
        while (mhi < upto) { meta_issue(base_pos); meta_commit(); }
    };

    // ---- the marker loop, software-pipelined over exchange generations -----------------------
    // While the totals of the current batch are in flight, every wavefront already computes
    // and publishes the dots of the NEXT batch, assuming the current one ends without a residual
    // update.  If it does end that way the next batch is promoted (its partials are already at
    // the reducers); otherwise it is discarded and a fresh batch starts after the stopping
    // marker.  Generation g uses tag g+1 and buffer g&1; a buffer is rewritten only after every
    // workgroup has sampled the generation that used it.
    struct Batch { int p0, nb, nv; unsigned gen; bool fast; };
    int pos = 0;
    unsigned gen_next = 0;
    long long n_upd = 0, n_batch = 0, n_disc = 0;
    int max_nb = 0;
    bool ok = true;
#ifdef GM_SWEEP_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memrealtime();
#endif
    int npf = 0;

    auto prefetch_issue = [&](int limit) {
        int want = limit < a.M ? limit : a.M;
        if (want > pos + RPOS) want = pos + RPOS;
        npf = want - hi;
        if (npf > PFN) npf = PFN;
        if (npf < 0) npf = 0;
        if (loader) {
            // chunks of 8 behind wave-uniform branches: only about npf loads are issued, none waits
            // on a select (indices are clamped, surplus lanes of the last chunk are dropped at commit)
#pragma unroll
            for (int c8 = 0; c8 < PFN; c8 += 8) {
                if (c8 < npf) {
#pragma unroll
                    for (int i = c8; i < c8 + 8; i++) {
                        const int pi = hi + i < a.M ? hi + i : a.M - 1;
                        pf[i] = *reinterpret_cast<const ld_t*>(a.bed + (size_t)a.order[pi] * a.stride + cb);
                    }
                }
            }
        }
    };
    // park the slices fetched by the previous prefetch_issue in the ring (readers see them after
    // the next barrier).  Called late -- right before the next issue -- so nobody waits for HBM.
    auto prefetch_commit = [&]() {
        if (loader) {
#pragma unroll
            for (int i = 0; i < PFN; i++)
                if (i < npf) *reinterpret_cast<ld_t*>(ring + (size_t)((hi + i) & (RPOS - 1)) * (SW_TPB * R) + (size_t)lt * 2 * R) = (pf[i] & lkeep) | lforce;
        }
        hi += npf;
        npf = 0;
    };
    auto ensure = [&](int upto) {                      // slow path: ring must hold [pos, upto)
        if (hi < upto) prefetch_commit();
        if (hi < upto) { fill(hi, upto); hi = upto; }
    };

    // phase A for positions [b.p0, b.p0 + b.nb) (slices already in the ring) + publish
    auto compute_publish = [&](Batch& b, LaneIn& li) {
        lds_barrier();                                // ring writes are visible (no vmcnt drain)
        li = LaneIn{0, 0, 0.0, 0.0, 1.0};
        if (wave == 0 && lane < b.nb) {
            const int sl = (b.p0 + lane) & (META_POS - 1);
            li.m = mr_m[sl]; li.g = mr_g[sl]; li.beta_old = mr_beta[sl]; li.mave = mr_mave[sl]; li.msig = mr_msig[sl];
        }
        const bool fast = a.all_nomiss != 0;          // exchange layout, fixed per launch
        const int nb = b.nb, p0 = b.p0;
        max_nb = nb > max_nb ? nb : max_nb;
        if (fast) {
            // b == 1 wherever the residual is non-zero: only the a-sums depend on the marker
            for (int g0 = 0; g0 < nb; g0 += 2 * SW_GB) {
                double acc[32];
#pragma unroll
                for (int gm = 0; gm < 2 * SW_GB; gm++) {
                    // unconditional ring read: slots past the batch hold finite junk that nobody consumes
                    const uint32_t wd = *reinterpret_cast<const own_t*>(ring + (size_t)((p0 + g0 + gm) & (RPOS - 1)) * (SW_TPB * R) + (size_t)tid * R);
                    double sa1 = 0.0, sa2 = 0.0;
#pragma unroll
                    for (int k = 0; k < R; k++) {                  // one table row per genotype byte
                        const double* row = blut + blut_row((wd >> (8 * k)) & 0xFFu) * 4;
                        const double2 a01 = *reinterpret_cast<const double2*>(row);
                        const double2 a23 = *reinterpret_cast<const double2*>(row + 2);
                        sa1 = fma_(a01.x, q1[4 * k + 0], sa1); sa2 = fma_(a01.x, q2[4 * k + 0], sa2);
                        sa1 = fma_(a01.y, q1[4 * k + 1], sa1); sa2 = fma_(a01.y, q2[4 * k + 1], sa2);
                        sa1 = fma_(a23.x, q1[4 * k + 2], sa1); sa2 = fma_(a23.x, q2[4 * k + 2], sa2);
                        sa1 = fma_(a23.y, q1[4 * k + 3], sa1); sa2 = fma_(a23.y, q2[4 * k + 3], sa2);
                    }
                    acc[gm * 2 + 0] = sa1; acc[gm * 2 + 1] = sa2;
                }
                const double r = reduce32(acc, lane);
                if ((lane & 1) == 0) s_wsum[wave * SW_VMAX + g0 * 2 + (lane >> 1)] = r;
            }
            const double r2 = reduce2(sq1, sq2);
            if (lane == 0) s_wsum[wave * SW_VMAX + 2 * nb] = r2;
            if (lane == 32) s_wsum[wave * SW_VMAX + 2 * nb + 1] = r2;
        } else {
            for (int g0 = 0; g0 < nb; g0 += SW_GB) {
                double acc[32];
#pragma unroll
                for (int gm = 0; gm < SW_GB; gm++) {
                    const uint32_t wd = *reinterpret_cast<const own_t*>(ring + (size_t)((p0 + g0 + gm) & (RPOS - 1)) * (SW_TPB * R) + (size_t)tid * R);
                    double sa1 = 0.0, sa2 = 0.0, sb1 = 0.0, sb2 = 0.0;
                    const uint32_t nw = ~wd;
                    const uint32_t present = ~(wd & ~(wd >> 1));               // bit 2i clear only for code 01
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        const double av = code_a_bits(wd, nw, i), bv = code_b_bits(present, i);
                        sa1 = fma_(av, q1[i], sa1); sa2 = fma_(av, q2[i], sa2);
                        sb1 = fma_(bv, q1[i], sb1); sb2 = fma_(bv, q2[i], sb2);
                    }
                    acc[gm * 4 + 0] = sa1; acc[gm * 4 + 1] = sa2; acc[gm * 4 + 2] = sb1; acc[gm * 4 + 3] = sb2;
                }
                const double r = reduce32(acc, lane);
                if ((lane & 1) == 0) s_wsum[wave * SW_VMAX + g0 * 4 + (lane >> 1)] = r;
            }
        }
        __syncthreads();
        const int nv = fast ? 2 * nb + 2 : 4 * nb;
        if (tid < nv) {
            const double tot = s_wsum[tid] + s_wsum[SW_VMAX + tid] + s_wsum[2 * SW_VMAX + tid] + s_wsum[3 * SW_VMAX + tid];
            put_value(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)tid * a.Wpad + wg), b.gen + 1u, tot);
        }
        b.fast = fast;
        b.nv = nv;
    };

    // reduce role: workgroup v sums row v of generation b.gen over all workgroups
    auto reduce_role = [&](const Batch& b) -> bool {
        bool bad = false;
        if (wg < b.nv) {
            const unsigned long long* Pb = Pg + 2 * (size_t)(b.gen & 1u) * SW_VMAX * a.Wpad;
            unsigned long long* Tb = Ttg + 2 * (size_t)(b.gen & 1u) * SW_VMAX;
            for (int v = wg; v < b.nv; v += W) {
                double x = 0.0;
                if (tid < W) {
                    Spin sp;
                    sp.start();
                    const unsigned long long* gp = Pb + 2 * ((size_t)v * a.Wpad + tid);
                    while (!get_value(gp, b.gen + 1u, x)) {
                        if (sp.expired(abort_word)) { bad = true; x = 0.0; break; }
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                if (lane == 0) s_red[wave] = x;
                lds_barrier();
                if (tid == 0) put_value(Tb + 2 * v, b.gen + 1u, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
                lds_barrier();
            }
        }
        return bad;
    };

    Batch cur{0, 0, 0, 0u, false}, nxt{0, 0, 0, 0u, false}, tb{0, 0, 0, 0u, false};
    LaneIn li_cur{0, 0, 0.0, 0.0, 1.0}, li_nxt{0, 0, 0.0, 0.0, 1.0}, tl{0, 0, 0.0, 0.0, 1.0};
    bool bad = false;
    {
        int first = 3 * BMAX < a.M ? 3 * BMAX : a.M;
        if (first > RPOS) first = RPOS;
        fill(0, first);
        hi = first;
        ensure_meta(0, META_POS < a.M ? META_POS : a.M);
        __syncthreads();
    }
    // One compute site and one reduce site serve the three cases (first batch, restart after a
    // residual update, speculative next batch); `restart` / `need_reduce` select the case.
    bool restart = true, need_reduce = false, have_next = false;
    while (pos < a.M) {
        TRACE(restart ? 0 : 7);
        if (need_reduce) { bad = reduce_role(cur); need_reduce = false; TRACE(2); }
        PROF(0);   // reduce role
        bool do_compute = false;
        if (restart) {
            int nb0 = ctl[C_NBNEXT];
            if (nb0 > a.M - pos) nb0 = a.M - pos;
            tb.p0 = pos; tb.nb = nb0; tb.gen = gen_next++;
            do_compute = true;
        } else if (ctl[C_EMA] >= a.spec_factor16 * cur.nb) {   // speculate only when the recent run length (1/16 units)
            const int p1 = pos + cur.nb;              // is >= 1.5 batches: P(no residual update) >~ 1/2
            if (p1 < a.M) {
                int nb1 = ctl[C_NBNEXT];
                if (nb1 > a.M - p1) nb1 = a.M - p1;
                tb.p0 = p1; tb.nb = nb1; tb.gen = gen_next++;
                do_compute = true;
            }
        }
        if (do_compute) {
            ensure(tb.p0 + tb.nb);
            ensure_meta(pos, tb.p0 + tb.nb);
            compute_publish(tb, tl);
        }
        if (restart) TRACE(1);
        PROF(1);   // dots + publish (restart: on the critical path; speculative: overlaps the exchange)
        if (restart) {
            cur = tb; li_cur = tl;
            restart = false;
            need_reduce = true;
            continue;
        }
        have_next = do_compute;
        if (have_next) { nxt = tb; li_nxt = tl; }
        prefetch_commit();
        meta_commit();
        prefetch_issue(pos + cur.nb + (have_next ? nxt.nb : 0) + BMAX);
        meta_issue(pos);
        PROF(2);   // ring write of the previous prefetch + issue of the next

        // ---- sampling step of the current batch (wavefront 0, every workgroup, identical inputs)
        if (wave == 0) {
            const SampleOut so{a.acum, a.betas_out, a.comp};
            const unsigned long long* Tb = Ttg + 2 * (size_t)(cur.gen & 1u) * SW_VMAX;
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            const bool okw = poll_totals(cur.nb, cur.fast, lane < cur.nb, Tb, cur.gen + 1u, t0, t1, t2, t3, abort_word);
            TRACE(3);
            PROF(4);   // wait for the totals
            bad |= !okw;
            if (okw) {
                switch (K) {
                    case 2: sample_batch<2>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                    case 3: sample_batch<3>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                    case 4: sample_batch_body<4>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                    case 5: sample_batch<5>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                    case 6: sample_batch<6>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                    case 7: sample_batch<7>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                    default: sample_batch<8>(cur.nb, BMAX, a.all_nomiss ? 2 * SW_GB : SW_GB, a.nb_factor16, G, smem, tabp, li_cur, t0, t1, t2, t3, a.sigmae, a.inv2sige, a.nm1, so, wg == 0); break;
                }
            }
        }
        PROF(7);   // sampling step (wavefront 0's own time)
        if (bad) ctl[C_BAD] = 1;
        lds_barrier();                                // no vmcnt drain: prefetches stay in flight
        if (ctl[C_BAD] || ctl[C_RNGERR]) { ok = false; break; }
        TRACE(5);
        PROF(5);   // barrier after sampling

        // ---- phase C: residual update of the stopping marker (its slice is in the ring) ----
        const int n_done = ctl[C_NDONE];
        const bool upd = ctl[C_UPD] != 0;
        if (upd) {
            n_upd++;
            const uint32_t wd = *reinterpret_cast<const own_t*>(ring + (size_t)((pos + ctl[C_SUPD]) & (RPOS - 1)) * (SW_TPB * R) + (size_t)tid * R);
#pragma unroll
            for (int i = 0; i < NI; i++) {
                eps[i] += s_val[(wd >> (2 * i)) & 3u];
                split2(eps[i], q1[i], q2[i]);
            }
            sq1 = 0.0; sq2 = 0.0;
#pragma unroll
            for (int i = 0; i < NI; i++) { sq1 += q1[i]; sq2 += q2[i]; }
        }
        pos += n_done;
        n_batch++;
        TRACE(6);
        PROF(6);   // residual update + ring write
        if (ctl[C_CURSOR] >= 624) block_advance(s_rng0, s_rng1, ctl, true);
        if (upd || n_done < cur.nb || !have_next) {   // the speculative batch (if any) is stale: restart
            if (have_next) n_disc++;
            restart = true;
        } else {                                      // promote: its partials are already at the reducers
            cur = nxt;
            li_cur = li_nxt;
            need_reduce = true;
        }
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
        for (int i = tid; i < G * K; i += SW_TPB) a.cass[i] = s_cass[i];
        if (tid == 0) {
            *a.rng_index = ctl[C_CURSOR];
            a.stats[0] = n_upd; a.stats[1] = n_batch; a.stats[2] = max_nb; a.stats[3] = n_disc;
        }
    }
#ifdef GM_SWEEP_PROF
    if (tid == 0 && (wg == 0 || wg == W / 2)) {
        for (int i = 0; i < 8; i++) a.stats[(wg == 0 ? 4 : 12) + i] = (long long)prof[i];
        if (wg != 0) for (int i = 0; i < 4; i++) a.stats[20 + i] = (long long)reinterpret_cast<unsigned long long*>(smem + L_M + 64)[i];
    }
#endif
}

// Bytes per thread: the smallest R in {1,2,4} whose grid fits max_wg (<= 256) workgroups.
int sweep_pick_R(size_t stride, int max_wg, int* W_out) {
    if (max_wg > SW_TPB) max_wg = SW_TPB;            // one reducer thread per workgroup
    for (int R = 1; R <= 4; R *= 2) {
        const size_t per_wg = (size_t)SW_TPB * R;
        const size_t W = (stride + per_wg - 1) / per_wg;
        if (W <= (size_t)max_wg) { *W_out = (int)W; return R; }
    }
    return -1;
}

template <int R> static hipError_t launch_R(const SweepArgs& a, hipStream_t st) {
    const int lds = lds_total(R);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sweep<R>, dim3(a.W), dim3(SW_TPB), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st) {
    if (a.W > SW_TPB) return hipErrorInvalidValue;
    switch (R) {
        case 1: return launch_R<1>(a, st);
        case 2: return launch_R<2>(a, st);
        case 4: return launch_R<4>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gm

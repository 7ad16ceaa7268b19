// sweep.hip -- the marker loop of Bayes::process (reference src/bayes.cpp:375-553) for one
// phenotype on one GPU as ONE persistent kernel launch:
//   Bayes::dot_product          src/bayes.cpp:709-770      -> phase A (all workgroups)
//   Gibbs step                  src/bayes.cpp:396-492      -> sample_batch (wavefront 0)
//   Phenotype::update_epsilon   src/phenotype.cpp:326-393  -> phase C (all workgroups)
//
// Layout.  The residual never leaves the chip during a sweep: workgroup w owns SB = 256*R
// consecutive bytes of every genotype column (4*SB individuals); every thread keeps the residual
// eps_i of 4R of them in VGPRs (for the update).  For the dot products the two exact parts
// of every residual (gm_common.h: q1 on the 2^-22 grid, q2 on the 2^-53 grid, both < 2^31 grid
// units) are ALSO kept in LDS as 31-bit integers cut into four signed base-256 digits: eight "digit
// planes" of one byte per individual.
//
// Phase A is an int8 contraction on the matrix cores (v_mfma_i32_16x16x64_i8): 16 markers of the
// batch x 8 digit planes x 64 individuals per instruction, integer accumulation.  The genotype
// slices sit in the LDS ring RECODED so that a 2-bit field is the reference's dotp_lut_a value
// (src/dotp_lut.hpp: a = 2, 0, 1, 0 for codes 00, 01, 10, 11) and one v_and per register turns a
// dword of the ring into an operand register; the missing-genotype layout adds a second pass with
// the 0/1 indicator of code 01.  Integer sums are exact, so the results are bit-identical to the
// f64 formulation (oracle "canon" mode).  The four wavefronts split the batch's tiles of 16 markers
// (and, for small batches, the slice); partial sums meet in LDS through 64-bit integer atomics.
//
// Schedule.  The chain is sequential (marker j+1 needs the residual after marker j), and a
// grid-wide exchange costs microseconds on an 8-XCD part, so markers are processed in
// speculative batches: the dots of the next nb markers of the shuffled order are computed
// against the current residual in one pass; wavefront 0 then walks the batch in order and
// stops at the first marker whose effect changes (dbeta != 0), because every later dot in
// the batch is then stale.  The residual update is applied and the next batch starts after
// that marker.  Results are exactly those of the one-marker-at-a-time loop.
//
// Genotype stream (round 4).  The visit order is known for the whole sweep, so each workgroup keeps a WINDOW of upcoming
// order positions on chip, in tiles of 16 positions (tile T = positions 16 T .. 16 T + 15, what one MFMA takes as rows).
// A tile has a fixed home by its number: most live in LDS (15 tile slots of 8 KB at R = 2, 16-byte chunks XOR-swizzled by
// position so that 64 lanes reading 64 slices hit different banks), three of every eight live in the REGISTERS of
// wavefronts 1-3 (one each, in the lane layout the matrix instruction takes operand A in: 32 AGPRs per tile at R = 2) --
// the register file of a compute unit is three times its LDS, and a batch may only be as long as what is resident.
// HBM holds the columns in the device code (gm_common.h: the 2-bit field is the genotype value), so nothing is recoded on
// the way: LDS tiles are filled by global_load_lds_dwordx4 (HBM -> LDS, no register pass, the swizzle applied to the
// source address), register tiles by global_load_dwordx4 into AGPRs.  The loads of the tiles a walk has freed are issued
// by wavefronts 1-3 while wavefront 0 samples the next batch, and are waited for one round later: nobody waits for HBM.
// (Individuals without a phenotype have residual 0, hence all-zero digit planes: phase A needs no mask; the update
// applies it.)
// Wavefronts 1-3 do the same for the per-marker inputs of the sampling step (marker id, group,
// previous effect, mave, msig): a 256-position ring in LDS.
//
// Exchange per batch (placement-independent, gfx950: private L2 per XCD).  "The data is the
// flag": every exchanged double travels as two 8-byte granules {32 data bits, tag = generation + 1}
// side by side, written by ONE 16-byte sc1 (write-through) store and read by 16-byte sc1 loads
// until both tags match -- no counters, no fences.  Generation g uses buffer g & 1.
//   1. every workgroup stores its partial sums (4 per marker, or 2 per marker + 2 per batch in
//      the no-missing-genotype layout) to P[g&1][v][wg];
//   2. workgroup v polls row v, reduces it (exact sums: any order), stores the total Tt[g&1][v];
//   3. wavefront 0 of EVERY workgroup fetches the whole row of totals (one round trip per look) and
//      runs the SAME sampling step on the SAME RNG stream (kept in LDS) -- redundant, hence no
//      broadcast hop.
// Every spin is bounded (wall-clock timeout -> error word -> all workgroups leave).
#include "gm_common.h"
#include "gm_rng.h"
#include "gm_internal.h"
#include <type_traits>

namespace gm {

#ifndef GM_ROWS_BY_LANE
#define GM_ROWS_BY_LANE 1         // 0: the long-batch kernel stores a tile's rows one after the other (A/B builds)
#endif
#ifndef GM_PACK_ROWS
#define GM_PACK_ROWS 1            // 0: the kernels other than the long-batch ones exchange one value per granule pair (rounds 1-3; A/B builds)
#endif

// ---- geometry per R (bytes of a column per thread) -------------------------------------------
template <int R> struct Geo {
    static constexpr int SB = SW_TPB * R;                 // slice bytes per workgroup (= genotype bytes = plane records)
    static constexpr int CPP = SB / 16;                   // 16-byte chunks per position
    static constexpr int SS = CPP / 4;                    // super-steps of phase A: 4 chunks = 256 individuals, one chunk per lane group
    static constexpr int TILE_B = 16 * SB;                // bytes of a tile: the slices of 16 consecutive order positions
    static constexpr int GPT = TILE_B / 1024;             // global_load_lds_dwordx4 instructions per LDS tile (1 KiB per wave-instruction)
    static constexpr int PPG = 16 / GPT;                  // positions one of them covers
    // Homes of the tiles (by tile number T): RPER of every 8 consecutive tiles live in registers -- tile T with T & 7 = w in
    // the registers of wavefront w = 1..3, slot (T >> 3) % NP -- the others in LDS, slot (index among the LDS tiles) % nl.
    // At R = 1 a tile is 4 KB and LDS holds more tiles than a batch can use: no register tiles.  At R = 4 a register tile
    // costs 64 registers: one slot per wavefront.
    static constexpr int RPER = R == 1 ? 0 : 3;
    static constexpr int NP = R == 2 ? 3 : (R == 4 ? 1 : 0);
    static constexpr int NLMAX = R == 1 ? 30 : (R == 2 ? 15 : 7);    // LDS tile slots at most (what fits beside the rest decides: carve_for)
    // Planes of operand B (one byte per individual): the seven digit planes of the residual (four signed base-256 digits of
    // its part on the 2^-22 grid, three of the rest on the 2^-44 grid) and, in the kernels that cross stops, two planes for the
    // genotype values of those markers (NSTOP).  Plane n starts at n * PSTRIDE + (n >> 2) * 64 bytes (n < 7), stop plane s at
    // (7 + s) * PSTRIDE + 64: in phase A a 16-lane LDS access group reads 16 bytes of each plane, and these offsets
    // put the reads of the nine planes on different bank quads (0,16,32,48,128,144,160,176,192 mod 256).
    static constexpr int PSTRIDE = 4 * SB + 16;
    static_assert(GPT * PPG == 16 && PPG * SB == 1024, "LDS-DMA mapping");
};
template <int R, bool CONT> constexpr int planes_bytes() { return (7 + (CONT ? NSTOP : 0)) * Geo<R>::PSTRIDE + 64; }
// the longest batch: 240 markers where two values per marker are exchanged and nothing is crossed (four passes of the sampling
// wavefront); 128 in the other layouts (two passes, as before: their batches end at the first marker in the model or run out
// of exchange slots long before)
template <bool LONG> constexpr int batch_cap() { return LONG ? 240 : 128; }

// ---- LDS carve (bytes, all multiples of 16) --------------------------------------------
constexpr int L_CTL  = 96;                      // int[16]      control words
constexpr int L_RED  = 160;                     // double[8]    reducer scratch (two rows)
constexpr int L_WSQ  = 224;                     // double[4][2] per-wavefront sum of q1 / q2
constexpr int L_AB   = 288;                     // int[4]       per-wavefront sums of a crossed stop's two planes (all-dirty layout)
[[maybe_unused]] constexpr int L_M = 288;       // diagnostic stamps (64 B at +64)
constexpr int L_RNG0 = 416;                     // uint32[624]  current MT block (untempered)
constexpr int L_RNG1 = L_RNG0 + 2496;           // uint32[624]  next MT block
constexpr int L_SUM  = L_RNG1 + 2496;           // int64[SW_VMAX]  per-batch integer sums of this workgroup (plain stores: a tile has one owner)
constexpr int META_POS = 256;                   // per-marker inputs of the sampling step, ring over order positions
constexpr int L_META = L_SUM + SW_VMAX * 8;     // int m[256], int g[256], double beta[256], mave[256], msig[256]
constexpr int L_NM   = L_META + META_POS * 32;  // uint8[256]: "no missing genotype" flag of the marker at each ring position
constexpr int L_TOT  = L_NM + META_POS;         // double[SW_VMAX]: the batch totals as wavefront 0 fetched them.  The same bytes serve, at
                                                // other times of the round: the reducers' row accumulators (between publish and poll) and
                                                // the staged slices of register-home markers for the residual update (after the walk)
constexpr int L_ZSP  = L_TOT + SW_VMAX * 8;     // int64[16][2]: missing-genotype terms of a batch's few dirty markers (sparse_z)
constexpr int L_ZNX  = L_ZSP + 16 * 16;         // double[129] (+ pad): the x table of the normal ziggurat (gm_rng.h), copied at kernel start
constexpr int L_UPD  = L_ZNX + 130 * 8;         // the residual updates of the round: int n, pos[3]; double val[3][4] (by device code)
constexpr int L_VAR  = L_UPD + 16 + 96 + 16;    // from here on the carve depends on G, K (carve_for): component counts int[G*K],
                                                // per-group tables double[G*(1+3K)], the planes, the LDS tiles
static_assert(L_VAR % 16 == 0, "LDS carve");
constexpr int L_TOTAL = 160 * 1024;
constexpr int NSTAGE = 3;                       // slices staged for one round's residual updates (the crossed stops and the last one)

// The part of the carve that depends on the launch (number of groups and components, the kernel's planes): offsets, the number
// of LDS tile slots that fit and the tile window that follows from it.
struct Carve { int cass, tab, pln, ring, nl, win; unsigned nl_magic; };
// the longest run of consecutive tiles in which every tile has a slot of its own: at most nl LDS tiles, at most NP register tiles
// of any one wavefront (pattern of Geo: tiles T & 7 = 1, 2, 3 in registers when RPER == 3)
static int window_tiles(int rper, int nl, int np) {
    for (int w = 96; w >= 1; w--) {
        bool ok = true;
        for (int start = 0; start < 8 && ok; start++) {
            int nlds = 0, nreg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = start; t < start + w; t++) {
                const int j = t & 7;
                if (rper && j >= 1 && j <= 3) nreg[j]++; else nlds++;
            }
            if (nlds > nl) ok = false;
            for (int j = 1; j <= 3; j++) if (nreg[j] > np) ok = false;
        }
        if (ok) return w;
    }
    return 0;
}
template <int R, bool CONT, bool LONGB> static Carve carve_for(int G, int K) {
    Carve c;
    c.cass = L_VAR;
    c.tab = c.cass + (G * K * 4 + 15) / 16 * 16;
    const int tabd = G * (1 + 3 * K);
    c.pln = c.tab + (tabd * 8 + 15) / 16 * 16;        // always in LDS: large tables take tile slots
    c.ring = (c.pln + planes_bytes<R, CONT>() + 1023) / 1024 * 1024;          // LDS-DMA destinations: 1 KiB blocks
    int nl = (L_TOTAL - c.ring) / Geo<R>::TILE_B;
    if (nl > Geo<R>::NLMAX) nl = Geo<R>::NLMAX;
    c.nl = nl < 1 ? 0 : nl;
    c.win = c.nl ? window_tiles(LONGB ? Geo<R>::RPER : 0, c.nl, LONGB ? Geo<R>::NP : 0) : 0;
    c.nl_magic = c.nl ? (unsigned)((1ull << 32) / (unsigned)c.nl) + 1u : 0u;    // x % nl = x - nl * umulhi(x, magic) for x < 2^32 / nl (x < 2^22 here)
    return c;
}

enum { C_NDONE = 0, C_UPD, C_SUPD, C_NBNEXT, C_CURSOR, C_EMA, C_RNGERR, C_BAD, C_TOTF, C_RANGE, C_PLN, C_SCRMIN, C_NSCRT, C_NSCR, C_XFLAG };
static_assert(C_XFLAG < 16, "control words");

size_t sweep_lds_bytes() { return (size_t)L_TOTAL; }

// Every word another workgroup reads or writes inside the launch is accessed through a
// GLOBAL (address space 1) agent-scope atomic: global_load/store ... sc1, never flat_.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define GM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned long long ld_g(const unsigned long long* p) { return __hip_atomic_load((const gu64*)p, GM_RLX_AGENT); }
__device__ __forceinline__ void st_g(unsigned long long* p, unsigned long long v) { __hip_atomic_store((gu64*)p, v, GM_RLX_AGENT); }
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load((const gu32*)p, GM_RLX_AGENT); }
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, GM_RLX_AGENT); }

// one double as two tagged granules {32 data bits, tag}; the pair is 16-byte aligned and moves
// as ONE 16-byte sc1 store / load (half the requests of two 8-byte ones; the tags make tearing
// between the granules harmless)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put_value(unsigned long long* g, unsigned tag, double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const u32x4 d = {(unsigned)u, tag, (unsigned)(u >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ bool get_value(const unsigned long long* g, unsigned tag, double& v) {
    u32x4 d;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(g) : "memory");
    v = __longlong_as_double((long long)(((unsigned long long)d.z << 32) | d.x));
    return d.y == tag && d.w == tag;
}
// two values in one round trip (a reducer workgroup with two rows)
__device__ __forceinline__ void get_value2(const unsigned long long* g0, const unsigned long long* g1, unsigned tag,
                                           double& v0, bool& ok0, double& v1, bool& ok1) {
    u32x4 d0, d1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc1\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(d0), "=&v"(d1) : "v"(g0), "v"(g1) : "memory");
    if (!ok0 && d0.y == tag && d0.w == tag) { ok0 = true; v0 = __longlong_as_double((long long)(((unsigned long long)d0.z << 32) | d0.x)); }
    if (!ok1 && d1.y == tag && d1.w == tag) { ok1 = true; v1 = __longlong_as_double((long long)(((unsigned long long)d1.z << 32) | d1.x)); }
}
// Packed partial sums (the long-batch kernel, round 4): BOTH exact parts of one marker's sum over ONE workgroup's slice in
// a single 16-byte granule pair, so that a batch of 240 markers exchanges 241 rows instead of 482 (one row per reducer
// workgroup, one store per publishing thread, as with batches of 120).  A slice's part on the 2^-22 grid is below 2^43 units
// (|k| <= 2^30 per individual, 4096 individuals, a <= 2) and the rest below 2^34 units: 44 + 35 = 79 bits beside two 24-bit tags:
//   granule 0 = {part1 bits 0..39, tag}     granule 1 = {part1 bits 40..43 | part2 << 4, tag}
// (tags count the rounds of a launch modulo 2^24; gmrm_sweep_launch refuses launches of 2^24 markers or more.)
__device__ __forceinline__ void put_packed(unsigned long long* g, unsigned tag24, long long p1, long long p2) {
    const unsigned long long M40 = (1ull << 40) - 1ull;
    const unsigned long long g0 = ((unsigned long long)p1 & M40) | ((unsigned long long)tag24 << 40);
    const unsigned long long g1 = ((((unsigned long long)p1 >> 40) & 0xFull) | (((unsigned long long)p2 & ((1ull << 35) - 1ull)) << 4)) | ((unsigned long long)tag24 << 40);
    const u32x4 d = {(unsigned)g0, (unsigned)(g0 >> 32), (unsigned)g1, (unsigned)(g1 >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ bool get_packed(const unsigned long long* g, unsigned tag24, double& x1, double& x2) {
    u32x4 d;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(g) : "memory");
    const unsigned long long g0 = ((unsigned long long)d.y << 32) | d.x, g1 = ((unsigned long long)d.w << 32) | d.z;
    const unsigned long long M40 = (1ull << 40) - 1ull;
    const long long p1 = (long long)(((g0 & M40) | ((g1 & 0xFull) << 40)) << 20) >> 20;       // sign-extend 44 bits
    const long long p2 = (long long)(((g1 & M40) >> 4) << 29) >> 29;                           // sign-extend 35 bits
    x1 = (double)p1; x2 = (double)p2;                                                          // exact (< 2^53)
    return (unsigned)(g0 >> 40) == tag24 && (unsigned)(g1 >> 40) == tag24;
}

// ... four of them per lane in one round trip (a row of up to 256 workgroups' partial sums read by ONE wavefront): pairs lane,
// 64 + lane, 128 + lane, 192 + lane of the row at `row`; a pair that has arrived (tags match) is added to x1 / x2 once (got).
__device__ __forceinline__ void get_packed4(const unsigned long long* row, int lane, unsigned tag24, int W, double& x1, double& x2, unsigned& got) {
    const unsigned long long* g = row + 2 * lane;
    u32x4 d[4];
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(g) : "memory");
    const unsigned long long M40 = (1ull << 40) - 1ull;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long g0 = ((unsigned long long)d[k].y << 32) | d[k].x, g1 = ((unsigned long long)d[k].w << 32) | d[k].z;
        const bool ok = 64 * k + lane < W && !((got >> k) & 1u) && (unsigned)(g0 >> 40) == tag24 && (unsigned)(g1 >> 40) == tag24;
        if (ok) {
            x1 += (double)((long long)(((g0 & M40) | ((g1 & 0xFull) << 40)) << 20) >> 20);       // exact sums of integers below 2^53
            x2 += (double)((long long)(((g1 & M40) >> 4) << 29) >> 29);
            got |= 1u << k;
        }
    }
}

// Totals of the long-batch kernel that crosses stops: the walk patches the sums of the markers behind a crossed stop, and a
// patch is exact only on the exact sum -- so the reducer publishes the sum itself instead of its rounding.  A row's sum in grid
// units (2^-44) is S = P1 2^22 + P2 (P1: the parts on the 2^-22 grid, |P1| < 2^51; P2: the rest); it travels as h = floor(S / 2^22)
// (53 bits, signed) and l = S mod 2^22 (the pair exact_split makes of it: the walk adds h 2^-22 and l 2^-44 with one rounding,
// what every other kernel does with the two totals of a marker) beside two 24-bit tags:
//   granule 0 = {h bits 0..39, tag}     granule 1 = {h bits 40..52 | l << 13, tag}
// The rows of the crossed stops' genotype products (G of stop 0 | G of stop 1 << 24, each below 2^23) use h alone.
__device__ __forceinline__ void put_total_x(unsigned long long* g, unsigned tag24, long long h, unsigned l) {
    const unsigned long long M40 = (1ull << 40) - 1ull;
    const unsigned long long g0 = ((unsigned long long)h & M40) | ((unsigned long long)tag24 << 40);
    const unsigned long long g1 = (((unsigned long long)h >> 40) & 0x1FFFull) | ((unsigned long long)(l & 0x3FFFFFu) << 13) | ((unsigned long long)tag24 << 40);
    const u32x4 d = {(unsigned)g0, (unsigned)(g0 >> 32), (unsigned)g1, (unsigned)(g1 >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ void total_x_decode(const u32x4 d, long long& h, int& l) {
    const unsigned long long g0 = ((unsigned long long)d.y << 32) | d.x, g1 = ((unsigned long long)d.w << 32) | d.z;
    const unsigned long long M40 = (1ull << 40) - 1ull;
    h = (long long)(((g0 & M40) | ((g1 & 0x1FFFull) << 40)) << 11) >> 11;                       // sign-extend 53 bits
    l = (int)((g1 >> 13) & 0x3FFFFFull);
}

// lane l of a wavefront: values l, 64+l, 128+l, 192+l of one generation's totals in a single round trip
__device__ __forceinline__ void get_row4(const unsigned long long* base, int lane, u32x4 (&d)[4]) {
    const unsigned long long* g = base + 2 * lane;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(g) : "memory");
}

// the same for up to 384 values (a batch with markers behind a crossed stop, or more than 63 dirty markers)
__device__ __forceinline__ void get_row6(const unsigned long long* base, int lane, u32x4 (&d)[6]) {
    const unsigned long long* g = base + 2 * lane;
    const unsigned long long* g2 = g + 512;      // + 4096 bytes: the instruction offset is 13-bit signed
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                 "global_load_dwordx4 %1, %6, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %6, off offset:3072 sc1\n\t"
                 "global_load_dwordx4 %4, %7, off sc1\n\t"
                 "global_load_dwordx4 %5, %7, off offset:1024 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]) : "v"(g), "v"(g2) : "memory");
}

// ... and up to 512 (a batch of 240 markers in the two-value layout)
__device__ __forceinline__ void get_row8(const unsigned long long* base, int lane, u32x4 (&d)[8]) {
    const unsigned long long* g = base + 2 * lane;
    const unsigned long long* g2 = g + 512;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off sc1\n\t"
                 "global_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]) : "v"(g), "v"(g2) : "memory");
}

// A register-home tile's registers, read while OTHER slots of the wavefront may have loads in flight (the staging of a stopping
// marker's slice, after the round's tile loads have been issued).  Which slot is being loaded and which is read are different
// by construction -- loads go to slots of tiles behind the walk, reads to tiles of the current batch (the tile window) -- but
// no per-register data-flow can see that.  The read is therefore one asm statement that says so (an LDS store straight from
// the accumulator registers); the checker (tools/check_prefetch_regs.py) exempts exactly these statements and keeps flagging
// everything the COMPILER does to a register in flight (copies, spills, reads of its own).  The stores are not counted by
// hipcc: the caller's barrier (lds_barrier: lgkmcnt(0)) is the wait.
__device__ __forceinline__ void stage_tile_regs_guarded(uint32_t lds_dst, const u32x4& r) {
    // (the store takes its data straight from the accumulator registers: no copy through a VGPR that the compiler would schedule)
    asm volatile("; tile-window guarded read\n\tds_write_b128 %0, %1" : : "v"(lds_dst), "a"(r) : "memory");
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
template <int CTRL> __device__ __forceinline__ double dpp64(double x) {
    const int l = __builtin_amdgcn_update_dpp(0, (int)lo32(x), CTRL, 0xf, 0xf, false);
    const int h = __builtin_amdgcn_update_dpp(0, (int)hi32(x), CTRL, 0xf, 0xf, false);
    return mk64((unsigned)l, (unsigned)h);
}
constexpr int DPP_ROW_MIRROR = 0x140;        // lane ^ 15 within a row of 16
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // lane ^ 7
constexpr int DPP_QUAD_3210 = 0x1B;          // lane ^ 3
constexpr int DPP_QUAD_1032 = 0xB1;          // lane ^ 1

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

// Spin budget shared by every poll loop: give up after `limit` ticks of wall clock (100 MHz;
// SweepArgs::spin_ticks, 4 s by default) or when another workgroup has raised the abort word.
struct Spin {
    unsigned long long t0, limit;
    unsigned n;
    __device__ __forceinline__ void start(unsigned long long lim) { t0 = __builtin_amdgcn_s_memrealtime(); limit = lim; n = 0; }
    __device__ __forceinline__ bool expired(unsigned* abort_word) {
#ifndef GM_SCREEN
#define GM_SCREEN 1
#endif
#ifndef GM_POLL_SLEEP
#define GM_POLL_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(GM_POLL_SLEEP);
        if ((++n & 63u) != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit || ld_u32(abort_word) != 0u) {
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
// The per-group tables of the Gibbs step as one lane needs them (group g of its marker): sigmaG, and for
// every component denom, log(pi), -0.5 log(...) (capi.cpp, gmrm_sweep_launch), read in one burst at the start of
// the step through a pointer of KNOWN address space (a generic pointer compiles to flat loads, each followed
// by a wait for both memory counters in the middle of the arithmetic).
template <int K> struct LaneTab { double sg; double denom[K], logpi[K], mhl[K]; };
typedef const __attribute__((address_space(3))) double* TabLds;      // the tables live in the LDS carve (carve_for)
template <int K, class TP> __device__ __forceinline__ LaneTab<K> load_tab(TP tab, int G, int g) {
    LaneTab<K> t;
    t.sg = tab[g];
#pragma unroll
    for (int i = 0; i < K; i++) {
        t.denom[i] = tab[G + g * K + i];
        t.logpi[i] = tab[G + G * K + g * K + i];
        t.mhl[i] = tab[G + 2 * G * K + g * K + i];
    }
    return t;
}
template <int K>
__device__ __forceinline__ double decide0(double num, const LaneTab<K>& tb, double inv2sige, double (&muk)[K], double (&logl)[K]) {
    muk[0] = 0.0;
    logl[0] = tb.logpi[0];
#pragma unroll
    for (int i = 1; i < K; i++) {
        muk[i] = num / tb.denom[i];
        logl[i] = tb.logpi[i] + (tb.mhl[i] + muk[i] * num * inv2sige);
    }
    // bayes.cpp:437-445: tmp1 = sum_i exp(logl[i] - logl[0]).  Term 0 is exp(0) = 1 exactly (0.0 + 1.0 = 1.0) whenever
    // logl[0] is finite; a non-finite logl[0] (pi_0 == 0) makes every difference non-finite: |d| > 700 is false for
    // NaN, the sum is NaN and the comparison `prob <= acum` false -- reproduced by adding the NaN term itself.
    // No branch around exp_ (exp_(0) is exactly 1, exp_(NaN) is NaN): the K - 1 chains stay in one basic block
    // and the scheduler interleaves them; each alone is a serial chain of ~30 dependent f64 operations.
    bool zero_acum = false;
    const double d0 = logl[0] - logl[0];
    double tmp1 = (d0 == 0.0) ? 1.0 : d0;            // 1.0, or NaN
    double e[K];
#pragma unroll
    for (int i = 1; i < K; i++) {
        const double d = logl[i] - logl[0];
        if (fabs(d) > 700.0) zero_acum = true;
        e[i] = exp_(d);
    }
#pragma unroll
    for (int i = 1; i < K; i++) tmp1 += e[i];
    return zero_acum ? 0.0 : 1.0 / tmp1;
}
// bayes.cpp:450-477: the component search.  Only the lane that stops the walk needs it (a lane
// whose draw exceeds acum0 ends with a component > 0, i.e. it is the stopping lane):
//   for i = 0..K-1: stop at i if prob <= acum or i == K-1; else, unless some |logl[j] - logl[i+1]| > 700
//   (j > i), acum += 1 / sum_k exp(logl[k] - logl[i+1]).
// For the stopping lane `s` (wave-uniform) it is computed by the whole wavefront: the (K-1) x K
// exponentials of the search are independent, so lane i*K + k evaluates term k of step i, lane i sums
// its row in the reference's order (k = 0..K-1) and forms the increment, and lane s walks the steps.
// Same operations on the same values as the one-lane version, ~1 exp_ deep instead of K*(K-1).
__device__ __forceinline__ double readlane64(double x, int l) {   // l wave-uniform: two v_readlane_b32, no LDS crossbar
    return mk64((unsigned)__builtin_amdgcn_readlane((int)lo32(x), l), (unsigned)__builtin_amdgcn_readlane((int)hi32(x), l));
}
template <int K>
__device__ __forceinline__ void decide_rest_wave(int s, double prob, double acum0, const double (&logl)[K], int& kc, double& acum_v) {
    static_assert(K * (K - 1) <= 64, "one lane per (step, component)");
    const int lane = threadIdx.x & 63;
    double ls[K];
#pragma unroll
    for (int k = 0; k < K; k++) ls[k] = readlane64(logl[k], s);      // the stopping lane's log-likelihoods, everywhere
    const int ti = lane / K, tk = lane % K;                          // this lane's term: step ti, component tk
    const int ti1 = ti + 1 < K ? ti + 1 : K - 1;
    double lk = ls[0], lref = ls[0];
#pragma unroll
    for (int j = 1; j < K; j++) { lk = tk == j ? ls[j] : lk; lref = ti1 == j ? ls[j] : lref; }
    const double d = lk - lref;
    const double e = exp_(d);                                        // exp_(0) is exactly 1
    bool zero_inc = false;                                           // of step ti
#pragma unroll
    for (int j = 1; j < K; j++)
        if (j > ti && fabs(ls[j] - lref) > 700.0) zero_inc = true;
    double incs[K - 1];
    if constexpr (K == 4) {
        // a step's four terms sit in one quad: ordered sum ((e0 + e1) + e2) + e3 through DPP quad broadcasts
        const double esum = ((dpp64<0x00>(e) + dpp64<0x55>(e)) + dpp64<0xAA>(e)) + dpp64<0xFF>(e);
        const double inc = zero_inc ? 0.0 : 1.0 / esum;              // x + 0.0 == x for the acum values that occur (>= 0)
#pragma unroll
        for (int i = 0; i < K - 1; i++) incs[i] = readlane64(inc, i * K);
    } else {
        // generic K: lane L < K-1 gathers row L in the reference's order (k = 0..K-1)
        const int row = lane < K - 1 ? lane : 0;
        double esum = 0.0;
#pragma unroll
        for (int k = 0; k < K; k++) esum += __shfl(e, row * K + k, 64);
        const bool zi = __shfl((int)zero_inc, row * K, 64) != 0;
        const double inc = zi ? 0.0 : 1.0 / esum;
#pragma unroll
        for (int i = 0; i < K - 1; i++) incs[i] = readlane64(inc, i);
    }
    if (lane == s) {
        double acum = acum0;
        kc = K - 1;
        bool done = false;
#pragma unroll
        for (int i = 0; i < K; i++) {
            if (!done) {
                if (prob <= acum || i == K - 1) { kc = i; done = true; }
                else acum = acum + incs[i < K - 1 ? i : K - 2];
            }
        }
        acum_v = acum;
    }
}

// What the sampling wavefront needs about a batch position; lane j holds positions j and j + 64.
// Fetched at batch start (these loads do not depend on the dots) so they are in registers when
// the totals land.
struct LaneIn {
    int m, g;
    double beta_old, mave, msig;
};
// Exchange layout of a batch of nb markers, nd of which have a missing genotype among the phenotyped individuals
// ("dirty": the per-marker flag comes from the marker statistics):
//   [2 p], [2 p + 1]                    sum a q1, sum a q2 of batch position p            (every marker)
//   [2 nb], [2 nb + 1]                  sum q1, sum q2 over all individuals = the b-sums of every clean marker
//   [2 nb + 2 + 2 r], [.. + 1]          sum b q1, sum b q2 of the r-th dirty marker of the batch
// 2 nb + 2 + 2 nd <= SW_VMAX values: a block without missing genotypes exchanges 2 per marker (batches of up to
// 120), one with missing genotypes everywhere 4 per marker (63), anything in between pays per dirty marker.
struct SampleOut {                 // global outputs, written by workgroup 0 only
    double* acum;
    double* betas_out;
    int* comp;
};
struct Totals { double t0, t1, t2, t3; };

// The Gibbs step for a whole batch, run by wavefront 0 of EVERY workgroup on identical
// inputs.  Lane j handles batch positions j and 64 + j (two passes); the walk stops at the first
// marker whose effect may change.
#ifdef GM_SWEEP_PROF
#define SSTAMP(i) do { if (lane == 0) { unsigned long long* sp_ = reinterpret_cast<unsigned long long*>(smem + L_M + 64); \
                       const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); sp_[i] += t_ - sp_[7]; sp_[7] = t_; } } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

// gm_rng.h's unit_normal / norm with the layer table x read from LDS: the table sits in global memory otherwise,
// behind an L1 that the genotype stream keeps flushing, and the stopping lane's draw (two dependent look-ups)
// waits for L2 on the round's critical path.  Same operations on the same values; the wedge and tail cases
// (a few per cent of the draws) keep using the tables in global memory.
template <class Src> __device__ __forceinline__ double unit_normal_lx(Src& s, TabLds znx) {
    for (;;) {
        int b;
        const double x01 = int_float_pair(s, b);
        const int sign = (b & 1) * 2 - 1;
        const int i = b >> 1;
        const double xi = znx[i], xi1 = znx[i + 1];
        const double x = x01 * xi;
        if (x < xi1) return x * sign;
        if (i == 0) {
            const double tail_start = znx[1];
            for (;;) {
                const double xx = unit_exponential(s) / tail_start;
                const double yy = unit_exponential(s);
                if (2.0 * yy > xx * xx) return (xx + tail_start) * sign;
            }
        }
        const double y01 = u01(s);
        const double y = GM_ZNY[i] + y01 * (GM_ZNY[i + 1] - GM_ZNY[i]);
        double y_above_ubound, y_above_lbound;
        if (xi >= 1.0) {
            y_above_ubound = (xi - xi1) * y01 - (xi - x);
            y_above_lbound = y - (GM_ZNY[i] + (xi - x) * GM_ZNY[i] * xi);
        } else {
            y_above_lbound = (xi - xi1) * y01 - (xi - x);
            y_above_ubound = y - (GM_ZNY[i] + (xi - x) * GM_ZNY[i] * xi);
        }
        if (y_above_ubound < 0.0 && (y_above_lbound < 0.0 || y < exp_(-(x * x / 2.0)))) return x * sign;
    }
}
template <class Src> __device__ __forceinline__ double norm_lx(Src& s, double mean, double sigma2, TabLds znx) {
    const double sigma = __builtin_sqrt(sigma2);
    return unit_normal_lx(s, znx) * sigma + mean;
}

// The uniform draws of a batch (bayes.cpp:435, one per marker whose group has sigmaG != 0) do not depend on the
// dots: which word of the stream a marker gets follows from the cursor the previous batch left and the sigmaG
// flags of the markers before it.  Wavefront 0 looks them up BEFORE it polls for the totals, so that the table
// read, the ballot and the read of the stream (two dependent LDS round trips) are off the critical path.
// p1 is the draw of position 64 + lane provided the walk gets past position 63.
struct Draws { double p0, p1; };
template <class TP>
__device__ __forceinline__ Draws sample_prepare(int nb, char* smem, TP tab, int g0, int g1) {
    const int lane = threadIdx.x & 63;
    const int* ctl = reinterpret_cast<const int*>(smem + L_CTL);
    const LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1), 0, nullptr};
    const int cursor = ctl[C_CURSOR];
    const bool use0 = lane < nb && tab[g0] != 0.0;
    const bool use1 = lane + 64 < nb && tab[g1] != 0.0;
    const unsigned long long um0 = __ballot(use0), um1 = __ballot(use1);
    const unsigned long long below = (1ull << lane) - 1ull;
    Draws d;
    d.p0 = unif_from_word(rs.peek(cursor + __popcll(um0 & below)));
    d.p1 = unif_from_word(rs.peek(cursor + __popcll(um0) + __popcll(um1 & below)));
    return d;
}

// ---- walking past a marker whose effect changes ("continuation") ---------------------------------------
// The residual lives on the grid 2^-44 and a marker's update adds beta_ + a * alpha_ (grid values, gm_common.h) to every
// phenotyped individual whose genotype is not missing, exactly.  The dot products of the markers behind it therefore
// change by integers (grid units):
//     sum_i a_j(i) eps_i  +=  alpha_ * G_sj + beta_ * X_j        G_sj = sum_i a_j(i) a_s(i) na_i,  X_j = sum_i a_j(i) na_i
//     sum_i eps_i         +=  alpha_ * X_s  + beta_ * nonas
// for a stopping marker s without missing genotypes among the phenotyped individuals.  G_sj comes out of phase A (the
// genotype values of s as one more plane of operand B, exchanged like a digit sum), X_j = mave_j * nonas (an integer: the
// marker statistics' numerator).  The sums are patched as 128-bit integers and handed back as two exact parts of the same
// kinds as the exchanged ones: the dot product is the nearest double of the exact sum either way.
__device__ __forceinline__ __int128 exact_sum(double t0, double t1) {     // t0 on the 2^-22 grid (< 2^31), t1 on the 2^-44 grid: grid units
    return (__int128)((unsigned __int128)(__int128)(long long)(t0 * 0x1p22) << 22) + (__int128)(long long)(t1 * GRID_INV);
}
// ... and back: the exact sum s (grid units) as two doubles of the same kinds (t0 a multiple of 2^-22 below 2^31, 0 <= t1 < 2^-22 on
// the grid), so that the patched sums go through the same f64 expressions as exchanged ones (one rounding of the exact sum)
__device__ __forceinline__ void exact_split(__int128 s, double& t0, double& t1) {
    const long long h = (long long)(s >> 22);                             // floor
    const long long l = (long long)(s - ((__int128)h << 22));             // 0 .. 2^22 - 1
    t0 = (double)h * 0x1p-22;
    t1 = (double)l * GRID;
}

// The residual updates of one round, for phase C: n, then per update the batch position and the four values by ring code.
struct UpdList { int* n; int* pos; double* val; };
__device__ __forceinline__ UpdList upd_list(char* smem) {
    return UpdList{reinterpret_cast<int*>(smem + L_UPD), reinterpret_cast<int*>(smem + L_UPD) + 1, reinterpret_cast<double*>(smem + L_UPD + 16)};
}

// The state of a walk between its two pieces (below): wave-uniform except where noted.
struct Walk {
    int cursor;          // RNG words consumed so far
    int run;             // markers walked, for the batch-size estimate
    int nupd, ncross;    // residual updates recorded for phase C; how many of them the walk went past
    int from;            // first batch position the walk has not passed yet
    int ndone;
    bool stopped, planned;
    bool repeek;         // a stop consumed words of the stream: the prepared draws no longer hold
    bool screen;         // the recent runs are long: try the cheap certain bound before the exact probabilities (walk_piece)
    // a stop the walk may cross, met by the first piece and left to the second: its index among the batch's crossable
    // markers (or -1), its position, the two values of its update in grid units and its sum of genotype values
    int q, at;
    long long ai, bi, xs;
};

// One piece of the walk over a batch.  HOT = true: the piece every round runs -- positions 0..63, then 64..127, until a
// marker's effect changes; if that marker may be crossed (CONT: it is one of the batch's registered stops) the piece
// returns with w.q >= 0 instead of ending the round.  HOT = false: the continuation behind such a marker -- it patches
// the sums of the markers behind it (exact integers, see above) and walks on, crossing further registered stops itself.
// Two instantiations of the same code: the first stays the straight-line two-pass code the compiler makes of it when
// nothing can resume inside (its loop-carried state is small), the second is a general loop entered ~0.3 times per round.
template <int K, int CK, bool HOT, bool LONGB, class TP>
__device__ __forceinline__ void walk_piece(Walk& w, int nb, int mpos0, int G, char* smem, TP tab,
                                           const LaneIn& lin0, const LaneIn& lin1, Totals& tq0, Totals& tq1,
                                           const Draws draws, double sigmae, double inv2sige, double nm1, const SampleOut out, bool writer, int l_cass,
                                           int ns, int ps0, int ps1) {
    constexpr bool CONT = CK != 0;
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    const UpdList ul = upd_list(smem);
    int* s_cass = reinterpret_cast<int*>(smem + l_cass);
    LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1),
                 w.cursor, &ctl[C_RNGERR]};
    // The sums of the markers behind the crossed marker w.at, patched in place (exact integers, see above).  CK == 1: no
    // marker of the block has a missing genotype among the phenotyped individuals -- one packed G per marker behind the stop
    // (stop w.q in bits 24 q ..), X_j = mave_j * nonas, a common b-sum.  CK == 2: every marker may have missing genotypes --
    // per marker behind the stop two slots {Ga | Gab << 26, Za | Zb << 26} (sum a_j a_s, sum a_j b_s, sum miss_j a_s,
    // sum miss_j b_s over the phenotyped individuals) and per stop {A_s, B_s} = {sum a_s, sum b_s}:
    //     sum a_j eps += alpha_ Ga + beta_ Gab        sum b_j eps += alpha_ (A_s - Za) + beta_ (B_s - Zb)
    auto patch = [&]() {
        const double* s_tot = reinterpret_cast<const double*>(smem + L_TOT);
        if constexpr (CK == 1) {
            const double nonas = nm1 + 1.0;
            const __int128 cb = (__int128)w.ai * w.xs + (__int128)w.bi * (long long)nonas;
            const long long x0 = (long long)__builtin_rint(lin0.mave * nonas), x1 = (long long)__builtin_rint(lin1.mave * nonas);
            long long ex0 = 0, ex1 = 0;
            if (lane > ps0 && lane < nb) ex0 = (long long)s_tot[2 * nb + 2 + lane - ps0 - 1];
            if (lane + 64 > ps0 && lane + 64 < nb) ex1 = (long long)s_tot[2 * nb + 2 + lane + 64 - ps0 - 1];
            const long long g0 = w.q ? (ex0 >> 24) : (ex0 & 0xFFFFFFll), g1 = w.q ? (ex1 >> 24) : (ex1 & 0xFFFFFFll);
            if (lane > w.at) {
                exact_split(exact_sum(tq0.t0, tq0.t1) + ((__int128)w.ai * g0 + (__int128)w.bi * x0), tq0.t0, tq0.t1);
                exact_split(exact_sum(tq0.t2, tq0.t3) + cb, tq0.t2, tq0.t3);
            }
            if (lane + 64 > w.at) {
                exact_split(exact_sum(tq1.t0, tq1.t1) + ((__int128)w.ai * g1 + (__int128)w.bi * x1), tq1.t0, tq1.t1);
                exact_split(exact_sum(tq1.t2, tq1.t3) + cb, tq1.t2, tq1.t3);
            }
        } else {
            const int nv0 = 4 * nb + 2;
            const long long As = (long long)s_tot[nv0], Bs = (long long)s_tot[nv0 + 1];
            auto one = [&](int p, Totals& t) {
                if (p > w.at && p < nb) {
                    const long long e0 = (long long)s_tot[nv0 + 2 + 2 * (p - ps0 - 1)], e1 = (long long)s_tot[nv0 + 3 + 2 * (p - ps0 - 1)];
                    const long long ga = e0 & 0x3FFFFFFll, gab = e0 >> 26, za = e1 & 0x3FFFFFFll, zb = e1 >> 26;
                    exact_split(exact_sum(t.t0, t.t1) + ((__int128)w.ai * ga + (__int128)w.bi * gab), t.t0, t.t1);
                    exact_split(exact_sum(t.t2, t.t3) + ((__int128)w.ai * (As - za) + (__int128)w.bi * (Bs - zb)), t.t2, t.t3);
                }
            };
            one(lane, tq0);
            one(lane + 64, tq1);
        }
    };
    int cursor = w.cursor, run = w.run, nupd = w.nupd, ncross = w.ncross, from = w.from, ndone = w.ndone;
    bool stopped = false, planned = false, repeek = w.repeek;
    int part = from >> 6;
    if (!HOT) {                                      // the walk goes on behind the marker the first piece stopped at
        patch();
        ncross++;
    }
    w.q = -1;
#pragma unroll 1
    while (true) {                                                   // one copy of the code per piece (instruction cache)
        const int base = 64 * part;
        if (base >= nb) break;
        const int nbp = nb - base < 64 ? nb - base : 64;
        const int lo = from > base ? from - base : 0;                // lanes below lo have been walked
        // field-wise selects between two register-resident sets (an indexed array would live in scratch)
        LaneIn in;
        in.m = part ? lin1.m : lin0.m; in.g = part ? lin1.g : lin0.g;
        in.beta_old = part ? lin1.beta_old : lin0.beta_old; in.mave = part ? lin1.mave : lin0.mave; in.msig = part ? lin1.msig : lin0.msig;
        Totals tt;
        tt.t0 = part ? tq1.t0 : tq0.t0; tt.t1 = part ? tq1.t1 : tq0.t1; tt.t2 = part ? tq1.t2 : tq0.t2; tt.t3 = part ? tq1.t3 : tq0.t3;
        if constexpr (LONGB) {
            // positions 128.. of a long batch (two-value layout, nothing crossed; up to 240 markers: four passes).  The walk
            // gets here in a minority of the rounds: their inputs stay in the LDS meta ring and their totals where the
            // poll parked them, instead of in registers carried through every round.
            if (part >= 2) {                                         // (uniform)
                const int pp = base + lane < nb ? base + lane : nb - 1;
                const int sl = (mpos0 + pp) & (META_POS - 1);
                const int* mr_m = reinterpret_cast<const int*>(smem + L_META);
                const int* mr_g = mr_m + META_POS;
                const double* mr_beta = reinterpret_cast<const double*>(mr_g + META_POS);
                in = LaneIn{mr_m[sl], mr_g[sl], mr_beta[sl], mr_beta[META_POS + sl], mr_beta[2 * META_POS + sl]};
                const double* s_tot = reinterpret_cast<const double*>(smem + L_TOT);
                tt = Totals{s_tot[pp], 0.0, s_tot[nb], 0.0};          // (the long-batch kernel exchanges one total per marker: poll_totals<true>)
            }
        }
        const bool act = lane < nbp && lane >= lo;
        const int m = in.m, g = in.g;
        const double beta_old = in.beta_old;
        const LaneTab<K> tb = load_tab<K>(tab, G, g);
        const bool sig0 = act && (tb.sg == 0.0);                    // bayes.cpp:396-400
        const bool use = act && !sig0;
        const unsigned long long use_mask = __ballot(use);
        const int prefix = __popcll(use_mask & ((1ull << lane) - 1ull));
        const int cursor0 = cursor;
        double prob = part ? draws.p1 : draws.p0;                        // bayes.cpp:435: word cursor0 + prefix of the stream (sample_prepare)
        if (repeek || (LONGB && part >= 2)) prob = unif_from_word(rs.peek(cursor0 + prefix));

        SSTAMP(0);   // inputs, RNG peek
        int kc = 0;
        double acum_v = 1.0, muk_c = 0.0, denom_c = 1.0;
        double muk[K], logl[K];
#pragma unroll
        for (int i = 0; i < K; i++) { muk[i] = 0.0; logl[i] = 0.0; }
        double num = 0.0;
        if (use) {
            const double dpa = tt.t0 + tt.t1, dpb = tt.t2 + tt.t3;
            num = in.msig * (dpa - in.mave * dpb);                       // bayes.cpp:765
            num += beta_old * nm1;                                       // bayes.cpp:421
        }
        // Screen: in most passes every marker stays in component 0 by a wide margin (acum0 ~ 0.99, the draw uniform), and the
        // exact probability -- three f64 divisions, three exp_ chains and a fourth division per lane -- is needed only to SAY so.
        // A bound that is cheap and certain does as well: S~ = sum_i exp(d~_i) from one f64 fma per component (the
        // reciprocal of denom from v_rcp_f64) and v_exp_f32, within 1e-4 of the exact sum; if
        //     draw * (1 + 1.002 S~) <= 0.999999
        // then draw <= 1 / (1 + S) = acum0 as the exact arithmetic rounds it, whatever its last bits are.  A lane that is
        // not certain (the draw within ~0.2 % of acum0, an effect that was non-zero, |d| near the 700 cut-off, anything not
        // finite) sends the whole pass through the exact code below -- same chain, bit for bit, either way.
        bool screened = false;                                           // (uniform)
#if GM_SCREEN
        if (CK == 0 && w.screen) {                                       // (uniform; at high update rates nearly every pass holds a stop: skip the attempt --
                                                                         //  and the kernels that cross stops are launched for such sweeps only: not compiled in)
            bool sure = true;
            if (use) {
                const double n2 = num * num;
                float sf = 0.f;
                bool okd = beta_old == 0.0;
#pragma unroll
                for (int i = 1; i < K; i++) {
                    const double dt = (tb.logpi[i] - tb.logpi[0]) + (tb.mhl[i] + n2 * (inv2sige * __builtin_amdgcn_rcp(tb.denom[i])));
                    okd = okd && (fabs(dt) < 690.0);                     // (false for NaN)
                    sf += __builtin_amdgcn_exp2f((float)dt * 1.44269504f);
                }
                sure = okd && (prob * (1.0 + 1.002 * (double)sf) <= 0.999999);
            }
            screened = __ballot(act && !sure) == 0ull;
            if (lane == 0) { ctl[C_NSCRT]++; if (screened) ctl[C_NSCR]++; }   // (counters of the launch: plain LDS adds, nobody waits for them)
        }
#endif
        if (!screened && use) acum_v = decide0<K>(num, tb, inv2sige, muk, logl);
        SSTAMP(1);   // decide0
        // a lane whose draw exceeds acum0 ends in a component > 0 (bayes.cpp:451,476): it stops the walk
        const bool stop = !screened && use && (!(prob <= acum_v) || beta_old != 0.0);
        const unsigned long long stop_mask = __ballot(stop);
        const int s = stop_mask ? (__ffsll((long long)stop_mask) - 1) : nbp;
        const int n_done = s < nbp ? s + 1 : nbp;
        // the component search of the stopping marker (bayes.cpp:450-477), spread over the wavefront
        const bool need_search = s < nbp && __builtin_amdgcn_readlane((int)!(prob <= acum_v), s < nbp ? s : 0) != 0;   // wave-uniform
        if (need_search) {
            decide_rest_wave<K>(s, prob, acum_v, logl, kc, acum_v);
            if (lane == s) {
#pragma unroll
                for (int i = 1; i < K; i++)
                    if (i == kc) { muk_c = muk[i]; denom_c = tb.denom[i]; }
            }
        }

        SSTAMP(2);   // component search
        if (act && lane < n_done && lane != s) {
            if (sig0) {
                if (writer) out.betas_out[m] = 0.0;
            } else if (writer) {                                         // component 0, effect stays 0
                out.betas_out[m] = 0.0; out.comp[m] = 0;
                atomicAdd(&s_cass[g * K + 0], 1);
            }
        }
        if (s >= nbp) {                                                  // nobody stopped: on to the next 64 positions
            cursor = cursor0 + __popcll(use_mask);
            from = base + 64;
            part++;
            continue;
        }
        // ---- the stopping marker (lane s)
        int upd = 0, cur_s = 0;
        double alpha_ = 0.0, beta_ = 0.0;
        if (lane == s) {
            rs.cursor = cursor0 + prefix + 1;
            double beta_new = 0.0;
            if (kc > 0) beta_new = norm_lx(rs, muk_c, sigmae / denom_c, (TabLds)reinterpret_cast<const double*>(smem + L_ZNX));   // bayes.cpp:455
            const double dbeta = beta_old - beta_new;                    // bayes.cpp:479
            if (fabs(dbeta) > 0.0) {                                     // bayes.cpp:483, phenotype.cpp:328-329,388
                upd = 1;
                // (mdb * b + a) * bs_ per genotype on the residual's grid (gm_common.h: v(a) = beta_ + a * alpha_), indexed
                // by the RING code c' (k_sweep, phase A): c' = a for a = 0, 1, 2 and 3 for a missing genotype
                update_values(dbeta, in.mave, in.msig, alpha_, beta_);
                const double v1 = beta_ + alpha_;
                double* uv = ul.val + 4 * nupd;
                uv[0] = beta_;
                uv[1] = v1;
                uv[2] = v1 + alpha_;
                uv[3] = 0.0;
                ul.pos[nupd] = base + s;
            }
            if (writer) {
                out.betas_out[m] = beta_new; out.comp[m] = kc;
                atomicAdd(&s_cass[g * K + kc], 1);
            }
            cur_s = rs.cursor;
        }
        SSTAMP(3);   // commit + stop lane
        upd = __builtin_amdgcn_readlane(upd, s);
        cursor = __builtin_amdgcn_readlane(cur_s, s);
        nupd += upd;
        from = base + s + 1;
        ndone = from;
        repeek = true;
        if (!upd) {                                                      // the draw left the effect as it was: nothing moved, walk on
            if (from >= base + nbp) part++;
            continue;
        }
        const int q = !CONT ? -1 : ((ns > 0 && base + s == ps0) ? 0 : ((ns > 1 && base + s == ps1) ? 1 : -1));
        if (CONT && __builtin_expect(q >= 0, 0)) {                       // (uniform) one of the batch's markers the walk may cross
            // |values| >= 2^10 would not fit the integers of the patch; they put the residual out of range anyway (error 4 in phase C)
            const double al = readlane64(alpha_, s), be = readlane64(beta_, s);
            if (fabs(al) < 1024.0 && fabs(be) < 1024.0) {
                w.q = q; w.at = base + s;
                w.ai = (long long)(al * GRID_INV); w.bi = (long long)(be * GRID_INV);
                w.xs = (long long)__builtin_rint(readlane64(in.mave, s) * (nm1 + 1.0));
                if (HOT) break;                                          // the continuation takes over
                patch();                                                 // (second piece) patch and walk on
                ncross++;
                w.q = -1;
                if (from >= base + nbp) part++;
                continue;
            }
        }
        stopped = true;
        run = base + s + 1;
        // a stop at a marker with a non-zero effect was planned (the batch ends there by construction,
        // see compute_publish): it says nothing about how long a batch may usefully be
        planned = __builtin_amdgcn_readlane((int)(beta_old != 0.0), s) != 0;
        break;
    }
    w.cursor = cursor; w.run = run; w.nupd = nupd; w.ncross = ncross; w.from = from; w.ndone = ndone;
    w.stopped = stopped; w.planned = planned; w.repeek = repeek;
}

template <int K, int CK, bool LONGB, class TP>
__device__ __forceinline__ void sample_batch_body(int nb, int mpos0, int bmax_, int nbf16, int G, char* smem, TP tab,
                                                  const LaneIn& lin0, const LaneIn& lin1, Totals& tq0, Totals& tq1,
                                                  const Draws draws, double sigmae, double inv2sige, double nm1, const SampleOut out, bool writer, int l_cass,
                                                  int ns, int ps0, int ps1) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
#ifdef GM_SWEEP_PROF
    if (lane == 0) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[7] = __builtin_amdgcn_s_memrealtime();
#endif
    Walk w;
    w.cursor = ctl[C_CURSOR]; w.run = 2 * nb; w.nupd = 0; w.ncross = 0; w.from = 0; w.ndone = nb;
    w.stopped = false; w.planned = false; w.repeek = false; w.q = -1; w.at = 0; w.ai = 0; w.bi = 0; w.xs = 0;
    w.screen = ctl[C_EMA] >= ctl[C_SCRMIN];                              // recent run length (1/16 marker): a pass of 64 markers has a fair chance to hold no stop
    walk_piece<K, CK, true, LONGB>(w, nb, mpos0, G, smem, tab, lin0, lin1, tq0, tq1, draws, sigmae, inv2sige, nm1, out, writer, l_cass, ns, ps0, ps1);
    if (CK != 0) {
        if (__builtin_expect(w.q >= 0, 0))                           // (uniform) the walk met a marker it may cross
            walk_piece<K, CK, false, LONGB>(w, nb, mpos0, G, smem, tab, lin0, lin1, tq0, tq1, draws, sigmae, inv2sige, nm1, out, writer, l_cass, ns, ps0, ps1);
    }
    if (lane == 0) {
        ctl[C_UPD] = w.nupd;
        ctl[C_SUPD] = w.ncross;
        ctl[C_CURSOR] = w.cursor;
        ctl[C_NDONE] = w.stopped ? w.ndone : nb;
        ctl[C_PLN] = w.planned ? 1 : 0;
    }
    if (lane == 0 && !w.planned) {                                   // next batch size from the recent run length
        const int ema = (3 * ctl[C_EMA] + 16 * w.run) / 4;          // fixed point, 1/16 marker
        ctl[C_EMA] = ema;
        const int want = nbf16 * ema / 256;
        int nxt = 16;                                                // a power of two >= want: lanes are free up to it
        while (nxt < want && nxt < bmax_) nxt *= 2;
        ctl[C_NBNEXT] = nxt > bmax_ ? bmax_ : nxt;
    }
}

// K = 4 (the reference's example mixtures) is inlined into the kernel; other K share out-of-line copies.
template <int K, int CK, bool LONGB, class TP>
__device__ __noinline__ void sample_batch(int nb, int mpos0, int bmax_, int nbf16, int G, char* smem, TP tab,
                                          const LaneIn& lin0, const LaneIn& lin1, const Totals& tot0, const Totals& tot1,
                                          double p0, double p1, double sigmae, double inv2sige, double nm1, const SampleOut out, bool writer, int l_cass,
                                          int ns, int ps0, int ps1) {
    Totals tq0 = tot0, tq1 = tot1;
    sample_batch_body<K, CK, LONGB>(nb, mpos0, bmax_, nbf16, G, smem, tab, lin0, lin1, tq0, tq1, Draws{p0, p1}, sigmae, inv2sige, nm1, out, writer, l_cass, ns, ps0, ps1);
}

// lanes 0..3 of every 16-lane row take `src` from the lane SHR places below them where that lane is in the row; the others keep
// `old` (DPP row_shr, bank 0, bound_ctrl off): after SHR = 1, 2, 3 in this order lane r of a row holds what lane 0 held in call r
template <int SHR> __device__ __forceinline__ long long row_take64(long long old, long long src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x110 + SHR, 0xf, 0x1, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(old >> 32), (int)(src >> 32), 0x110 + SHR, 0xf, 0x1, false);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// ... and for the long batches (four groups of 64 positions)
struct DirtyMasks { unsigned long long m0, m1, m2, m3; };
__device__ __forceinline__ bool dirty_at4(const DirtyMasks& d, int p) {
    const unsigned long long m = p < 64 ? d.m0 : (p < 128 ? d.m1 : (p < 192 ? d.m2 : d.m3));
    return ((m >> (p & 63)) & 1ull) != 0ull;
}
__device__ __forceinline__ int dirty_rank4(const DirtyMasks& d, int p) {
    const unsigned long long below = (1ull << (p & 63)) - 1ull;
    int r = 0;
    r += p >= 64 ? __popcll(d.m0) : __popcll(d.m0 & below);
    if (p >= 64) r += p >= 128 ? __popcll(d.m1) : __popcll(d.m1 & below);
    if (p >= 128) r += p >= 192 ? __popcll(d.m2) : __popcll(d.m2 & below);
    if (p >= 192) r += __popcll(d.m3 & below);
    return r;
}
// batch position of the u-th dirty marker (uniform; u below the number of dirty markers)
__device__ __forceinline__ int dirty_pos4(DirtyMasks d, int u) {
    int pos = 0;
    for (int i = 0; i <= u; i++) {
        if (d.m0) { pos = __ffsll((long long)d.m0) - 1; d.m0 &= d.m0 - 1ull; }
        else if (d.m1) { pos = 64 + __ffsll((long long)d.m1) - 1; d.m1 &= d.m1 - 1ull; }
        else if (d.m2) { pos = 128 + __ffsll((long long)d.m2) - 1; d.m2 &= d.m2 - 1ull; }
        else { pos = 192 + __ffsll((long long)d.m3) - 1; d.m3 &= d.m3 - 1ull; }
    }
    return pos;
}
__device__ __forceinline__ DirtyMasks dirty_trim4(DirtyMasks d, int nb) {          // keep the flags of positions below nb
    auto keep = [&](unsigned long long m, int lo) { return nb >= lo + 64 ? m : (nb > lo ? (m & ((1ull << (nb - lo)) - 1ull)) : 0ull); };
    return DirtyMasks{keep(d.m0, 0), keep(d.m1, 64), keep(d.m2, 128), keep(d.m3, 192)};
}

// ---- the walk of a long batch on four wavefronts ("parallel passes", round 4) -------------------------------
// In the two-value layout with nothing to cross (the stationary sweeps of a block without missing genotypes) a batch holds up to
// 240 markers and the walk is the largest item of a round: four passes of 64 markers, one after the other, on one wavefront
// (4.4 us of a 14 us round), while wavefronts 1-3 wait.  Pass k does not depend on pass k - 1 unless that one holds the stop --
// and then pass k is not needed at all.  So wavefront w evaluates positions 64 w .. 64 w + 63 on its own: its draws start at
// cursor + (draws of the passes before it: one per marker whose group has sigmaG != 0, known without the totals), its
// totals are two values of the reducers' row.  Nothing is written until every wavefront has said where its first stop is
// (LDS, one barrier); then the wavefronts in front of the first stop commit "component 0" for their markers, the one
// that holds it commits the markers in front of it and samples the stopping marker (bayes.cpp:450-477), the others drop
// what they computed.  Same operations on the same values as the one-wavefront walk: the chain is the same, bit for bit.
struct PassEval {
    int nbp;                       // active positions of this pass (0: the batch ends before it)
    int s;                         // first lane whose marker stops the walk (nbp: none)
    int cursor0;                   // stream position of this pass's first draw
    int ndraw;                     // draws of the whole pass (markers whose group has sigmaG != 0)
    int prefix;                    // this lane's draw within the pass
    bool sig0;                     // this lane's group has sigmaG == 0 (bayes.cpp:396-400)
    bool active;
    unsigned long long use_mask;   // the lanes that draw
};
// lo: the lanes below it have been walked already (a continuation behind a crossed stop; 0 otherwise)
template <int K, class TP>
__device__ __forceinline__ PassEval pass_eval(int nb, int base, int lo, int cursor0, const LaneIn& in, double dpa, double dpb, bool screen_on,
                                              double inv2sige, double nm1, int G, char* smem, TP tab,
                                              LaneTab<K>& tb, double& prob, double& acum_v, double (&muk)[K], double (&logl)[K]) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    PassEval ev;
    ev.nbp = nb - base < 0 ? 0 : (nb - base < 64 ? nb - base : 64);
    ev.cursor0 = cursor0;
    ev.active = lane < ev.nbp && lane >= lo;
    tb = load_tab<K>(tab, G, in.g);
    ev.sig0 = ev.active && (tb.sg == 0.0);
    const bool use = ev.active && !ev.sig0;
    const unsigned long long use_mask = __ballot(use);
    ev.use_mask = use_mask;
    ev.ndraw = __popcll(use_mask);
    ev.prefix = __popcll(use_mask & ((1ull << lane) - 1ull));
    const LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1), 0, nullptr};
    prob = unif_from_word(rs.peek(cursor0 + ev.prefix));           // bayes.cpp:435
    acum_v = 1.0;
#pragma unroll
    for (int i = 0; i < K; i++) { muk[i] = 0.0; logl[i] = 0.0; }
    double num = 0.0;
    if (use) {
        num = in.msig * (dpa - in.mave * dpb);                       // bayes.cpp:765 (dpa, dpb: the exact sums, rounded once by the reducer)
        num += in.beta_old * nm1;                                    // bayes.cpp:421
    }
    bool screened = false;                                           // (uniform) the cheap certain bound: walk_piece
#if GM_SCREEN
    if (screen_on) {
        bool sure = true;
        if (use) {
            const double n2 = num * num;
            float sf = 0.f;
            bool okd = in.beta_old == 0.0;
#pragma unroll
            for (int i = 1; i < K; i++) {
                const double dt = (tb.logpi[i] - tb.logpi[0]) + (tb.mhl[i] + n2 * (inv2sige * __builtin_amdgcn_rcp(tb.denom[i])));
                okd = okd && (fabs(dt) < 690.0);                     // (false for NaN)
                sf += __builtin_amdgcn_exp2f((float)dt * 1.44269504f);
            }
            sure = okd && (prob * (1.0 + 1.002 * (double)sf) <= 0.999999);
        }
        screened = __ballot(ev.active && !sure) == 0ull;
        if (lane == 0 && ev.nbp > 0) { atomicAdd(&ctl[C_NSCRT], 1); if (screened) atomicAdd(&ctl[C_NSCR], 1); }
    }
#endif
    if (!screened && use) acum_v = decide0<K>(num, tb, inv2sige, muk, logl);
    const bool stop = !screened && use && (!(prob <= acum_v) || in.beta_old != 0.0);
    const unsigned long long stop_mask = __ballot(stop);
    ev.s = stop_mask ? (__ffsll((long long)stop_mask) - 1) : ev.nbp;
    return ev;
}
// Commit one wavefront's pass.  upto: its lanes below `upto` stay in component 0 (the whole pass when the walk goes on behind it);
// winner: lane ev.s stops the walk -- component search, draw, residual update values, the round's control words.
template <int K>
__device__ __forceinline__ void pass_commit(const PassEval& ev, bool winner, int nb, int base, const LaneIn& in, const LaneTab<K>& tb,
                                            double prob, double acum_v, const double (&muk)[K], const double (&logl)[K],
                                            double sigmae, const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16, char* smem) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    int* s_cass = reinterpret_cast<int*>(smem + l_cass);
    const int s = ev.s;
    const int upto = winner ? s : ev.nbp;
    if (ev.active && lane < upto) {
        if (ev.sig0) {
            if (writer) out.betas_out[in.m] = 0.0;
        } else if (writer) {                                         // component 0, effect stays 0
            out.betas_out[in.m] = 0.0; out.comp[in.m] = 0;
            atomicAdd(&s_cass[in.g * K + 0], 1);
        }
    }
    if (!winner) return;                                             // (uniform)
    const UpdList ul = upd_list(smem);
    int kc = 0;
    double acum2 = acum_v, muk_c = 0.0, denom_c = 1.0;
    const bool need_search = __builtin_amdgcn_readlane((int)!(prob <= acum_v), s) != 0;   // wave-uniform
    if (need_search) {
        decide_rest_wave<K>(s, prob, acum_v, logl, kc, acum2);
        if (lane == s) {
#pragma unroll
            for (int i = 1; i < K; i++)
                if (i == kc) { muk_c = muk[i]; denom_c = tb.denom[i]; }
        }
    }
    int upd = 0, cur_s = 0;
    if (lane == s) {
        LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1),
                     ev.cursor0 + ev.prefix + 1, &ctl[C_RNGERR]};
        double beta_new = 0.0;
        if (kc > 0) beta_new = norm_lx(rs, muk_c, sigmae / denom_c, (TabLds)reinterpret_cast<const double*>(smem + L_ZNX));   // bayes.cpp:455
        const double dbeta = in.beta_old - beta_new;                 // bayes.cpp:479
        if (fabs(dbeta) > 0.0) {                                     // bayes.cpp:483, phenotype.cpp:328-329,388
            upd = 1;
            double alpha_, beta_;
            update_values(dbeta, in.mave, in.msig, alpha_, beta_);
            const double v1 = beta_ + alpha_;
            ul.val[0] = beta_; ul.val[1] = v1; ul.val[2] = v1 + alpha_; ul.val[3] = 0.0;
            ul.pos[0] = base + s;
        }
        if (writer) {
            out.betas_out[in.m] = beta_new; out.comp[in.m] = kc;
            atomicAdd(&s_cass[in.g * K + kc], 1);
        }
        cur_s = rs.cursor;
        // the round's control words.  (A draw that leaves the effect as it was -- dbeta == 0 exactly -- simply ends the round
        // behind the marker: a round boundary means nothing to the chain.)
        const bool planned = in.beta_old != 0.0;
        ctl[C_UPD] = upd;
        ctl[C_SUPD] = 0;
        ctl[C_CURSOR] = cur_s;
        ctl[C_NDONE] = base + s + 1;
        ctl[C_PLN] = (upd && planned) ? 1 : 0;
        if (!(upd && planned)) {                                     // next batch size from the recent run length (walk_piece / sample_batch_body)
            const int run = base + s + 1;
            const int ema = (3 * ctl[C_EMA] + 16 * run) / 4;
            ctl[C_EMA] = ema;
            const int want = nbf16 * ema / 256;
            int nxt = 16;
            while (nxt < want && nxt < bmax_) nxt *= 2;
            ctl[C_NBNEXT] = nxt > bmax_ ? bmax_ : nxt;
        }
    }
}
// ... and when no wavefront found a stop: the wavefront of the last pass closes the round
__device__ __forceinline__ void pass_close_no_stop(const PassEval& ev, int nb, int bmax_, int nbf16, char* smem) {
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    if ((threadIdx.x & 63) == 0) {
        ctl[C_UPD] = 0; ctl[C_SUPD] = 0; ctl[C_PLN] = 0;
        ctl[C_CURSOR] = ev.cursor0 + ev.ndraw;
        ctl[C_NDONE] = nb;
        const int ema = (3 * ctl[C_EMA] + 16 * 2 * nb) / 4;         // (a batch walked to its end counts as a run of twice its length)
        ctl[C_EMA] = ema;
        const int want = nbf16 * ema / 256;
        int nxt = 16;
        while (nxt < want && nxt < bmax_) nxt *= 2;
        ctl[C_NBNEXT] = nxt > bmax_ ? bmax_ : nxt;
    }
}
// One wavefront's part of a long batch's walk, start to end: evaluate my pass, meet the others, commit.  Called by all four
// wavefronts (the barrier inside is the workgroup's).  mine: this wavefront has positions and its totals; okw: the totals arrived.
template <int K, bool MIXED, class TP>
__device__ __forceinline__ void long_batch_body(int nb, int wave, bool mine, bool okw, int cursor_w, const LaneIn& in, const double* s_tot, const DirtyMasks& dk, int* s_res,
                                                double sigmae, double inv2sige, double nm1, int G, char* smem, TP tab,
                                                const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16) {
    const int lane = threadIdx.x & 63;
    const int* ctl = reinterpret_cast<const int*>(smem + L_CTL);
    PassEval ev{0, 0, 0, 0, 0, false, false, 0ull};
    LaneTab<K> tb{};
    double prob = 0.0, acum_v = 1.0, muk[K], logl[K];
#pragma unroll
    for (int i = 0; i < K; i++) { muk[i] = 0.0; logl[i] = 0.0; }
    if (mine) {                                                          // (uniform per wavefront)
        const int base = 64 * wave;
        const int pc = base + lane < nb ? base + lane : 0;
        // one total per marker and the common one (packed exchange); a marker with missing genotypes has a row of its own for sum b eps
        const double dpa = s_tot[pc];
        double dpb = s_tot[nb];
        if constexpr (MIXED) {
            if ((dk.m0 | dk.m1 | dk.m2 | dk.m3) != 0ull) {          // (uniform: most batches of a block with few dirty markers have none or a few)
                if (dirty_at4(dk, pc)) dpb = s_tot[nb + 1 + dirty_rank4(dk, pc)];
            }
        }
        ev = pass_eval<K>(nb, base, 0, cursor_w, in, dpa, dpb, ctl[C_EMA] >= ctl[C_SCRMIN], inv2sige, nm1, G, smem, tab, tb, prob, acum_v, muk, logl);
    }
    if (lane == 0) s_res[wave] = (mine && ev.s < ev.nbp) ? ev.s : -1;
    lds_barrier();                                                       // every pass has reported
    const int r0 = s_res[0], r1 = s_res[1], r2 = s_res[2], r3 = s_res[3];
    const int wstop = r0 >= 0 ? 0 : (r1 >= 0 ? 1 : (r2 >= 0 ? 2 : (r3 >= 0 ? 3 : 4)));   // the wavefront that holds the first stop (4: none)
    if (mine && wave <= wstop)
        pass_commit<K>(ev, wave == wstop, nb, 64 * wave, in, tb, prob, acum_v, muk, logl, sigmae, out, writer, l_cass, bmax_, nbf16, smem);
    if (okw && wstop == 4 && wave == ((nb - 1) >> 6)) pass_close_no_stop(ev, nb, bmax_, nbf16, smem);
}
template <int K, bool MIXED, class TP>
__device__ __noinline__ void long_batch_k(int nb, int wave, bool mine, bool okw, int cursor_w, const LaneIn& in, const double* s_tot, const DirtyMasks& dk, int* s_res,
                                          double sigmae, double inv2sige, double nm1, int G, char* smem, TP tab,
                                          const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16) {
    long_batch_body<K, MIXED>(nb, wave, mine, okw, cursor_w, in, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16);
}
// K = 4 (the reference's example mixtures) is inlined into the kernel; other K share out-of-line copies.
template <bool MIXED, class TP>
__device__ __forceinline__ void long_batch_other_k(int K, int nb, int wave, bool mine, bool okw, int cursor_w, const LaneIn& in, const double* s_tot, const DirtyMasks& dk, int* s_res,
                                                   double sigmae, double inv2sige, double nm1, int G, char* smem, TP tab,
                                                   const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16) {
    const LaneIn lc = in;                                                // (by address: hand over a copy, the kernel's own stays in registers)
    switch (K) {
        case 2: long_batch_k<2, MIXED>(nb, wave, mine, okw, cursor_w, lc, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16); break;
        case 3: long_batch_k<3, MIXED>(nb, wave, mine, okw, cursor_w, lc, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16); break;
        case 5: long_batch_k<5, MIXED>(nb, wave, mine, okw, cursor_w, lc, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16); break;
        case 6: long_batch_k<6, MIXED>(nb, wave, mine, okw, cursor_w, lc, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16); break;
        case 7: long_batch_k<7, MIXED>(nb, wave, mine, okw, cursor_w, lc, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16); break;
        default: long_batch_k<8, MIXED>(nb, wave, mine, okw, cursor_w, lc, s_tot, dk, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16); break;
    }
}

// ---- ... and the same walk when it may cross stops (the long-batch kernel with continuation) ---------------------------
// A marker whose effect is non-zero always stops the walk, its position is known in advance, and compute_publish has made its
// genotype values a plane of operand B: the exchange then also carries G = sum a_j a_s for every marker j behind such a stop s.
// When the first stop of the parallel evaluation IS that marker, the round does not end: its wavefront records the update and
// its values (XRec, LDS), every wavefront patches the exact sums of its markers behind the stop --
//     sum a_j eps += alpha_ G_js + beta_ X_j        sum eps += alpha_ X_s + beta_ nonas        (grid units, integers)
// (X = mave * nonas, the marker's sum of genotype values: the layout without missing genotypes) -- and the positions behind the
// stop are evaluated again, in parallel as before, from the stream position the stopping marker left.  Same operations on the
// same values as walk_piece<.., CK = 1>: the same chain, bit for bit.  The exact sums arrive as (h, l) (put_total_x).
struct XRec { long long ai, bi, xs; int at, q, cur, rem; };     // the crossed stop: update values in grid units, its X; position, which of the batch's
                                                                // registered stops, stream position behind it, draws left in its pass behind it
static_assert(sizeof(XRec) <= 64, "XRec lives in the reducer scratch (L_RED)");
template <int K>
__device__ __forceinline__ void pass_commit_x(const PassEval& ev, bool winner, int nb, int base, const LaneIn& in, const LaneTab<K>& tb,
                                              double prob, double acum_v, const double (&muk)[K], const double (&logl)[K],
                                              double sigmae, double nm1, const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16, char* smem,
                                              int nupd, int ncross, int ns, int ps0, int ps1) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    int* s_cass = reinterpret_cast<int*>(smem + l_cass);
    const int s = ev.s;
    const int upto = winner ? s : ev.nbp;
    if (ev.active && lane < upto) {
        if (ev.sig0) {
            if (writer) out.betas_out[in.m] = 0.0;
        } else if (writer) {                                         // component 0, effect stays 0
            out.betas_out[in.m] = 0.0; out.comp[in.m] = 0;
            atomicAdd(&s_cass[in.g * K + 0], 1);
        }
    }
    if (!winner) return;                                             // (uniform)
    const UpdList ul = upd_list(smem);
    int kc = 0;
    double acum2 = acum_v, muk_c = 0.0, denom_c = 1.0;
    const bool need_search = __builtin_amdgcn_readlane((int)!(prob <= acum_v), s) != 0;   // wave-uniform
    if (need_search) {
        decide_rest_wave<K>(s, prob, acum_v, logl, kc, acum2);
        if (lane == s) {
#pragma unroll
            for (int i = 1; i < K; i++)
                if (i == kc) { muk_c = muk[i]; denom_c = tb.denom[i]; }
        }
    }
    if (lane == s) {
        LdsStream rs{reinterpret_cast<const uint32_t*>(smem + L_RNG0), reinterpret_cast<const uint32_t*>(smem + L_RNG1),
                     ev.cursor0 + ev.prefix + 1, &ctl[C_RNGERR]};
        double beta_new = 0.0;
        if (kc > 0) beta_new = norm_lx(rs, muk_c, sigmae / denom_c, (TabLds)reinterpret_cast<const double*>(smem + L_ZNX));   // bayes.cpp:455
        const double dbeta = in.beta_old - beta_new;                 // bayes.cpp:479
        int upd = 0;
        double alpha_ = 0.0, beta_ = 0.0;
        if (fabs(dbeta) > 0.0) {                                     // bayes.cpp:483, phenotype.cpp:328-329,388
            upd = 1;
            update_values(dbeta, in.mave, in.msig, alpha_, beta_);
            const double v1 = beta_ + alpha_;
            double* uv = ul.val + 4 * nupd;
            uv[0] = beta_; uv[1] = v1; uv[2] = v1 + alpha_; uv[3] = 0.0;
            ul.pos[nupd] = base + s;
        }
        if (writer) {
            out.betas_out[in.m] = beta_new; out.comp[in.m] = kc;
            atomicAdd(&s_cass[in.g * K + kc], 1);
        }
        const int cur_s = rs.cursor;
        const int at = base + s;
        const int q = (ns > 0 && at == ps0) ? 0 : ((ns > 1 && at == ps1) ? 1 : -1);
        // (|values| >= 2^10 would not fit the integers of the patch; they put the residual out of range anyway: error 4 in phase C)
        if (q >= 0 && upd && at + 1 < nb && fabs(alpha_) < 1024.0 && fabs(beta_) < 1024.0) {
            XRec* xr = reinterpret_cast<XRec*>(smem + L_RED);
            xr->ai = (long long)(alpha_ * GRID_INV); xr->bi = (long long)(beta_ * GRID_INV);
            xr->xs = (long long)__builtin_rint(in.mave * (nm1 + 1.0));
            xr->at = at; xr->q = q; xr->cur = cur_s;
            xr->rem = s < 63 ? __popcll(ev.use_mask >> (s + 1)) : 0;
            ctl[C_XFLAG] = 1;
        } else {
            const bool planned = in.beta_old != 0.0;
            ctl[C_XFLAG] = 0;
            ctl[C_UPD] = nupd + upd;
            ctl[C_SUPD] = ncross;
            ctl[C_CURSOR] = cur_s;
            ctl[C_NDONE] = at + 1;
            ctl[C_PLN] = (upd && planned) ? 1 : 0;
            if (!(upd && planned)) {                                 // next batch size from the recent run length (sample_batch_body)
                const int run = at + 1;
                const int ema = (3 * ctl[C_EMA] + 16 * run) / 4;
                ctl[C_EMA] = ema;
                const int want = nbf16 * ema / 256;
                int nxt = 16;
                while (nxt < want && nxt < bmax_) nxt *= 2;
                ctl[C_NBNEXT] = nxt > bmax_ ? bmax_ : nxt;
            }
        }
    }
}
template <int K, class TP>
__device__ __forceinline__ void long_cont_body(int nb, int wave, bool mine, bool okw, int cursor_w, const LaneIn& in, const long long* s_toth, const int* s_totl,
                                               int* s_res, double sigmae, double inv2sige, double nm1, int G, char* smem, TP tab,
                                               const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16, int ns, int ps0, int ps1) {
    const int lane = threadIdx.x & 63;
    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    const int base = 64 * wave;
    const int nbp = nb - base < 0 ? 0 : (nb - base < 64 ? nb - base : 64);
    const bool screen_on = ctl[C_EMA] >= ctl[C_SCRMIN];
    __int128 sm = 0, sq = 0;                                         // this lane's marker: sum a eps; the common sum of eps (grid units)
    long long xm = 0;
    if (mine) {
        const int pc = base + lane < nb ? base + lane : 0;
        sm = ((__int128)s_toth[pc] << 22) + (__int128)s_totl[pc];
        sq = ((__int128)s_toth[nb] << 22) + (__int128)s_totl[nb];
        xm = (long long)__builtin_rint(in.mave * (nm1 + 1.0));
    }
    int from = 0, cursor0 = cursor_w, nupd = 0, ncross = 0;
#pragma unroll 1
    for (;;) {                                                       // (the same number of turns in every wavefront)
        const int lo = from > base ? from - base : 0;
        const bool act_w = mine && lo < nbp;                         // this wavefront holds positions the walk has not passed
        PassEval ev{0, 0, 0, 0, 0, false, false, 0ull};
        LaneTab<K> tb{};
        double prob = 0.0, acum_v = 1.0, muk[K], logl[K];
#pragma unroll
        for (int i = 0; i < K; i++) { muk[i] = 0.0; logl[i] = 0.0; }
        if (act_w) {
            double t0, t1, t2, t3;
            exact_split(sm, t0, t1);
            exact_split(sq, t2, t3);
            ev = pass_eval<K>(nb, base, lo, cursor0, in, t0 + t1, t2 + t3, screen_on, inv2sige, nm1, G, smem, tab, tb, prob, acum_v, muk, logl);
        }
        if (lane == 0) { s_res[wave] = (act_w && ev.s < ev.nbp) ? ev.s : -1; s_res[4 + wave] = act_w ? ev.ndraw : 0; }
        lds_barrier();                                               // every pass has reported
        const int r0 = s_res[0], r1 = s_res[1], r2 = s_res[2], r3 = s_res[3];
        const int nd1 = s_res[5], nd2 = s_res[6];                    // (read now: the next turn overwrites them)
        const int wstop = r0 >= 0 ? 0 : (r1 >= 0 ? 1 : (r2 >= 0 ? 2 : (r3 >= 0 ? 3 : 4)));
        if (act_w && wave <= wstop)
            pass_commit_x<K>(ev, wave == wstop, nb, base, in, tb, prob, acum_v, muk, logl, sigmae, nm1, out, writer, l_cass, bmax_, nbf16, smem,
                             nupd, ncross, ns, ps0, ps1);
        if (okw && wstop == 4 && wave == ((nb - 1) >> 6)) {          // nobody stopped: the wavefront of the last pass closes the round
            pass_close_no_stop(ev, nb, bmax_, nbf16, smem);
            if (lane == 0) { ctl[C_UPD] = nupd; ctl[C_SUPD] = ncross; }
        }
        if (wstop == 4) break;                                       // (uniform)
        lds_barrier();                                               // the stopping marker's wavefront has said how the round goes on
        if (!ctl[C_XFLAG]) break;                                    // (uniform)
        const XRec x = *reinterpret_cast<const XRec*>(smem + L_RED);
        const int p = base + lane;
        if (mine && p > x.at && p < nb) {
            const long long hg = s_toth[nb + 1 + p - ps0 - 1];
            const long long g = x.q ? (hg >> 24) : (hg & 0xFFFFFFll);
            sm += (__int128)x.ai * g + (__int128)x.bi * xm;
        }
        sq += (__int128)x.ai * x.xs + (__int128)x.bi * (long long)(nm1 + 1.0);
        from = x.at + 1; nupd++; ncross++;
        const int wa = x.at >> 6;
        cursor0 = x.cur;
        if (wave > wa) cursor0 += x.rem + ((wa < 1 && wave > 1) ? nd1 : 0) + ((wa < 2 && wave > 2) ? nd2 : 0);
    }
}
template <int K, class TP>
__device__ __noinline__ void long_cont_k(int nb, int wave, bool mine, bool okw, int cursor_w, const LaneIn& in, const long long* s_toth, const int* s_totl,
                                         int* s_res, double sigmae, double inv2sige, double nm1, int G, char* smem, TP tab,
                                         const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16, int ns, int ps0, int ps1) {
    long_cont_body<K>(nb, wave, mine, okw, cursor_w, in, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1);
}
template <class TP>
__device__ __forceinline__ void long_cont_other_k(int K, int nb, int wave, bool mine, bool okw, int cursor_w, const LaneIn& in, const long long* s_toth, const int* s_totl,
                                                  int* s_res, double sigmae, double inv2sige, double nm1, int G, char* smem, TP tab,
                                                  const SampleOut out, bool writer, int l_cass, int bmax_, int nbf16, int ns, int ps0, int ps1) {
    const LaneIn lc = in;                                                // (by address: hand over a copy, the kernel's own stays in registers)
    switch (K) {
        case 2: long_cont_k<2>(nb, wave, mine, okw, cursor_w, lc, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1); break;
        case 3: long_cont_k<3>(nb, wave, mine, okw, cursor_w, lc, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1); break;
        case 5: long_cont_k<5>(nb, wave, mine, okw, cursor_w, lc, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1); break;
        case 6: long_cont_k<6>(nb, wave, mine, okw, cursor_w, lc, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1); break;
        case 7: long_cont_k<7>(nb, wave, mine, okw, cursor_w, lc, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1); break;
        default: long_cont_k<8>(nb, wave, mine, okw, cursor_w, lc, s_toth, s_totl, s_res, sigmae, inv2sige, nm1, G, smem, tab, out, writer, l_cass, bmax_, nbf16, ns, ps0, ps1); break;
    }
}

// rank of batch position p among the dirty markers of the batch (dm0: positions 0..63, dm1: 64..127)
__device__ __forceinline__ int dirty_rank(unsigned long long dm0, unsigned long long dm1, int p) {
    return p < 64 ? __popcll(dm0 & ((1ull << p) - 1ull)) : __popcll(dm0) + __popcll(dm1 & ((1ull << (p - 64)) - 1ull));
}
__device__ __forceinline__ bool dirty_at(unsigned long long dm0, unsigned long long dm1, int p) {
    return ((p < 64 ? dm0 >> p : dm1 >> (p - 64)) & 1ull) != 0ull;
}

// Wavefront 0 fetches ALL totals of the generation with four 16-byte loads per lane (whole cache
// lines, one round trip per look, ~6x fewer requests to the one hot 4 KB region than per-marker
// polling), parks them in LDS and lane j picks the values of batch positions j and 64 + j.
// Returns false on timeout.
template <bool PACKED>
__device__ __forceinline__ bool poll_totals(int nb, int nv, unsigned long long dm0, unsigned long long dm1,
                                            const unsigned long long* Ttg, unsigned tag, char* smem,
                                            Totals& tot0, Totals& tot1, unsigned* abort_word, unsigned long long spin_limit, int& nlooks) {
    const int lane = threadIdx.x & 63;
    double* s_tot = reinterpret_cast<double*>(smem + L_TOT);
    Spin sp;
    sp.start(spin_limit);
    bool bad = false;
    auto look = [&](auto ng_tag) {                // NG groups of 64 values: one look = NG 16-byte loads per lane, one round trip
        constexpr int NG = decltype(ng_tag)::value;
        u32x4 d[NG];
        for (;;) {
            nlooks++;
#ifdef GM_SWEEP_PROF
            if (lane == 0) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[5]++;      // looks at the totals
#endif
            if constexpr (NG == 4) get_row4(Ttg, lane, d);
            else if constexpr (NG == 6) get_row6(Ttg, lane, d);
            else get_row8(Ttg, lane, d);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NG; k++)
                if (64 * k + lane < nv) ok &= (d[k].y == tag && d[k].w == tag);
            if (__all(ok)) break;
            if (sp.expired(abort_word)) { bad = true; break; }
        }
#pragma unroll
        for (int k = 0; k < NG; k++)
            s_tot[64 * k + lane] = __longlong_as_double((long long)(((unsigned long long)d[k].z << 32) | d[k].x));
    };
    if (nv <= 256) look(std::integral_constant<int, 4>{});            // (uniform) the usual case
    else if (nv <= 384) look(std::integral_constant<int, 6>{});
    else look(std::integral_constant<int, 8>{});
    Totals t0{0.0, 0.0, 0.0, 0.0}, t1{0.0, 0.0, 0.0, 0.0};
    if constexpr (PACKED) {                      // one total per marker: the reducer has added the two exact parts already (one rounding)
        const double sq = s_tot[nb];
        if (lane < nb) t0 = Totals{s_tot[lane], 0.0, sq, 0.0};
        if (lane + 64 < nb) t1 = Totals{s_tot[lane + 64], 0.0, sq, 0.0};
        tot0 = t0; tot1 = t1;
        return !__any(bad);
    }
    const double sq1 = s_tot[2 * nb], sq2 = s_tot[2 * nb + 1];
    if ((dm0 | dm1) == 0ull) {                   // (uniform) the usual case: no dirty marker in the batch
        if (lane < nb) t0 = Totals{s_tot[2 * lane], s_tot[2 * lane + 1], sq1, sq2};
        if (lane + 64 < nb) t1 = Totals{s_tot[2 * lane + 128], s_tot[2 * lane + 129], sq1, sq2};
    } else if (__popcll(dm0) + __popcll(dm1) == nb) {   // (uniform) every marker dirty: the r-th dirty marker is position r
        if (lane < nb) t0 = Totals{s_tot[2 * lane], s_tot[2 * lane + 1], s_tot[2 * nb + 2 + 2 * lane], s_tot[2 * nb + 3 + 2 * lane]};
        if (lane + 64 < nb) t1 = Totals{s_tot[2 * lane + 128], s_tot[2 * lane + 129], s_tot[2 * nb + 130 + 2 * lane], s_tot[2 * nb + 131 + 2 * lane]};
    } else {
        if (lane < nb) {
            const int zs = dirty_at(dm0, dm1, lane) ? 2 * nb + 2 + 2 * dirty_rank(dm0, dm1, lane) : 2 * nb;
            t0 = Totals{s_tot[2 * lane], s_tot[2 * lane + 1], s_tot[zs], s_tot[zs + 1]};
        }
        if (lane + 64 < nb) {
            const int p = lane + 64;
            const int zs = dirty_at(dm0, dm1, p) ? 2 * nb + 2 + 2 * dirty_rank(dm0, dm1, p) : 2 * nb;
            t1 = Totals{s_tot[2 * p], s_tot[2 * p + 1], s_tot[zs], s_tot[zs + 1]};
        }
    }
    tot0 = t0; tot1 = t1;
    return !__any(bad);
}

// The same for the exact totals of the long-batch kernel that crosses stops (put_total_x): h parked where the other kernels park
// their doubles (L_TOT), l in the first half of L_SUM (unused in the long-batch kernels: they publish from the tile pass).
__device__ __forceinline__ bool poll_totals_x(int nv, const unsigned long long* Ttg, unsigned tag24, char* smem, unsigned* abort_word, unsigned long long spin_limit) {
    const int lane = threadIdx.x & 63;
    long long* s_toth = reinterpret_cast<long long*>(smem + L_TOT);
    int* s_totl = reinterpret_cast<int*>(smem + L_SUM);
    Spin sp;
    sp.start(spin_limit);
    bool bad = false;
    auto look = [&](auto ng_tag) {
        constexpr int NG = decltype(ng_tag)::value;
        u32x4 d[NG];
        for (;;) {
#ifdef GM_SWEEP_PROF
            if (lane == 0) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[5]++;      // looks at the totals
#endif
            if constexpr (NG == 4) get_row4(Ttg, lane, d);
            else if constexpr (NG == 6) get_row6(Ttg, lane, d);
            else get_row8(Ttg, lane, d);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NG; k++)
                if (64 * k + lane < nv) ok &= ((d[k].y >> 8) == tag24 && (d[k].w >> 8) == tag24);
            if (__all(ok)) break;
            if (sp.expired(abort_word)) { bad = true; break; }
        }
#pragma unroll
        for (int k = 0; k < NG; k++) {
            long long h; int l;
            total_x_decode(d[k], h, l);
            s_toth[64 * k + lane] = h;
            s_totl[64 * k + lane] = l;
        }
    };
    if (nv <= 256) look(std::integral_constant<int, 4>{});            // (uniform) nothing crossed, or a short batch
    else if (nv <= 384) look(std::integral_constant<int, 6>{});
    else look(std::integral_constant<int, 8>{});
    return !__any(bad);
}

// Diagnostic build only (-DGM_SWEEP_PROF): thread 0 of every workgroup accumulates wall-clock
// ticks (100 MHz) per phase; workgroups 0 and W/2 write them to stats[4..]/stats[12..].
#ifdef GM_SWEEP_PROF
#define TRACE(k) do { if (tid == 0 && a.trace && n_batch >= 2000 && n_batch < 2064) \
    a.trace[((size_t)wg * 64 + (size_t)(n_batch - 2000)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRACE(k) do { } while (0)
#endif
#ifdef GM_SWEEP_PROF
#ifndef GM_PROF_TID
#define GM_PROF_TID 0
#endif
#define PROF(i) do { if (tid == GM_PROF_TID) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
                                     prof[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PROF(i) do { } while (0)
#endif

// ---- phase A building blocks --------------------------------------------------------------
// Phase A is an int8 contraction on the matrix cores: out[marker][plane] = sum_i a_i(marker) * digit_i[plane]
// with v_mfma_i32_16x16x64_i8 (16 markers x 16 columns x 64 individuals per instruction; columns 0..7 are
// the eight digit planes).  Operand A must be int8 per individual; expanding 2-bit codes is the cost, so
// the ring holds RECODED genotypes c' whose 2-bit field IS the reference's dotp_lut_a value:
//     .bed code 00 (a=2,b=1) -> 10    01 (missing, a=b=0) -> 11    10 (a=1,b=1) -> 01    11 (a=0,b=1) -> 00
// and one v_and per register isolates field i of the four bytes of a dword (value a * 4^i; the factor is
// divided out of that field's own accumulator afterwards; the top field is shifted down first).  A lane
// (marker m = lane & 15, kg = lane >> 4) reads ONE 16-byte chunk (64 individuals) per super-step and feeds
// four MFMAs, MFMA i taking field i of each dword.  Operand B therefore holds the digit planes in the
// matching order: position 16 i + 4 j + b of a chunk <- individual 16 j + 4 b + i (dword j, byte b, field i).
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t recode_codes(uint32_t w) {
    return (~w & 0xAAAAAAAAu) | ((w ^ (w >> 1)) & 0x55555555u);
}
// four signed base-256 digits of x (|x| <= 2^30) as the four bytes of the result
__device__ __forceinline__ uint32_t signed_digits(int x) { return ((uint32_t)x + 0x00808080u) ^ 0x00808080u; }
// byte p of z0..z3 -> one dword (digit plane p of four individuals)
__device__ __forceinline__ uint4 digit_planes(uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3) {
    const uint32_t lo01 = __builtin_amdgcn_perm(z1, z0, 0x05010400u);   // z0.b0 z1.b0 z0.b1 z1.b1
    const uint32_t hi01 = __builtin_amdgcn_perm(z1, z0, 0x07030602u);   // z0.b2 z1.b2 z0.b3 z1.b3
    const uint32_t lo23 = __builtin_amdgcn_perm(z3, z2, 0x05010400u);
    const uint32_t hi23 = __builtin_amdgcn_perm(z3, z2, 0x07030602u);
    uint4 r;
    r.x = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);               // b0 of z0 z1 z2 z3
    r.y = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);               // b1
    r.z = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);               // b2
    r.w = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);               // b3
    return r;
}
// sum of x over the four lanes of a quad (both halves of a 64-bit integer through DPP quad_perm)
__device__ __forceinline__ long long quad_sum64(long long x) {
    int lo = (int)x, hi = (int)(x >> 32);
    long long y = x + (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, hi, DPP_QUAD_1032, 0xf, 0xf, false) << 32) |
                                  (unsigned)__builtin_amdgcn_update_dpp(0, lo, DPP_QUAD_1032, 0xf, 0xf, false));
    lo = (int)y; hi = (int)(y >> 32);
    y = y + (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false) << 32) |
                        (unsigned)__builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false));
    return y;
}

// sum of x over the wavefront, in every lane
__device__ __forceinline__ long long wave_sum64(long long x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int lo = __shfl_xor((int)x, o, 64), hi = __shfl_xor((int)(x >> 32), o, 64);
        x += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
    return x;
}
// A batch of a mixed block with at most this many dirty markers (but not all of them dirty) gathers their
// missing-genotype terms thread by thread instead of running the indicator MFMAs on their tiles.
#ifndef GM_SPARSE_ZMAX
#define GM_SPARSE_ZMAX 8
#endif
constexpr int SPARSE_ZMAX = GM_SPARSE_ZMAX;

// One super-step's LDS operands: the 64 bytes of the lane's digit plane (operand B of the four MFMAs) and the lane's
// 16-byte chunk of its marker's slice (operand A before the field masks) for the NLD = 0, 1 or 2 LDS-home tiles of the
// pass (B is shared by all its tiles; a register-home tile's operand A is in the wavefront's registers already).  The reads
// are inline asm so that they can be issued a whole super-step ahead of their use: hipcc does not count them, the
// matching stage_wait does (N = LDS reads issued after this stage's; LDS returns in order).  Native vector types: "+v"
// operands must be registers.
template <int NLD> struct Stage { v4i b0, b1, b2, b3; u32x4 w[NLD ? NLD : 1]; };
template <int NLD> __device__ __forceinline__ Stage<NLD> stage_read(uint32_t baddr, uint32_t waddr0, uint32_t waddr1) {
    Stage<NLD> st;
    v4i b0, b1, b2, b3; u32x4 w0, w1;
    if constexpr (NLD == 2) {
        asm volatile("ds_read_b128 %0, %6\n\t"
                     "ds_read_b128 %1, %6 offset:16\n\t"
                     "ds_read_b128 %2, %6 offset:32\n\t"
                     "ds_read_b128 %3, %6 offset:48\n\t"
                     "ds_read_b128 %4, %7\n\t"
                     "ds_read_b128 %5, %8"
                     : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(w0), "=&v"(w1) : "v"(baddr), "v"(waddr0), "v"(waddr1) : "memory");
        st.w[0] = w0; st.w[1] = w1;
    } else if constexpr (NLD == 1) {
        asm volatile("ds_read_b128 %0, %5\n\t"
                     "ds_read_b128 %1, %5 offset:16\n\t"
                     "ds_read_b128 %2, %5 offset:32\n\t"
                     "ds_read_b128 %3, %5 offset:48\n\t"
                     "ds_read_b128 %4, %6"
                     : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(w0) : "v"(baddr), "v"(waddr0) : "memory");
        st.w[0] = w0;
    } else {
        asm volatile("ds_read_b128 %0, %4\n\t"
                     "ds_read_b128 %1, %4 offset:16\n\t"
                     "ds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48"
                     : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(baddr) : "memory");
        st.w[0] = u32x4{0u, 0u, 0u, 0u};
    }
    st.b0 = b0; st.b1 = b1; st.b2 = b2; st.b3 = b3;
    return st;
}
// wait until this stage's reads have landed; LATER = true: the 4 + NLD reads of the next stage stay in flight
template <int NLD, bool LATER> __device__ __forceinline__ Stage<NLD> stage_wait(const Stage<NLD> st) {
    v4i b0 = st.b0, b1 = st.b1, b2 = st.b2, b3 = st.b3; u32x4 w0 = st.w[0];
    Stage<NLD> r;
    if constexpr (NLD == 2) {
        u32x4 w1 = st.w[1];
        if constexpr (LATER) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1) : : "memory");
        r.w[1] = w1;
    } else if constexpr (NLD == 1) {
        if constexpr (LATER) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0) : : "memory");
    } else {
        if constexpr (LATER) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : : "memory");
    }
    r.b0 = b0; r.b1 = b1; r.b2 = b2; r.b3 = b3; r.w[0] = w0;
    return r;
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// HBM -> LDS without a register pass: every lane's 16 bytes at `gsrc` (any address per lane: a gather) land at LDS byte
// address lds_dst + 16 * lane (lds_dst wave-uniform, passed in M0).  Counted by vmcnt like any load; hipcc does not know.
// M0 is the compiler's: saved and restored inside the statement (tools/micro/glds_probe.hip: destinations up to 160 KB).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);      // (uniform by construction; the compiler does not always see it)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// MODE: which markers of the block have a missing genotype among the phenotyped individuals ("dirty"; host: the
// flags of the marker statistics) -- 0: none, 1: some (per-marker flags decide), 2: all.
// CONT: the walk may cross markers whose effect was non-zero ("continuation", above; fast layout and all-dirty layout).  A kernel
// of its own: the code it adds costs the rounds that cross nothing ~3 % (register allocation), so the host launches it only for
// sweeps in which enough markers are in the model for the crossings to pay (capi.cpp, gmrm_sweep_launch).
template <int R, int MODE, bool CONT, bool LONG>
__global__ __launch_bounds__(SW_TPB, 1) void k_sweep(const SweepArgs a) {
    static_assert(!CONT || MODE == 0 || MODE == 2, "continuation: the fast layout and the all-dirty layout");
    static_assert(!LONG || MODE == 0 || (MODE == 1 && !CONT), "long batches: no missing genotypes, or a few markers with some (their Z terms gathered: sparse_z)");
    static_assert(LONG || CONT || MODE != 0, "the fast layout without crossings is the long-batch kernel");
    constexpr int CK = !CONT ? 0 : (MODE == 0 ? 1 : 2);   // 1: no marker has a missing genotype (among the phenotyped), 2: every marker may
    constexpr bool FAST = MODE == 0;
    constexpr bool LONGB = LONG;                          // batches of up to 240 markers: the walk on four wavefronts (with CONT: it crosses stops)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GE = Geo<R>;
    constexpr int NI = 4 * R;                        // individuals per thread
    constexpr int ND = NI / 4;                       // dwords of a column slice a thread takes its genotypes from
    constexpr int SB = GE::SB, CPP = GE::CPP, SS = GE::SS, TILE_B = GE::TILE_B, GPT = GE::GPT, PPG = GE::PPG;
    // register-home tiles only where batches are long (the two-value layout with nothing to cross): the other kernels' batches
    // (<= 128 markers) fit the LDS tiles, and their short batches want a tile's slice split over the wavefronts (below), which a
    // tile in ONE wavefront's registers cannot be
    constexpr int RPER = LONGB ? GE::RPER : 0, NP = LONGB ? GE::NP : 0, NPX = NP ? NP : 1;
    constexpr int PST = GE::PSTRIDE;                 // bytes per digit plane
    constexpr int BCAP = batch_cap<LONG>();
    constexpr int NPASS = BCAP > 128 ? 4 : 2;        // groups of 64 batch positions
    const int bcap_rt = (LONG && a.bcap >= 16 && a.bcap < BCAP) ? a.bcap : BCAP;   // (schedule knob of the long-batch kernels: GMRM_BATCH_CAP)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wg = blockIdx.x;
    const int W = a.W, K = a.K, G = a.G;
    const int NLS = a.nl;                            // LDS tile slots of this launch
    const unsigned nl_magic = a.nl_magic;
    const int WIN = a.win;                           // tiles [pos >> 4, (pos >> 4) + WIN) have slots of their own

    int* ctl = reinterpret_cast<int*>(smem + L_CTL);
    uint32_t* s_rng0 = reinterpret_cast<uint32_t*>(smem + L_RNG0);
    uint32_t* s_rng1 = reinterpret_cast<uint32_t*>(smem + L_RNG1);
    int* s_cass = reinterpret_cast<int*>(smem + a.lds_cass);
    long long* s_sum = reinterpret_cast<long long*>(smem + L_SUM);
    double* s_red = reinterpret_cast<double*>(smem + L_RED);
    double* s_wsq = reinterpret_cast<double*>(smem + L_WSQ);
    int* s_ab = reinterpret_cast<int*>(smem + L_AB);
    long long* s_zsp = reinterpret_cast<long long*>(smem + L_ZSP);
    double* s_tab = reinterpret_cast<double*>(smem + a.lds_tab);
    char* planes = smem + a.lds_pln;
    char* ring = smem + a.lds_ring;
    char* stage = smem + L_TOT;                      // staged slices of register-home markers (free between the walk and the next poll; NSTAGE * SB <= SW_VMAX * 8)
    static_assert(NSTAGE * GE::SB <= SW_VMAX * 8, "staging area");
    unsigned* abort_word = a.cnt + 64;
    const unsigned long long spin_limit = a.spin_ticks;
    unsigned long long* Pg = reinterpret_cast<unsigned long long*>(a.P);
    unsigned long long* Ttg = reinterpret_cast<unsigned long long*>(a.Tt);

    for (int i = tid; i < 624; i += SW_TPB) s_rng0[i] = a.rng_state[i];
    for (int i = tid; i < G * K; i += SW_TPB) s_cass[i] = 0;
    for (int i = tid; i < G * (1 + 3 * K); i += SW_TPB) s_tab[i] = a.sigmag[i];
    for (int i = tid; i < SW_VMAX; i += SW_TPB) s_sum[i] = 0;
#ifdef GM_SWEEP_PROF
    if (tid < 8) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[tid] = 0ull;
#endif
    if (tid < 129) reinterpret_cast<double*>(smem + L_ZNX)[tid] = GM_ZNX[tid];
    if (tid == 0) {
        ctl[C_CURSOR] = *a.rng_index;
        ctl[C_RNGERR] = 0;
        ctl[C_BAD] = 0;
        ctl[C_TOTF] = 0;
        ctl[C_RANGE] = 0;
        ctl[C_SCRMIN] = a.screen_min_run16; ctl[C_NSCRT] = 0; ctl[C_NSCR] = 0;
        ctl[C_EMA] = 16 * a.batch_init / 2;
        ctl[C_NBNEXT] = a.batch_init < 16 ? 16 : (a.batch_init > bcap_rt ? bcap_rt : a.batch_init);
    }
    __syncthreads();
    block_advance(s_rng0, s_rng1, ctl, false);       // S1 = twist(S0)

    // ---- this thread's individuals -----------------------------------------------------
    // The workgroup's 4*SB individuals are numbered by POSITION in operand B's order (header of this
    // section): position = 64 chunk + 16 field + 4 dword + byte.  Thread t owns positions [t*NI, t*NI + NI):
    // one 2-bit field (o_fld) of NI consecutive bytes (from byte o_jb) of one 16-byte chunk (o_chunk), so
    // its digit-plane bytes are contiguous and its genotypes sit in ND dwords of a column slice.
    const int p0w = tid * NI;
    const int o_chunk = p0w >> 6, o_fld = (p0w >> 4) & 3, o_jb = p0w & 15;
    const size_t o_byte = (size_t)wg * SB + (size_t)o_chunk * 16 + (size_t)o_jb;   // first of the NI column bytes
    const bool valid = o_byte < a.stride;                   // stride is a multiple of 16: a chunk is inside or outside
    double eps[NI];
#pragma unroll
    for (int q = 0; q < NI; q++) eps[q] = valid ? a.eps[4 * (o_byte + (size_t)q) + (size_t)o_fld] : 0.0;
    // NA / out-of-range individuals: their residual is 0 and stays 0, so their digit planes are 0 and
    // phase A may see ANY genotype code for them; only the residual update has to skip them: the owner
    // thread forces their codes to 3 (missing, update value 0) when it reads its bytes in phase C.
    uint32_t na_or[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const uint32_t nam = valid ? *reinterpret_cast<const uint32_t*>(a.namask2 + o_byte + 4 * d) : 0u;
        na_or[d] = ~nam & (0x03030303u << (2 * o_fld));
    }
    // split the thread's residuals, park their digit planes in LDS, leave the wavefront's sum of
    // q1 / q2 in s_wsq (readers: after the next barrier)
    auto refresh_planes = [&]() __attribute__((always_inline))  {
        double sq1 = 0.0, sq2 = 0.0;
        bool big = false;
        uint32_t pl[7][ND];
#pragma unroll
        for (int g4 = 0; g4 < ND; g4++) {
            uint32_t z1[4], z2[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double x = eps[4 * g4 + j];
                big |= !(fabs(x) < EPS_ABS_LIMIT);               // the digits below need |x| < 2^8 (also catches NaN)
                double q1, q2;
                split2(x, q1, q2);
                sq1 += q1; sq2 += q2;
                z1[j] = signed_digits((int)(q1 * 0x1p22));      // exact: q1 is a multiple of 2^-22, |q1| <= 2^8
                z2[j] = signed_digits((int)(q2 * GRID_INV));    // exact: q2 is a multiple of 2^-44 (the residual's grid), |q2| <= 2^-23: three digits
            }
            const uint4 d1 = digit_planes(z1[0], z1[1], z1[2], z1[3]);
            const uint4 d2 = digit_planes(z2[0], z2[1], z2[2], z2[3]);
            pl[0][g4] = d1.x; pl[1][g4] = d1.y; pl[2][g4] = d1.z; pl[3][g4] = d1.w;
            pl[4][g4] = d2.x; pl[5][g4] = d2.y; pl[6][g4] = d2.z;
        }
#pragma unroll
        for (int n = 0; n < 7; n++) {
            char* dst = planes + n * PST + (n >> 2) * 64 + p0w;
            if constexpr (ND == 1) *reinterpret_cast<uint32_t*>(dst) = pl[n][0];
            else if constexpr (ND == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(pl[n][0], pl[n][1]);
            else *reinterpret_cast<uint4*>(dst) = make_uint4(pl[n][0], pl[n][1], pl[n][2], pl[n][3]);
        }
        if (big) ctl[C_RANGE] = 1;                               // reported as error 4; every workgroup leaves
        const double r2 = reduce2(sq1, sq2);
        if (lane == 0) s_wsq[wave * 2 + 0] = r2;
        if (lane == 32) s_wsq[wave * 2 + 1] = r2;
    };
    refresh_planes();
#ifdef GM_SWEEP_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pa[5] = {0, 0, 0, 0, 0};      // phase A: entry barrier, scan + inputs, tiles, exit barrier, publish
    unsigned long long tlast = __builtin_amdgcn_s_memrealtime();
#define PA(i) do { if (tid == GM_PROF_TID) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); pa[i] += t_ - tpa; tpa = t_; } } while (0)
    unsigned long long pt[3] = {0, 0, 0};            // "loop top -> barrier" taken apart: end of the round before (meta commit), batch bookkeeping, wait for the tile loads
    unsigned long long tpt = tlast;
#define PT(i) do { if (tid == GM_PROF_TID) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); pt[i] += t_ - tpt; tpt = t_; } } while (0)
#define PT_FROM_LAST() do { tpt = tlast; } while (0)
#else
#define PA(i) do { } while (0)
#define PT(i) do { } while (0)
#define PT_FROM_LAST() do { } while (0)
#endif

    // ---- the tile window --------------------------------------------------------------------------------
    // Tile T = order positions 16 T .. 16 T + 15.  Its home follows from T alone (Geo<R>): register tiles T & 7 = 1, 2, 3 belong
    // to wavefronts 1, 2, 3 (slot (T >> 3) % NP), the others live in LDS, slot (their index among the LDS tiles) % NLS.
    // Phase A hands tile T to wavefront T & 3 -- for the register tiles that is their owner.
    auto is_reg = [&](int T) __attribute__((always_inline)) -> bool { return RPER != 0 && (unsigned)((T & 7) - 1) < 3u; };
    auto lds_index = [&](int T) __attribute__((always_inline)) -> int { return RPER != 0 ? 5 * (T >> 3) + ((T & 7) == 0 ? 0 : (T & 7) - 3) : T; };
    auto lds_slot = [&](int T) __attribute__((always_inline)) -> int { const int li = lds_index(T); return li - NLS * (int)__umulhi((unsigned)li, nl_magic); };
    auto reg_slot = [&](int T) __attribute__((always_inline)) -> int { return NP > 1 ? (T >> 3) % (NP > 1 ? NP : 1) : 0; };
    const bool loader = wave != 0;                    // wavefront 0 polls: its loads must not queue behind tile loads
    const int mrow = lane & 15, kg = lane >> 4;       // the lane's row of a 16-marker tile / its chunk within a super-step (operand A layout)
    // Byte offset of chunk c of this workgroup's slice inside a column.  Out-of-range chunks (last workgroup: its slice sticks out
    // of the column) read the same chunk of the PREVIOUS slice: in bounds, never used (their individuals do not exist).
    auto chunk_off = [&](int c) __attribute__((always_inline)) -> size_t {
        const size_t t = (size_t)wg * SB + (size_t)c * 16;
        return t < a.stride ? t : (t >= (size_t)SB ? t - SB : (size_t)0);     // (a column shorter than one slice: workgroup 0, chunk 0)
    };
    u32x4 rt[NPX][SS];                                // register-home tiles of this wavefront (AGPRs; the inline asm below owns them)
#pragma unroll
    for (int k = 0; k < NPX; k++)
#pragma unroll
        for (int s2 = 0; s2 < SS; s2++) rt[k][s2] = u32x4{0u, 0u, 0u, 0u};
    const int ntiles = (a.M + 15) >> 4;
    int pos = 0;
    int t_hi = 0;                                     // tiles [pos >> 4, t_hi) are on chip or on their way (uniform)

    // one LDS tile: GPT wave-instructions of 1 KiB; lane l of instruction i: chunk slot l % CPP of position i * PPG + l / CPP,
    // the XOR swizzle applied on the source side
    auto load_lds_tile = [&](int T, int idl) __attribute__((always_inline)) {        // idl: lane l holds the marker id of position 16 T + (l & 15)
        const uint32_t dst0 = lds_addr(ring) + (uint32_t)lds_slot(T) * (uint32_t)TILE_B;
        const int cs = lane & (CPP - 1), pin = lane / CPP;
        // The marker ids of an instruction's PPG positions are uniform: v_readlane + select instead of a cross-lane LDS read
        // per instruction (a dependent ~100-cycle round trip in front of every load: the issue of a round's ~20 loads took
        // 3.5 us, and the loads then landed too late for the next round's top)
#pragma unroll
        for (int i = 0; i < GPT; i++) {
            const int pi = i * PPG + pin;             // position within the tile
            unsigned long long colb = 0ull;
#pragma unroll
            for (int h = 0; h < PPG; h++) {
                const int sid = __builtin_amdgcn_readlane(idl, i * PPG + h);
                const unsigned long long cb = (unsigned long long)(uintptr_t)a.bed + (unsigned long long)(unsigned)sid * (unsigned long long)a.stride;
                colb = (PPG == 1 || pin == h) ? cb : colb;
            }
            const int chunk = cs ^ ((16 * T + pi) & (CPP - 1));
            const uint8_t* src = reinterpret_cast<const uint8_t*>((uintptr_t)colb) + chunk_off(chunk);
            glds16(src, dst0 + (uint32_t)i * 1024u);
        }
    };
    // one register tile: lane (mrow, kg) takes chunk 4 s + kg of marker 16 T + mrow, s = 0 .. SS - 1
    auto load_reg_tile = [&](auto k_tag, int idl) __attribute__((always_inline))  {
        constexpr int KR = decltype(k_tag)::value;
        const uint8_t* col = a.bed + (size_t)idl * a.stride;
#pragma unroll
        for (int s2 = 0; s2 < SS; s2++) {
            const uint8_t* src = col + chunk_off(4 * s2 + kg);
            u32x4& dst = rt[KR][s2];                  // (named outside the asm: a generic lambda captures it only then)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+a"(dst) : "v"(src) : "memory");
        }
    };
    // ---- per-marker inputs of the sampling step (marker id, group, previous effect, mave, msig):
    // wavefronts 1-3 fetch them for upcoming order positions (dependent global loads) one round ahead -- two groups of 64
    // positions each, up to 384 per round -- and park them in a 256-position LDS ring, so a batch never waits on them.
    int* mr_m = reinterpret_cast<int*>(smem + L_META);
    int* mr_g = mr_m + META_POS;
    double* mr_beta = reinterpret_cast<double*>(mr_g + META_POS);
    double* mr_mave = mr_beta + META_POS;
    double* mr_msig = mr_mave + META_POS;
    uint8_t* mr_nm = reinterpret_cast<uint8_t*>(smem + L_NM);
    int mhi = 0, npm = 0;                             // meta ring holds positions [pos, mhi)
    int pm_m[2] = {0, 0}, pm_g[2] = {0, 0}, pm_nm[2] = {1, 1};
    double pm_beta[2] = {0.0, 0.0}, pm_mave[2] = {0.0, 0.0}, pm_msig[2] = {1.0, 1.0};
    // One step (round 4): the inputs sit in order-major arrays (ops.hip, k_order_inputs), so a group of 64 positions is five
    // contiguous loads -- no ids first, no gathers of 64 random addresses (each another page for the address pipeline the
    // genotype loads go through).  In FRONT of the tile loads (a wait for these behind those would be a wait for the columns).
    auto meta_request = [&](int want) __attribute__((always_inline)) {
        if (want > a.M) want = a.M;
        npm = want - mhi;
        if (npm > 384) npm = 384;
        if (npm < 0) npm = 0;
        if (loader) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int grp = (wave - 1) + 3 * h;       // this wavefront's groups of 64 positions: wave - 1 and wave + 2
                if (64 * grp < npm) {
                    const int p = mhi + 64 * grp + lane;
                    const int pi = p < a.M ? p : a.M - 1;
                    pm_m[h] = a.order[pi];
                    pm_g[h] = a.o_g[pi];
                    pm_beta[h] = a.o_beta[pi];
                    pm_mave[h] = a.o_mave[pi];
                    pm_msig[h] = a.o_msig[pi];
                    if (MODE == 1) pm_nm[h] = a.o_nm[pi];
                }
            }
        }
    };
    auto meta_commit = [&]() __attribute__((always_inline))  {
        int nc = pos + META_POS - mhi;
        if (nc > npm) nc = npm;
        if (nc < 0) nc = 0;
        if (loader) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int grp = (wave - 1) + 3 * h;
                if (64 * grp + lane < nc) {
                    const int sl = (mhi + 64 * grp + lane) & (META_POS - 1);
                    mr_m[sl] = pm_m[h]; mr_g[sl] = pm_g[h]; mr_beta[sl] = pm_beta[h]; mr_mave[sl] = pm_mave[h]; mr_msig[sl] = pm_msig[h];
                    if (MODE == 1) mr_nm[sl] = (uint8_t)pm_nm[h];
                }
            }
        }
        mhi += nc;
        npm = 0;
    };
    auto ensure_meta = [&](int upto) __attribute__((always_inline))  {                 // slow path (uniform): meta ring must hold [.., upto)
        if (mhi < upto) meta_commit();
        while (mhi < upto) { meta_request(upto); meta_commit(); }
    };

    // Request tiles [t_hi, t_lim) (at most 16 per call).  LDS tiles are shared out over the loader wavefronts by their index,
    // a register tile is loaded by its owner: among 16 consecutive tiles a wavefront owns at most two, in different slots, so
    // the loads of every slot are ONE asm site in the whole kernel (several sites per slot made the register allocator keep a
    // slot in different registers at different sites, with copies of registers whose loads were in flight in between:
    // tools/check_prefetch_regs.py).  All marker ids are requested before the first tile load; the tile loads themselves
    // wait until wavefront 0 has the totals (gate_tag != 0): every memory instruction of a compute unit goes through one
    // in-order address pipeline, and loads issued while wavefront 0 polls delay its looks.
    constexpr int JMAXL = RPER ? 4 : 6;               // LDS tiles one wavefront loads per call, at most
    // The marker ids of the tiles this wavefront may load next -- its register tiles (by slot) and its LDS tiles among
    // [t_hi, t_hi + 16) -- are requested AHEAD (at the top of the round): fetched inside tile_issue they put two dependent L2
    // round trips (ids, then the loads that hang on them) in front of the loaders' own pass of the walk, and wavefront 0 waited
    // for them at the walk's barrier (4.6 us of "issue" per round).
    int idq_for = -1;                                 // the ids below were requested for t_hi == idq_for (uniform)
    int idr0 = 0, idr1 = 0, idr2 = 0;                 // lane l: the marker at row l & 15 of my register tile in slot 0 / 1 / 2
    int idl[JMAXL];                                   // ... of my j-th LDS tile
#pragma unroll
    for (int j = 0; j < JMAXL; j++) idl[j] = 0;
    // (by value, and every slot's tile by its own select: out-parameters assigned in an if / else chain became ONE store through a
    //  selected address -- three words of scratch, and every scratch load is followed by s_waitcnt vmcnt(0): the wavefront waited
    //  for its id loads at the top of the round and for its tile loads in the middle of issuing them)
    struct MyTiles { int tk0, tk1, tk2; unsigned lmine; };
    auto my_tiles = [&](int t0, int t1) __attribute__((always_inline)) -> MyTiles {   // among [t0, t1) (uniform)
        MyTiles r{-1, -1, -1, 0u};
        if constexpr (NP > 0) {
            for (int T = t0 + ((wave - t0) & 7); T < t1; T += 8) {   // T & 7 == wave (1..3): a register tile of mine
                const int k = reg_slot(T);
                r.tk0 = (k == 0 && r.tk0 < 0) ? T : r.tk0;           // (the first per slot: with one slot a wavefront's next two tiles share it)
                r.tk1 = (k == 1 && r.tk1 < 0) ? T : r.tk1;
                r.tk2 = (k == 2 && r.tk2 < 0) ? T : r.tk2;
            }
        }
        for (int T = t0; T < t1; T++) {
            const bool m_ = !is_reg(T) && (1 + lds_index(T) % 3) == wave;
            r.lmine |= (m_ ? 1u : 0u) << (T - t0);
        }
        return r;
    };
    auto id_of = [&](int T) __attribute__((always_inline)) -> int {     // lane l: the marker at position 16 T + (l & 15)
        const int p = 16 * T + mrow;
        return a.order[p < a.M ? p : a.M - 1];
    };
    auto tile_ids = [&]() __attribute__((always_inline)) {
        if (loader) {
            const int t0 = __builtin_amdgcn_readfirstlane(t_hi);
            int t1 = t0 + 16;
            if (t1 > ntiles) t1 = ntiles;
            const MyTiles mt = my_tiles(t0, t1);
            const int tk0 = mt.tk0, tk1 = mt.tk1, tk2 = mt.tk2;
            const unsigned lmine = mt.lmine;
            // (unconditional, with a harmless tile where there is none: a load merged with the old value of its register would
            //  have to be waited for on the spot)
            if (NP > 0) idr0 = id_of(tk0 >= 0 ? tk0 : t0);
            if (NP > 1) idr1 = id_of(tk1 >= 0 ? tk1 : t0);
            if (NP > 2) idr2 = id_of(tk2 >= 0 ? tk2 : t0);
            unsigned mm = lmine;
#pragma unroll
            for (int j = 0; j < JMAXL; j++) {
                const int T = __builtin_amdgcn_readfirstlane(mm ? t0 + (__ffs((int)mm) - 1) : t0);
                mm &= mm - 1u;
                idl[j] = id_of(T);
            }
        }
        idq_for = t_hi;
    };
    auto tile_issue = [&](int t_lim, int meta_want) __attribute__((always_inline)) {
        if (t_lim > t_hi + 16) t_lim = t_hi + 16;
        if (t_lim < t_hi) t_lim = t_hi;
        if (idq_for != t_hi) tile_ids();              // (uniform: the ids of the tiles that may be loaded now)
        if (meta_want > 0) meta_request(meta_want);   // (every wavefront keeps the ring's bookkeeping; the loaders load)
        if (loader) {
            t_hi = __builtin_amdgcn_readfirstlane(t_hi);
            t_lim = __builtin_amdgcn_readfirstlane(t_lim);
            const MyTiles mt = my_tiles(t_hi, t_lim); // a prefix of what tile_ids saw: the same tiles, in the same order
            const int tk0 = mt.tk0, tk1 = mt.tk1, tk2 = mt.tk2;
            const unsigned lmine = mt.lmine;
            if constexpr (NP > 0) { if (tk0 >= 0) load_reg_tile(std::integral_constant<int, 0>{}, idr0); }
            if constexpr (NP > 1) { if (tk1 >= 0) load_reg_tile(std::integral_constant<int, 1>{}, idr1); }
            if constexpr (NP > 2) { if (tk2 >= 0) load_reg_tile(std::integral_constant<int, 2>{}, idr2); }
            PROF(3);   // (diagnostic build, loader wavefront: meta loads + register-tile loads; the LDS tiles' loads count as "requests")
            unsigned mm = lmine;
#pragma unroll
            for (int j = 0; j < JMAXL; j++) {
                if (mm) {                             // (uniform)
                    const int T = __builtin_amdgcn_readfirstlane(t_hi + (__ffs((int)mm) - 1));
                    mm &= mm - 1u;
                    load_lds_tile(T, idl[j]);
                }
            }
        }
        t_hi = t_lim;
    };
    // every tile load issued so far has landed (each wavefront waits for its own; visible to the others after the next barrier)
    auto tiles_wait = [&]() __attribute__((always_inline))  {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NPX; k++)
#pragma unroll
            for (int s2 = 0; s2 < SS; s2++) asm volatile("" : "+a"(rt[k][s2]));   // the values are defined HERE, after the wait: no use can be scheduled earlier
    };
    // The slice of order position p, as the threads read their own genotypes (residual update, stop planes): in its LDS tile,
    // or -- a register-home marker -- in staging slot `slot`, where its owner has put it (stage_slices).  `swz`: the XOR of the
    // chunk index (LDS tiles are swizzled by position, staged slices are not).
    struct SliceAt { const char* base; int swz; };
    auto slice_at = [&](int p, int slot) __attribute__((always_inline)) -> SliceAt {
        const int T = p >> 4;
        if (is_reg(T)) return SliceAt{stage + slot * SB, 0};
        return SliceAt{ring + (size_t)lds_slot(T) * TILE_B + (size_t)(p & 15) * SB, p & (CPP - 1)};
    };
    // Owners copy the slices of the register-home markers among n order positions (at(u), uniform) to the staging slots;
    // returns whether there was any (uniform): the caller then needs a barrier before anybody reads them.
    auto stage_slices = [&](int n, auto at) __attribute__((always_inline)) -> bool {
        bool any = false;
        if constexpr (NP > 0) {
#pragma unroll 1
            for (int u = 0; u < n; u++) {
                const int p = at(u);
                const int T = p >> 4;
                if (!is_reg(T)) continue;             // (uniform)
                any = true;
                if ((T & 7) != wave) continue;
                const int k = reg_slot(T);
                if (mrow == (p & 15)) {               // the four lanes that hold this marker's chunks
                    u32x4* d = reinterpret_cast<u32x4*>(stage + u * SB + 16 * kg);
                    if (k == 0) {
#pragma unroll
                        for (int s2 = 0; s2 < SS; s2++) stage_tile_regs_guarded(lds_addr(d + 4 * s2), rt[0][s2]);   // (opaque per slot: the three arms must not be merged into one load through a selected address -- rt would then live in scratch)
                    }
                    if constexpr (NP > 1) {
                        if (k == 1) {
#pragma unroll
                            for (int s2 = 0; s2 < SS; s2++) stage_tile_regs_guarded(lds_addr(d + 4 * s2), rt[1][s2]);   // (opaque per slot: the three arms must not be merged into one load through a selected address -- rt would then live in scratch)
                        }
                    }
                    if constexpr (NP > 2) {
                        if (k == 2) {
#pragma unroll
                            for (int s2 = 0; s2 < SS; s2++) stage_tile_regs_guarded(lds_addr(d + 4 * s2), rt[2][s2]);   // (opaque per slot: the three arms must not be merged into one load through a selected address -- rt would then live in scratch)
                        }
                    }
                }
            }
        }
        return any;
    };

    // ---- the marker loop ---------------------------------------------------------------------------------
    // Generation g uses tag g+1 and buffer g&1; a buffer is rewritten only after every
    // workgroup has sampled the generation that used it.
    struct Batch { int p0, nb, nv; unsigned gen; bool planned; unsigned long long dm0, dm1; int ns, ps0, ps1; int nr; unsigned long long dm2, dm3; };   // nv: totals the walk waits for; nr: rows of partial sums the reducers sum   // planned: ends at a marker known to stop the walk; dm: dirty positions;
                                                                                                                // ns, ps: the markers with a non-zero effect the walk may cross (batch positions)
    unsigned gen_next = 0;
    long long n_upd = 0, n_batch = 0, n_planned = 0, n_stale = 0, n_fastb = 0, n_cross = 0, n_short = 0;
    int max_nb = 0;
    bool ok = true;

    // phase A for positions [b.p0, b.p0 + b.nb) (their tiles are on chip) + publish
    auto compute_publish = [&](Batch& b, LaneIn& li0, LaneIn& li1) __attribute__((always_inline))  {
#ifdef GM_SWEEP_PROF
        unsigned long long tpa = tlast;
#endif
        lds_barrier();                                // tile / plane / meta writes are visible
        PA(0);
        const int p0 = b.p0;
        // Everything the scan needs from the meta ring is fetched in ONE burst, for ring positions that may lie behind
        // the end of the batch (the index is masked; what is not needed is dropped below): a load issued only after
        // the batch length is known would put a second and a third LDS round trip on the critical path.
        const int sl0 = (p0 + lane) & (META_POS - 1), sl1 = (p0 + lane + 64) & (META_POS - 1);
        const double rb0 = mr_beta[sl0], rb1 = mr_beta[sl1];
        double rb2 = 0.0, rb3 = 0.0;
        if (NPASS > 2) { rb2 = mr_beta[(p0 + lane + 128) & (META_POS - 1)]; rb3 = mr_beta[(p0 + lane + 192) & (META_POS - 1)]; }
        unsigned char rn0 = 1, rn1 = 1, rn2 = 1, rn3 = 1;
        if (MODE == 1) { rn0 = mr_nm[sl0]; rn1 = mr_nm[sl1]; }
        if (MODE == 1 && NPASS > 2) { rn2 = mr_nm[(p0 + lane + 128) & (META_POS - 1)]; rn3 = mr_nm[(p0 + lane + 192) & (META_POS - 1)]; }
        LaneIn r0i{0, 0, 0.0, 0.0, 1.0}, r1i{0, 0, 0.0, 0.0, 1.0};
        if constexpr (LONGB) {                        // parallel passes: wavefront w samples positions 64 w .. 64 w + 63
            const int slw = (p0 + lane + 64 * wave) & (META_POS - 1);
            r0i = LaneIn{mr_m[slw], mr_g[slw], mr_beta[slw], mr_mave[slw], mr_msig[slw]};
        } else if (wave == 0) {
            r0i = LaneIn{mr_m[sl0], mr_g[sl0], rb0, mr_mave[sl0], mr_msig[sl0]};
            r1i = LaneIn{mr_m[sl1], mr_g[sl1], rb1, mr_mave[sl1], mr_msig[sl1]};
        }
        // Dirty markers (a missing genotype among the phenotyped individuals) exchange two more values each: the batch
        // is cut where the slots run out.  (Every wavefront reads the same LDS bytes: uniform.)
        unsigned long long dm0 = 0ull, dm1 = 0ull, dm2 = 0ull, dm3 = 0ull;
        // The long-batch kernel for blocks with a FEW markers that have missing genotypes: every tile runs the one-MFMA-set pass and
        // the Z terms of a batch's dirty markers are gathered from the digit planes (sparse_z below).  At most ZCAP of them per
        // batch (staging slots for the register-home ones): the batch is cut before the next.
        constexpr int ZCAP = SPARSE_ZMAX < SW_VMAX * 8 / GE::SB ? SPARSE_ZMAX : SW_VMAX * 8 / GE::SB;
        if constexpr (LONGB && MODE == 1) {
            dm0 = __ballot(lane < b.nb && rn0 == 0); dm1 = __ballot(lane + 64 < b.nb && rn1 == 0);
            dm2 = __ballot(lane + 128 < b.nb && rn2 == 0); dm3 = __ballot(lane + 192 < b.nb && rn3 == 0);
            if (__popcll(dm0) + __popcll(dm1) + __popcll(dm2) + __popcll(dm3) > ZCAP) b.nb = dirty_pos4(DirtyMasks{dm0, dm1, dm2, dm3}, ZCAP);   // (uniform; >= 1: ZCAP >= 1)
        }
        if (MODE == 2) {                              // every marker dirty: 4 slots each, the r-th dirty marker is position r
            if (b.nb > (SW_VMAX - 2) / 4) b.nb = (SW_VMAX - 2) / 4;
            dm0 = ~0ull; dm1 = ~0ull;
        }
        if (MODE == 1 && !LONGB) {
            const bool q0 = lane < b.nb && rn0 == 0;
            const bool q1 = lane + 64 < b.nb && rn1 == 0;
            dm0 = __ballot(q0); dm1 = __ballot(q1);
            const int ndall = __popcll(dm0) + __popcll(dm1);
            if (ndall == b.nb) {                      // (uniform) every marker dirty: 4 slots each
                if (b.nb > (SW_VMAX - 2) / 4) b.nb = (SW_VMAX - 2) / 4;
            } else if (2 * b.nb + 2 + 2 * ndall > SW_VMAX) {   // (uniform) the slots run out inside the batch: find where
                // slots needed by the first n markers: 2 n + 2 + 2 (dirty among them); monotone in n
                const bool f0 = lane < b.nb && 2 * (lane + 1) + 2 + 2 * dirty_rank(dm0, dm1, lane + 1) <= SW_VMAX;
                const bool f1 = lane + 64 < b.nb && 2 * (lane + 65) + 2 + 2 * (lane + 65 <= 127 ? dirty_rank(dm0, dm1, lane + 65) : ndall) <= SW_VMAX;
                const int fit = __popcll(__ballot(f0)) + __popcll(__ballot(f1));
                if (fit < b.nb) b.nb = fit;           // fit >= 1
            }
        }
        // A marker whose effect is non-zero always changes it (bayes.cpp:479-483: the new draw differs), so the walk is
        // known to stop there.  Without continuation the batch ends at the first such marker (the dots behind it
        // are certain to go stale); with it the walk crosses up to NSTOP of them -- their genotype values become planes
        // of operand B below, so that the sums behind them can be patched exactly (sample_batch_body) -- and the batch
        // ends at the next one.  (Every wavefront scans the same LDS words: uniform.)
        {
            const bool nz0 = lane < b.nb && rb0 != 0.0;
            const bool nz1 = lane + 64 < b.nb && rb1 != 0.0;
            unsigned long long m0 = __ballot(nz0), m1 = __ballot(nz1);
            int first = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : b.nb);
            if (NPASS > 2 && first >= b.nb) {         // (uniform) positions 128.. of a long batch
                const unsigned long long m2 = __ballot(lane + 128 < b.nb && rb2 != 0.0), m3 = __ballot(lane + 192 < b.nb && rb3 != 0.0);
                first = m2 ? 128 + __ffsll((long long)m2) - 1 : (m3 ? 192 + __ffsll((long long)m3) - 1 : b.nb);
            }
            b.ns = 0; b.ps0 = 0; b.ps1 = 0;
            // A crossing costs about half a round (the sums behind the marker are patched and decided again, its genotype values
            // become a plane, more values are exchanged): it pays when a good part of the batch lies behind the marker
            const int cross_thr = (b.nb * a.cross + 15) >> 4;   // a.cross: sixteenths of the batch (0: never)
            if constexpr (LONGB && CONT) {
                // the long batch: all four groups of 64 positions are scanned; a row per marker behind the first crossed stop always
                // fits (2 nb <= 480 rows)
                unsigned long long m2 = 0ull, m3 = 0ull;
                if (NPASS > 2) { m2 = __ballot(lane + 128 < b.nb && rb2 != 0.0); m3 = __ballot(lane + 192 < b.nb && rb3 != 0.0); }
                auto pop4 = [&]() __attribute__((always_inline)) -> int {
                    if (m0) { const int f = __ffsll((long long)m0) - 1; m0 &= m0 - 1ull; return f; }
                    if (m1) { const int f = 64 + __ffsll((long long)m1) - 1; m1 &= m1 - 1ull; return f; }
                    if (m2) { const int f = 128 + __ffsll((long long)m2) - 1; m2 &= m2 - 1ull; return f; }
                    if (m3) { const int f = 192 + __ffsll((long long)m3) - 1; m3 &= m3 - 1ull; return f; }
                    return -1;
                };
                int f0 = pop4();
                if (f0 < 0) f0 = b.nb;
                if (a.cross && f0 + 1 < b.nb && f0 + cross_thr <= b.nb - 1) {   // (uniform)
                    b.ns = 1; b.ps0 = f0;
                    int last = pop4();                               // the stop that ends the batch, if any
                    if (NSTOP > 1 && last >= 0 && last + 1 < b.nb && last + cross_thr <= b.nb - 1) {
                        b.ns = 2; b.ps1 = last; last = pop4();
                    }
                    if (last >= 0 && last + 1 < b.nb) b.nb = last + 1;
                    b.planned = last >= 0 && last < b.nb;
                } else {
                    if (f0 + 1 < b.nb) b.nb = f0 + 1;
                    b.planned = f0 < b.nb;
                }
            } else
            if (CONT && a.cross && first + 1 < b.nb && first + cross_thr <= b.nb - 1) {   // (uniform)
                auto pop_first = [&]() __attribute__((always_inline)) -> int {
                    if (m0) { const int f = __ffsll((long long)m0) - 1; m0 &= m0 - 1ull; return f; }
                    if (m1) { const int f = 64 + __ffsll((long long)m1) - 1; m1 &= m1 - 1ull; return f; }
                    return -1;
                };
                b.ns = 1; b.ps0 = pop_first();
                int last = pop_first();                          // the stop that ends the batch, if any
                if (CK == 2) {                                   // 4 slots per marker + 2 for the stop + 2 per marker behind it
                    const int fit = (SW_VMAX - 2 + 2 * b.ps0) / 6;
                    if (b.nb > fit) b.nb = fit;
                }
                if (CK == 1) {                                   // 2 slots per marker + 2 + 1 per marker behind the first stop
                    const int fit = (SW_VMAX - 2 + b.ps0 + 1) / 3;
                    if (b.nb > fit) b.nb = fit;
                }
                if (CK == 1 && NSTOP > 1 && last >= 0 && last + 1 < b.nb && last + cross_thr <= b.nb - 1) {
                    b.ns = 2; b.ps1 = last; last = pop_first();
                }
                if (last >= 0 && last + 1 < b.nb) b.nb = last + 1;
                b.planned = last >= 0 && last < b.nb;
                if (b.ps0 + 1 >= b.nb) { b.ns = 0; b.planned = true; }   // (the slots left nobody behind it: the batch ends at the marker)
            } else {
                if (first + 1 < b.nb) b.nb = first + 1;
                b.planned = first < b.nb;                        // the last marker of the batch has a non-zero effect
            }
        }
        const int nb = b.nb;
        if (!FAST) {                                  // keep the flags of the batch's positions only
            dm0 = nb >= 64 ? dm0 : (dm0 & ((1ull << nb) - 1ull));
            dm1 = nb >= 128 ? dm1 : (nb > 64 ? (dm1 & ((1ull << (nb - 64)) - 1ull)) : 0ull);
        }
        if constexpr (LONGB && MODE == 1) {
            const DirtyMasks t = dirty_trim4(DirtyMasks{dm0, dm1, dm2, dm3}, nb);
            dm0 = t.m0; dm1 = t.m1; dm2 = t.m2; dm3 = t.m3;
        }
        b.dm0 = dm0; b.dm1 = dm1; b.dm2 = dm2; b.dm3 = dm3;
        const DirtyMasks dmk{dm0, dm1, dm2, dm3};
        const int nd = __popcll(dm0) + __popcll(dm1) + __popcll(dm2) + __popcll(dm3);
        const bool all_dirty = MODE == 2 || (!LONGB && nd == nb); // then the r-th dirty marker is batch position r
        // A few dirty markers among clean ones: every tile runs the one-MFMA-set pass (X = sum c d) and the
        // missing-genotype term Z of each dirty marker is gathered from the digit planes, one wavefront per
        // marker (sparse_z below), instead of a second MFMA set over every tile that holds such a marker.  Only where every
        // dirty marker of the batch lives in an LDS tile (a register tile's slice is not addressable lane by lane).
        bool sparse_z = MODE == 1 && !all_dirty && nd > 0 && nd <= SPARSE_ZMAX;      // uniform (the long-batch kernel: always, nd <= ZCAP by the cut above)
        if constexpr (MODE == 1 && RPER != 0 && !LONGB) {
            if (sparse_z) {
                unsigned long long r0 = dm0, r1 = dm1;
                while (r0 | r1) {
                    int m;
                    if (r0) { m = __ffsll((long long)r0) - 1; r0 &= r0 - 1ull; }
                    else    { m = 64 + __ffsll((long long)r1) - 1; r1 &= r1 - 1ull; }
                    if (is_reg((p0 + m) >> 4)) sparse_z = false;
                }
            }
        }
        if constexpr (LONGB) {
            const bool a0 = lane + 64 * wave < nb;
            li0 = LaneIn{a0 ? r0i.m : 0, a0 ? r0i.g : 0, a0 ? r0i.beta_old : 0.0, a0 ? r0i.mave : 0.0, a0 ? r0i.msig : 1.0};
        } else {
            const bool a0 = wave == 0 && lane < nb, a1 = wave == 0 && lane + 64 < nb;
            li0 = LaneIn{a0 ? r0i.m : 0, a0 ? r0i.g : 0, a0 ? r0i.beta_old : 0.0, a0 ? r0i.mave : 0.0, a0 ? r0i.msig : 1.0};
            li1 = LaneIn{a1 ? r1i.m : 0, a1 ? r1i.g : 0, a1 ? r1i.beta_old : 0.0, a1 ? r1i.mave : 0.0, a1 ? r1i.msig : 1.0};
        }
        max_nb = nb > max_nb ? nb : max_nb;
        PA(1);
        if constexpr (LONGB && MODE == 1) {
            if (nd > 0) {                             // (uniform) the long-batch kernel: the slices of register-home dirty markers are staged by their owners first
                if (stage_slices(nd, [&](int u) __attribute__((always_inline)) { return p0 + dirty_pos4(dmk, u); })) lds_barrier();
            }
        }
        if constexpr (MODE == 1) {
            if (sparse_z) {
                // Z of a dirty marker = sum over its missing genotypes of the residual's two exact parts as grid
                // integers -- the integers whose signed base-256 digits are the plane bytes (refresh_planes), so
                // this is the sum the indicator MFMAs form, term by term:  sum a d = X - 3 Z,  sum b d = (sum d) - Z.
                // One wavefront per dirty marker: it scans the marker's slice (SB / 4 dwords), and for every field
                // that reads 11 picks the eight plane bytes of that individual (position 64 chunk + 16 field +
                // 4 dword + byte, the order of operand B).  No f64 work, nothing that depends on who owns whom.
                // Done BEFORE the tile passes: its two dependent LDS round trips cost ~100 cycles each now and
                // several hundred once the four wavefronts stream operands for the MFMAs.
                int r = 0;
#pragma unroll 1
                for (; r < nd; ) {
                    const int m = dirty_pos4(dmk, r);
                    if ((r & 3) == wave) {
                        const int pm = p0 + m;
                        const SliceAt sa = slice_at(pm, r);                              // (an LDS tile, or -- the long-batch kernel -- staging slot r)
                        const char* slice = sa.base;
                        const int swz = sa.swz;
                        long long z1 = 0, z2 = 0;
#pragma unroll
                        for (int q0 = 0; q0 < SB / 4; q0 += 64) {
                            const int q = q0 + lane;                                      // dword of the stored slice
                            const uint32_t w = *reinterpret_cast<const uint32_t*>(slice + 4 * q);
                            uint32_t u = w & (w >> 1) & 0x55555555u;                      // bit 2 i of byte b: field i of byte b reads 11
                            const int cbase = 64 * ((q >> 2) ^ swz) + 4 * (q & 3);        // logical chunk, dword j
                            while (u) {                                                   // rare: a handful of calls per slice
                                const int bit = __ffs((int)u) - 1;
                                u &= u - 1u;
                                const int posn = cbase + 16 * ((bit & 7) >> 1) + (bit >> 3);
                                int v1 = 0, v2 = 0;
#pragma unroll
                                for (int n = 0; n < 4; n++) {
                                    v1 += (int)*reinterpret_cast<const signed char*>(planes + n * PST + posn) << (8 * n);
                                    if (n < 3) v2 += (int)*reinterpret_cast<const signed char*>(planes + (n + 4) * PST + 64 + posn) << (8 * n);
                                }
                                z1 += v1; z2 += v2;
                            }
                        }
                        // the few lanes that found something, one by one (a wavefront-wide sum only if there are many)
                        unsigned long long hm = __ballot((z1 | z2) != 0ll);
                        long long t1 = 0, t2 = 0;
                        if (__popcll(hm) > 8) { t1 = wave_sum64(z1); t2 = wave_sum64(z2); }
                        else
                            while (hm) {
                                const int l = __ffsll((long long)hm) - 1;
                                hm &= hm - 1ull;
                                t1 += (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(z1 >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)z1, l));
                                t2 += (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(z2 >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)z2, l));
                            }
                        if (lane == 0) { s_zsp[2 * r] = t1; s_zsp[2 * r + 1] = t2; }     // read at the publish, after the barrier
                    }
                    r++;
                }
                if constexpr (LONGB) {
                    // the long-batch kernel publishes from the tile passes: the Z terms have to be there first; the dirty markers' own
                    // rows (sum b d = sum d - Z: this slice's sums of q minus Z) leave right here
                    lds_barrier();
                    if (tid < nd) {
                        const long long q1 = (long long)((s_wsq[0] + s_wsq[2] + s_wsq[4] + s_wsq[6]) * 0x1p22);
                        const long long q2 = (long long)((s_wsq[1] + s_wsq[3] + s_wsq[5] + s_wsq[7]) * GRID_INV);
                        put_packed(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)(nb + 1 + tid) * a.Wpad + wg), (b.gen + 1u) & 0xFFFFFFu,
                                   q1 - s_zsp[2 * tid], q2 - s_zsp[2 * tid + 1]);
                    }
                }
            }
        }
        // The genotype values of the markers the walk may cross, as planes of operand B (columns 8, 9): a_s(i) for the
        // phenotyped individuals, 0 for the others and for a missing genotype.  Every thread writes the bytes of its own
        // individuals, from the marker's slice (as in phase C; a register-home marker's slice is staged by its owner first).
        const int ns = b.ns, ps0 = b.ps0, ps1 = b.ps1;
        if constexpr (CONT) {
            if (ns > 0) {                             // (uniform)
                if (stage_slices(ns, [&](int u) __attribute__((always_inline))  { return p0 + (u ? ps1 : ps0); })) lds_barrier();
#pragma unroll 1
                for (int q = 0; q < ns; q++) {
                    const int ps = p0 + (q ? ps1 : ps0);
                    const SliceAt sa = slice_at(ps, q);
                    const char* own = sa.base + 16 * (o_chunk ^ sa.swz) + o_jb;
                    uint32_t pv[ND], pb[ND];
                    int suma = 0, sumb = 0;
#pragma unroll
                    for (int d = 0; d < ND; d++) {
                        const uint32_t x = ((*reinterpret_cast<const uint32_t*>(own + 4 * d) | na_or[d]) >> (2 * o_fld)) & 0x03030303u;
                        const uint32_t miss = x & (x >> 1) & 0x01010101u;            // code 3: missing genotype (or no phenotype)
                        pv[d] = x & ~(miss | (miss << 1));
                        pb[d] = miss ^ 0x01010101u;                                  // b_s: 1 unless missing
                        suma += (int)((pv[d] * 0x01010101u) >> 24);                  // byte sums (<= 8, <= 4)
                        sumb += (int)((pb[d] * 0x01010101u) >> 24);
                    }
                    char* dst = planes + (7 + q) * PST + 64 + p0w;
                    if constexpr (ND == 1) *reinterpret_cast<uint32_t*>(dst) = pv[0];
                    else if constexpr (ND == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(pv[0], pv[1]);
                    else *reinterpret_cast<uint4*>(dst) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                    if constexpr (CK == 2) {
                        // the all-dirty layout has one stop: its second plane holds b_s, and the slice's sums of both planes
                        // (A_s = sum a_s, B_s = sum b_s over the phenotyped individuals) are exchanged like everything else
                        char* dstb = planes + 8 * PST + 64 + p0w;
                        if constexpr (ND == 1) *reinterpret_cast<uint32_t*>(dstb) = pb[0];
                        else if constexpr (ND == 2) *reinterpret_cast<uint2*>(dstb) = make_uint2(pb[0], pb[1]);
                        else *reinterpret_cast<uint4*>(dstb) = make_uint4(pb[0], pb[1], pb[2], pb[3]);
                        int pk = suma | (sumb << 16);                                // both sums of a wavefront fit 16 bits (<= 64 * 32)
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) pk += __shfl_xor(pk, o, 64);
                        if (lane == 0) s_ab[wave] = pk;                              // summed over the wavefronts at the publish
                    }
                }
                lds_barrier();
            }
        }
        // ---- the tile passes: tile T of the batch goes to wavefront T & 3, two tiles (T and T + 4) per pass where there are two
        // operand B: columns 0..6 = the digit planes, 8 and 9 = the planes of the markers the walk may cross; the others
        // (7, 10..15) read plane 0 and their results are dropped
        const int ncol = lane & 15;
        const int ntb = ((p0 + nb - 1) >> 4) - (p0 >> 4) + 1;             // tiles of the batch
        // Packed rows straight from the tile passes (as the long-batch kernel does) when every tile of the batch has ONE owner (four
        // tiles or more: no split passes meeting in LDS atomics) and nothing is merged in at the publish (the sparse Z terms): no
        // LDS round trip, no barrier, no publish loop between a wavefront's last MFMA and its partial sums being on their way.
#if GM_PACK_ROWS
        const bool direct = !LONGB && a.direct_pub && ntb >= 4 && !(MODE == 1 && sparse_z);       // (uniform)
#else
        const bool direct = false;
#endif
        const unsigned dtag24 = (b.gen + 1u) & 0xFFFFFFu;
        const int npair_d = nb + 1 + nd;
        long long dq1 = 0, dq2 = 0;                                       // this slice's sums of q as grid integers (rows of the dirty markers)
        if (direct && MODE != 0) {
            dq1 = (long long)((s_wsq[0] + s_wsq[2] + s_wsq[4] + s_wsq[6]) * 0x1p22);
            dq2 = (long long)((s_wsq[1] + s_wsq[3] + s_wsq[5] + s_wsq[7]) * GRID_INV);
        }
        auto pk_row = [&](int r) __attribute__((always_inline)) -> unsigned long long* {
            return Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)r * a.Wpad + wg);
        };
        auto shl4 = [&](long long v) __attribute__((always_inline)) -> long long {   // the value of lane n + 4 of the row (DPP row_shl:4)
            return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(v >> 32), 0x104, 0xf, 0xf, false) << 32) |
                               (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, false));
        };
        const int poff = ncol < 7 ? ncol * PST + (ncol >> 2) * 64 : ((CONT && (ncol == 8 || ncol == 9)) ? (ncol - 1) * PST + 64 : 0);
        const uint32_t pb0 = lds_addr(planes + poff + kg * 64);
        constexpr uint32_t M0 = 0x03030303u, M1 = 0x01010101u;
        // One pass = ONE or TWO tiles of 16 markers x SS super-steps of the slice; operand B is read once for both.  KR >= 0: the
        // first tile (tR) is register slot KR of this wavefront; NLD LDS-home tiles (tL0, tL1) follow.  All compile-time, so that the
        // body is ONE basic block; the LDS reads of super-step s + 1 are in flight during the arithmetic of s.
        auto tile_pass = [&](auto kr_tag, auto nld_tag, auto fast_tag, auto ns_tag, int tR, int tL0, int tL1, int ss_lo) __attribute__((always_inline))  {
            constexpr int KR = decltype(kr_tag)::value;
            constexpr int NS = decltype(ns_tag)::value;                   // super-steps of the slice this pass covers, from ss_lo (NS < SS: the slice is split
            constexpr bool ATOMIC = NS != SS;                             // over wavefronts, whose parts meet in LDS: integer atomics, exact in any order)
            constexpr int NLD = decltype(nld_tag)::value;
            constexpr int HR = KR >= 0 ? 1 : 0;
            constexpr int NTL = HR + NLD;                             // tiles in this pass
            constexpr bool TF = decltype(fast_tag)::value;            // this batch's layout: no missing genotypes
            static_assert(NTL >= 1 && NTL <= 2, "tiles per pass");
            int tq[NTL];
            if constexpr (HR) tq[0] = tR;
            if constexpr (NLD >= 1) tq[HR] = tL0;
            if constexpr (NLD == 2) tq[1] = tL1;
            uint32_t sl0[2] = {0u, 0u};
            int swz[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                const int pl = 16 * tq[HR + q] + mrow;
                sl0[q] = lds_addr(ring) + (uint32_t)lds_slot(tq[HR + q]) * (uint32_t)TILE_B + (uint32_t)mrow * (uint32_t)SB;
                swz[q] = pl & (CPP - 1);
            }
            v4i acc0[NTL], acc1[NTL], acc2[NTL], zcc0[NTL], zcc1[NTL], zcc2[NTL];   // zcc: general layout, the missing-genotype indicator
#pragma unroll
            for (int q = 0; q < NTL; q++) {
                acc0[q] = v4i{0, 0, 0, 0}; acc1[q] = v4i{0, 0, 0, 0}; acc2[q] = v4i{0, 0, 0, 0};
                zcc0[q] = v4i{0, 0, 0, 0}; zcc1[q] = v4i{0, 0, 0, 0}; zcc2[q] = v4i{0, 0, 0, 0};
            }
            auto waddr = [&](int q, int s2) __attribute__((always_inline))  { return sl0[q] + 16u * (uint32_t)((4 * (ss_lo + s2) + kg) ^ swz[q]); };
            Stage<NLD> stg[2];
            const uint32_t pbk = pb0 + (uint32_t)ss_lo * 256u;
            stg[0] = stage_read<NLD>(pbk, waddr(0, 0), waddr(1, 0));
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) {
                if (s2 + 1 < NS) {
                    stg[(s2 + 1) & 1] = stage_read<NLD>(pbk + (uint32_t)(s2 + 1) * 256u, waddr(0, s2 + 1), waddr(1, s2 + 1));
                    stg[s2 & 1] = stage_wait<NLD, true>(stg[s2 & 1]);
                } else {
                    stg[s2 & 1] = stage_wait<NLD, false>(stg[s2 & 1]);
                }
                const v4i b0 = stg[s2 & 1].b0, b1 = stg[s2 & 1].b1, b2 = stg[s2 & 1].b2, b3 = stg[s2 & 1].b3;
#pragma unroll
                for (int q = 0; q < NTL; q++) {
                    u32x4 w;
                    if constexpr (HR) {
                        if (q == 0) {
                            // (pinned behind this super-step's wait: hipcc otherwise forms the field masks of ALL super-steps of a
                            //  register tile up front -- the data is "there" -- and spills half the register file around them)
                            w = rt[KR < 0 ? 0 : KR][s2];
                            asm volatile("" : "+v"(w));
                        } else w = stg[s2 & 1].w[q - HR];
                    } else w = stg[s2 & 1].w[q];
                    const v4i a0 = {(int)(w.x & M0), (int)(w.y & M0), (int)(w.z & M0), (int)(w.w & M0)};
                    const v4i a1 = {(int)(w.x & (M0 << 2)), (int)(w.y & (M0 << 2)), (int)(w.z & (M0 << 2)), (int)(w.w & (M0 << 2))};
                    const v4i a2 = {(int)(w.x & (M0 << 4)), (int)(w.y & (M0 << 4)), (int)(w.z & (M0 << 4)), (int)(w.w & (M0 << 4))};
                    const v4i a3 = {(int)((w.x >> 6) & M0), (int)((w.y >> 6) & M0), (int)((w.z >> 6) & M0), (int)((w.w >> 6) & M0)};
                    acc0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc0[q], 0, 0, 0);
                    acc1[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc1[q], 0, 0, 0);
                    acc2[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, acc2[q], 0, 0, 0);
                    acc0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b3, acc0[q], 0, 0, 0);
                    if (!TF) {
                        // 1 in the low bit of every field that reads 11 (missing): field & (field >> 1)
                        const u32x4 u = {w.x & (w.x >> 1), w.y & (w.y >> 1), w.z & (w.z >> 1), w.w & (w.w >> 1)};
                        const v4i z0 = {(int)(u.x & M1), (int)(u.y & M1), (int)(u.z & M1), (int)(u.w & M1)};
                        const v4i z1 = {(int)(u.x & (M1 << 2)), (int)(u.y & (M1 << 2)), (int)(u.z & (M1 << 2)), (int)(u.w & (M1 << 2))};
                        const v4i z2 = {(int)(u.x & (M1 << 4)), (int)(u.y & (M1 << 4)), (int)(u.z & (M1 << 4)), (int)(u.w & (M1 << 4))};
                        const v4i z3 = {(int)((u.x >> 6) & M1), (int)((u.y >> 6) & M1), (int)((u.z >> 6) & M1), (int)((u.w >> 6) & M1)};
                        zcc0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(z0, b0, zcc0[q], 0, 0, 0);
                        zcc1[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(z1, b1, zcc1[q], 0, 0, 0);
                        zcc2[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(z2, b2, zcc2[q], 0, 0, 0);
                        zcc0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(z3, b3, zcc0[q], 0, 0, 0);
                    }
                }
            }
#ifdef GM_PA_SPLIT
            PA(1);   // (diagnostic: "scan + inputs" then also holds the MFMA loop, "tiles" only the tile output)
#endif
            // C: column n = lane & 15 (digit plane), rows 4 kg + r (marker of the tile).  The planes of one exact part meet in a
            // quad: planes n, n + 1 as an int32 pair (|sum| <= 2^20 per plane and slice: x + 256 x' fits), the two pairs as a 64-bit
            // integer sum_p digit_p * 256^p.  The tile has ONE owner, so the sums are stored, not added: no LDS atomics, nothing to zero.
            const int n = lane & 15;
            auto put_sum = [&](int slot, long long v) __attribute__((always_inline)) {
                if constexpr (ATOMIC) atomicAdd(reinterpret_cast<unsigned long long*>(&s_sum[slot]), (unsigned long long)v);
                else s_sum[slot] = v;
            };
            auto planes_sum = [&](int x) __attribute__((always_inline)) -> long long {                 // valid in the lanes with (n & 3) == 0
                const int y = x + (__builtin_amdgcn_update_dpp(0, x, DPP_QUAD_1032, 0xf, 0xf, false) << 8);
                const int y2 = __builtin_amdgcn_update_dpp(0, y, 0x4E, 0xf, 0xf, false);      // lane ^ 2
                return (long long)y + ((long long)y2 << 16);
            };
#pragma unroll
            for (int q = 0; q < NTL; q++) {
#if GM_ROWS_BY_LANE
                if constexpr (LONGB && !CONT && TF) {
                    // The four rows (markers 4 kg .. 4 kg + 3) of a lane group leave in ONE store instruction: their sums are formed one
                    // after the other (the quad sums need every lane), then row r's two parts move to lane n = r (row_take64) and lanes
                    // 0..3 pack, address and store together -- instead of four times with one lane of sixteen at work.
                    long long a1 = 0, a2 = 0;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int xr = acc0[q][r] + (acc1[q][r] >> 2) + (acc2[q][r] >> 4);
                        const long long sx = planes_sum(n == 7 ? 0 : xr);
                        const long long s2 = shl4(sx);
                        if (r == 0) { a1 = sx; a2 = s2; }
                        if (r == 1) { a1 = row_take64<1>(a1, sx); a2 = row_take64<1>(a2, s2); }
                        if (r == 2) { a1 = row_take64<2>(a1, sx); a2 = row_take64<2>(a2, s2); }
                        if (r == 3) { a1 = row_take64<3>(a1, sx); a2 = row_take64<3>(a2, s2); }
                    }
                    const int mr = 16 * tq[q] + 4 * kg + n - p0;                     // lane n < 4: batch position of row n
                    if constexpr (MODE == 1) {
                        if (nd > 0) {                          // (uniform) a dirty marker's row: sum a d = X - 3 Z (c = 3 where the genotype is missing)
                            if (n < 4 && (unsigned)mr < (unsigned)nb && dirty_at4(dmk, mr)) {
                                const int rk = dirty_rank4(dmk, mr);
                                a1 -= 3 * s_zsp[2 * rk]; a2 -= 3 * s_zsp[2 * rk + 1];
                            }
                        }
                    }
                    if (n < 4 && (unsigned)mr < (unsigned)nb)
                        put_packed(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)mr * a.Wpad + wg), (b.gen + 1u) & 0xFFFFFFu, a1, a2);
                } else
#endif
                // (the short-batch kernels' direct publish stores row by row: gathering the rows in lanes as above measured slower
                //  there -- c5 +1.5 %, c6 +0.4 % -- four more 64-bit values live across the loop in kernels that spill already)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = 16 * tq[q] + 4 * kg + r - p0;                      // batch position of this row
                    const bool in = (unsigned)m < (unsigned)nb;
                    const int xr = acc0[q][r] + (acc1[q][r] >> 2) + (acc2[q][r] >> 4);  // sum c * digit (c = a wherever the residual is not 0)
                    // columns 0..6: the digits of the two exact parts (the second has three: column 7 is not a plane)
                    const long long sx = planes_sum(n == 7 ? 0 : xr);
                    if (TF) {
                        if constexpr (LONGB) {
                            // the long-batch kernel publishes from here: lane n = 0 takes the second part from lane n = 4 of its row
                            // (DPP row_shl:4) and stores both as one packed granule pair -- no LDS round trip, no barrier, no publish
                            // loop between a wavefront's last MFMA and its partial sums being on their way
                            const long long s2 = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(sx >> 32), 0x104, 0xf, 0xf, false) << 32) |
                                                             (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0x104, 0xf, 0xf, false));
                            if (n == 0 && in)
                                put_packed(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)m * a.Wpad + wg), (b.gen + 1u) & 0xFFFFFFu, sx, s2);
                            if constexpr (CONT) {
                                // columns 8, 9: G = sum a_j a_s over the slice for the stops s the walk may cross, for the markers behind s only:
                                // row nb + 1 + (j - ps0 - 1), both stops in one granule pair (column 9 comes to lane 8 through the quad)
                                if (ns > 0) {              // (uniform)
                                    const int x9 = __builtin_amdgcn_update_dpp(0, xr, DPP_QUAD_1032, 0xf, 0xf, false);
                                    if (n == 8 && m > ps0 && m < nb)
                                        put_packed(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)(nb + m - ps0) * a.Wpad + wg), (b.gen + 1u) & 0xFFFFFFu,
                                                   (long long)xr, (ns > 1 && m > ps1) ? (long long)x9 : 0ll);
                                }
                            }
                        } else if (direct) {
                            const long long s2 = shl4(sx);
                            if (n == 0 && in) put_packed(pk_row(m), dtag24, sx, s2);
                        } else if ((n & 3) == 0 && n < 8 && in) put_sum(2 * m + (n >> 2), sx);
                    } else {
                        // a = c - 3 [missing], b = 1 - [missing]:  sum a d = X - 3 Z,  sum b d = (sum d) - Z; the
                        // slice's sum of d is added at the publish.  (A clean marker in this tile has Z = 0: code 3
                        // occurs for it only where the residual is 0.)
                        const int zr = zcc0[q][r] + (zcc1[q][r] >> 2) + (zcc2[q][r] >> 4);
                        const long long sz = planes_sum(n == 7 ? 0 : zr);
                        if (direct) {
                            const long long va = sx - 3 * sz, va2 = shl4(va), nz2 = shl4(-sz);
                            if (n == 0 && in) {
                                put_packed(pk_row(m), dtag24, va, va2);
                                if (all_dirty || dirty_at(dm0, dm1, m))
                                    put_packed(pk_row(nb + 1 + (all_dirty ? m : dirty_rank(dm0, dm1, m))), dtag24, dq1 - sz, dq2 + nz2);
                            }
                        } else
                        if ((n & 3) == 0 && n < 8 && in) {
                            put_sum(2 * m + (n >> 2), sx - 3 * sz);
                            if (all_dirty || dirty_at(dm0, dm1, m))
                                put_sum(2 * nb + 2 + 2 * (all_dirty ? m : dirty_rank(dm0, dm1, m)) + (n >> 2), -sz);
                        }
                    }
                    // columns 8, 9: G = sum a_j a_s over the slice for the markers s the walk may cross, kept for the markers behind s
                    // only and packed in one slot (column 9 comes to lane 8 through the quad)
                    if constexpr (CONT && !LONGB) {
                        if (ns > 0) {                  // (uniform)
                            if constexpr (CK == 1) {
                                const int x9 = __builtin_amdgcn_update_dpp(0, xr, DPP_QUAD_1032, 0xf, 0xf, false);
                                if (n == 8 && m > ps0 && m < nb) {
                                    if (direct) put_packed(pk_row(npair_d + m - ps0 - 1), dtag24, (long long)xr, (ns > 1 && m > ps1) ? (long long)x9 : 0ll);
                                    else put_sum(2 * nb + 2 + m - ps0 - 1, (long long)xr + ((ns > 1 && m > ps1) ? ((long long)x9 << 24) : 0ll));
                                }
                            } else if constexpr (!TF) {
                                // all-dirty layout: column 8 = a_s, column 9 = b_s of the one stop; per marker behind it
                                //   slot 0: G_a | G_ab << 26 = sum a_j a_s | sum a_j b_s  (a_j = c - 3 [missing]: X - 3 Z)
                                //   slot 1: Z_a | Z_b << 26  = sum [j missing] a_s | sum [j missing] b_s
                                const int zr = zcc0[q][r] + (zcc1[q][r] >> 2) + (zcc2[q][r] >> 4);
                                const int ga = xr - 3 * zr;
                                const int ga9 = __builtin_amdgcn_update_dpp(0, ga, DPP_QUAD_1032, 0xf, 0xf, false);
                                const int zr9 = __builtin_amdgcn_update_dpp(0, zr, DPP_QUAD_1032, 0xf, 0xf, false);
                                if (n == 8 && m > ps0 && m < nb) {
                                    if (direct) {
                                        put_packed(pk_row(npair_d + 1 + 2 * (m - ps0 - 1)), dtag24, (long long)ga, (long long)ga9);
                                        put_packed(pk_row(npair_d + 2 + 2 * (m - ps0 - 1)), dtag24, (long long)zr, (long long)zr9);
                                    } else {
                                        const int sl = 4 * nb + 4 + 2 * (m - ps0 - 1);
                                        put_sum(sl, (long long)ga + ((long long)ga9 << 26));
                                        put_sum(sl + 1, (long long)zr + ((long long)zr9 << 26));
                                    }
                                }
                            }
                        }
                    }
                }
            }
        };
        using ICm1 = std::integral_constant<int, -1>;
        using IC0 = std::integral_constant<int, 0>;
        using IC1 = std::integral_constant<int, 1>;
        using IC2 = std::integral_constant<int, 2>;
        // the 16 positions of tile t hold a dirty marker?  (batch positions 0..127: dm0, dm1)
        auto tile_dirty = [&](int t) __attribute__((always_inline)) -> bool {
            if (MODE != 1) return MODE == 2;
            if (sparse_z) return false;
            int lo = 16 * t - p0, hi = lo + 16;       // batch positions of the tile's rows
            if (lo < 0) lo = 0;
            if (hi > 128) hi = 128;
            bool d = false;
            for (int m = lo; m < hi; m++) d |= dirty_at(dm0, dm1, m);
            return d;
        };
        using ICSS = std::integral_constant<int, SS>;
        auto one_pass = [&](auto kr_tag, auto nld_tag, int tR, int tL0, int tL1, bool dirty) __attribute__((always_inline))  {
            if constexpr (MODE == 0 || (LONGB && MODE == 1)) tile_pass(kr_tag, nld_tag, std::true_type{}, ICSS{}, tR, tL0, tL1, 0);   // (the long-batch kernel gathers the Z terms: sparse_z)
            else if constexpr (MODE == 2) tile_pass(kr_tag, nld_tag, std::false_type{}, ICSS{}, tR, tL0, tL1, 0);
            else { if (dirty) tile_pass(kr_tag, nld_tag, std::false_type{}, ICSS{}, tR, tL0, tL1, 0); else tile_pass(kr_tag, nld_tag, std::true_type{}, ICSS{}, tR, tL0, tL1, 0); }
        };
        // a batch of fewer than four tiles (every tile in LDS: the kernels without register tiles): the slice is split over the
        // wavefronts instead -- ksplit parts per tile, met in LDS through integer atomics (s_sum is zero: the publish leaves it so)
        auto split_pass = [&](auto ns_tag, int t, int ss_lo, bool dirty) __attribute__((always_inline))  {
            if constexpr (MODE == 0) tile_pass(ICm1{}, IC1{}, std::true_type{}, ns_tag, 0, t, 0, ss_lo);
            else if constexpr (MODE == 2) tile_pass(ICm1{}, IC1{}, std::false_type{}, ns_tag, 0, t, 0, ss_lo);
            else { if (dirty) tile_pass(ICm1{}, IC1{}, std::false_type{}, ns_tag, 0, t, 0, ss_lo); else tile_pass(ICm1{}, IC1{}, std::true_type{}, ns_tag, 0, t, 0, ss_lo); }
        };
        if constexpr (LONGB) {
            // row nb of the packed exchange: this slice's sum of q (the wavefronts' shares: refresh_planes, before the barrier above)
            if (tid == 0) {
                const long long p1 = (long long)((s_wsq[0] + s_wsq[2] + s_wsq[4] + s_wsq[6]) * 0x1p22);      // exact: multiples of the grids
                const long long p2 = (long long)((s_wsq[1] + s_wsq[3] + s_wsq[5] + s_wsq[7]) * GRID_INV);
                put_packed(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)nb * a.Wpad + wg), (b.gen + 1u) & 0xFFFFFFu, p1, p2);
            }
        }
        if (!LONGB && ntb < 4) {                                          // (uniform; the long-batch kernel publishes whole tiles from the pass itself)
            const int tb0 = p0 >> 4;
            const int tsplit = ntb >= 2 ? 2 : 1, ksplit = 4 / tsplit;     // the 4 wavefronts = tsplit tile groups x ksplit parts of the slice
            const int wt = wave & (tsplit - 1), wk = wave / tsplit;
#pragma unroll 1
            for (int t = tb0 + wt; t < tb0 + ntb; t += tsplit) {
                if (ksplit == 2) split_pass(std::integral_constant<int, SS / 2>{}, t, wk * (SS / 2), tile_dirty(t));
                else split_pass(std::integral_constant<int, SS / 4>{}, t, wk * (SS / 4), tile_dirty(t));
            }
        } else
        {
            const int tb0 = p0 >> 4, tb1 = (p0 + nb - 1) >> 4;          // tiles of the batch
            int t = tb0 + ((wave - tb0) & 3);                           // this wavefront's first: t & 3 == wave
#pragma unroll 1
            for (; t <= tb1; t += 8) {
                const bool two = t + 4 <= tb1;
                const bool reg0 = is_reg(t), reg1 = two && is_reg(t + 4);   // (at most one of a pair lives in registers: T & 7 and (T + 4) & 7)
                if (!reg0 && !reg1) {
                    if (two) one_pass(ICm1{}, IC2{}, 0, t, t + 4, tile_dirty(t) || tile_dirty(t + 4));
                    else one_pass(ICm1{}, IC1{}, 0, t, 0, tile_dirty(t));
                } else if constexpr (NP > 0) {
                    const int tr = reg0 ? t : t + 4, tl = reg0 ? t + 4 : t;
                    const int k = reg_slot(tr);
                    const bool dirty = tile_dirty(t) || (two && tile_dirty(t + 4));
                    auto go = [&](auto k_tag) __attribute__((always_inline))  {
                        if (two) one_pass(k_tag, IC1{}, tr, tl, 0, dirty);
                        else one_pass(k_tag, IC0{}, tr, 0, 0, dirty);
                    };
                    if (k == 0) go(IC0{});
                    if constexpr (NP > 1) { if (k == 1) go(IC1{}); }
                    if constexpr (NP > 2) { if (k == 2) go(IC2{}); }
                }
            }
        }
        PA(2);
        if constexpr (LONGB) {                        // (published from the tile passes: nothing to collect)
            b.nv = nb + 1 + ((CONT && ns > 0) ? nb - 1 - ps0 : 0) + (MODE == 1 ? nd : 0);   // behind a crossed stop: one more row per marker (the packed G); a row per dirty marker
            b.nr = b.nv;
            PA(3);
            PA(4);
            return;
        }
        const int nv0 = 2 * nb + 2 + 2 * nd;
        // behind a crossed stop: one more value per marker (two, and two for the stop, in the all-dirty layout)
        const int nv = nv0 + ((CONT && ns > 0) ? (CK == 2 ? 2 + 2 * (nb - 1 - ps0) : nb - 1 - ps0) : 0);
        if (direct) {                                 // (uniform) the rows left the tile passes; the common row and the stop's own sums follow
            if (tid == 0)
                put_packed(pk_row(nb), dtag24, (long long)((s_wsq[0] + s_wsq[2] + s_wsq[4] + s_wsq[6]) * 0x1p22),
                           (long long)((s_wsq[1] + s_wsq[3] + s_wsq[5] + s_wsq[7]) * GRID_INV));
            if constexpr (CK == 2) {
                if (ns > 0 && tid == 64)
                    put_packed(pk_row(npair_d), dtag24, (s_ab[0] & 0xFFFF) + (s_ab[1] & 0xFFFF) + (s_ab[2] & 0xFFFF) + (s_ab[3] & 0xFFFF),
                               ((s_ab[0] >> 16) & 0xFFFF) + ((s_ab[1] >> 16) & 0xFFFF) + ((s_ab[2] >> 16) & 0xFFFF) + ((s_ab[3] >> 16) & 0xFFFF));
            }
            b.nv = nv;
            b.nr = npair_d + ((CONT && ns > 0) ? (CK == 2 ? 1 + 2 * (nb - 1 - ps0) : nb - 1 - ps0) : 0);
            PA(3);
            PA(4);
            return;
        }
        lds_barrier();                                // the LDS sums are complete (tile loads stay in flight)
        PA(3);
#if GM_PACK_ROWS
        // Packed rows (round 4, as in the long-batch kernel): the two exact parts of a value -- slots 2 r and 2 r + 1: a marker's
        // sum, the common sum, a dirty marker's Z -- leave as ONE granule pair (put_packed), so a batch has half the rows and a
        // reducer seldom more than one; the reducers publish the totals slot by slot as before (reduce_role), the walk does not
        // notice.  Behind the pairs: the rows of a crossed stop (one per marker behind it with the packed genotype products
        // split into the two fields; in the all-dirty layout the stop's own sums first, then two rows per marker).
        const int npair = nb + 1 + nd;                // (2 npair == nv0)
        const int nxr = (CONT && ns > 0) ? (CK == 2 ? 1 + 2 * (nb - 1 - ps0) : nb - 1 - ps0) : 0;
        auto val_at = [&](int vi) __attribute__((always_inline)) -> long long {    // the integer of slot vi < nv0 (grid units of its part)
            if (vi >= 2 * nb && vi < 2 * nb + 2) {    // sum q1, sum q2 over the slice
                const int w2 = vi - 2 * nb;
                return (long long)((s_wsq[w2] + s_wsq[2 + w2] + s_wsq[4 + w2] + s_wsq[6 + w2]) * (w2 ? GRID_INV : 0x1p22));   // exact: multiples of the grids
            }
            long long v = s_sum[vi];
            if constexpr (MODE == 1) {
                if (sparse_z) {                       // X - 3 Z for the a-sums of a dirty marker, -Z in its own b-slots
                    if (vi >= 2 * nb + 2) v = -s_zsp[vi - (2 * nb + 2)];
                    else if (dirty_at(dm0, dm1, vi >> 1)) v -= 3 * s_zsp[2 * dirty_rank(dm0, dm1, vi >> 1) + (vi & 1)];
                }
            }
            if (vi >= 2 * nb + 2) {                  // dirty marker: sum b d = sum d - Z, this slice's sum of the exact part as an integer
                const int w2 = vi & 1;
                const double dsum = s_wsq[w2] + s_wsq[2 + w2] + s_wsq[4 + w2] + s_wsq[6 + w2];
                v += (long long)(dsum * (w2 ? GRID_INV : 0x1p22));
            }
            s_sum[vi] = 0;                           // (split passes add into these slots)
            return v;
        };
        for (int r = tid; r < npair + nxr; r += SW_TPB) {
            long long p1, p2;
            if (r < npair) { p1 = val_at(2 * r); p2 = val_at(2 * r + 1); }
            else {
                const int k = r - npair;
                if constexpr (CK == 2) {
                    if (k == 0) {                     // the stop's own sums A_s, B_s: the wavefronts' shares of the plane build
                        p1 = (s_ab[0] & 0xFFFF) + (s_ab[1] & 0xFFFF) + (s_ab[2] & 0xFFFF) + (s_ab[3] & 0xFFFF);
                        p2 = ((s_ab[0] >> 16) & 0xFFFF) + ((s_ab[1] >> 16) & 0xFFFF) + ((s_ab[2] >> 16) & 0xFFFF) + ((s_ab[3] >> 16) & 0xFFFF);
                        s_sum[nv0] = 0; s_sum[nv0 + 1] = 0;
                    } else {
                        const long long v = s_sum[nv0 + 1 + k];
                        s_sum[nv0 + 1 + k] = 0;
                        p1 = v & 0x3FFFFFFll; p2 = v >> 26;
                    }
                } else {
                    const long long v = s_sum[nv0 + k];
                    s_sum[nv0 + k] = 0;
                    p1 = v & 0xFFFFFFll; p2 = v >> 24;
                }
            }
            put_packed(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)r * a.Wpad + wg), (b.gen + 1u) & 0xFFFFFFu, p1, p2);
        }
        b.nr = npair + nxr;
#else
        if constexpr (CONT) {
            if (ns > 0) {                             // (uniform) packed G counts: integers < 2^52, exact as doubles
                for (int vi = nv0 + tid; vi < nv; vi += SW_TPB) {
                    long long v = s_sum[vi];
                    s_sum[vi] = 0;
                    if constexpr (CK == 2) {          // the stop's own sums A_s, B_s: the wavefronts' shares of the plane build
                        if (vi < nv0 + 2) {
                            const int sh = vi == nv0 ? 0 : 16;
                            v = ((s_ab[0] >> sh) & 0xFFFF) + ((s_ab[1] >> sh) & 0xFFFF) + ((s_ab[2] >> sh) & 0xFFFF) + ((s_ab[3] >> sh) & 0xFFFF);
                        }
                    }
                    put_value(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)vi * a.Wpad + wg), b.gen + 1u, (double)v);
                }
            }
        }
        for (int vi = tid; vi < nv0; vi += SW_TPB) {          // up to SW_VMAX values, 256 threads
            double tot;
            if (vi >= 2 * nb && vi < 2 * nb + 2) {  // sum q1, sum q2 over the slice
                const int w2 = vi - 2 * nb;
                tot = s_wsq[w2] + s_wsq[2 + w2] + s_wsq[4 + w2] + s_wsq[6 + w2];
            } else {
                long long v = s_sum[vi];
                if constexpr (MODE == 1) {
                    if (sparse_z) {                   // X - 3 Z for the a-sums of a dirty marker, -Z in its own b-slots
                        if (vi >= 2 * nb + 2) v = -s_zsp[vi - (2 * nb + 2)];
                        else if (dirty_at(dm0, dm1, vi >> 1)) v -= 3 * s_zsp[2 * dirty_rank(dm0, dm1, vi >> 1) + (vi & 1)];
                    }
                }
                if (vi >= 2 * nb + 2) {              // dirty marker: sum b d = sum d - Z, this slice's sum of the exact part as an integer
                    const int w2 = vi & 1;
                    const double dsum = s_wsq[w2] + s_wsq[2 + w2] + s_wsq[4 + w2] + s_wsq[6 + w2];
                    v += (long long)(dsum * (w2 ? GRID_INV : 0x1p22));              // exact: a multiple of the grid, < 2^53 units
                }
                // |sum| < 2^53 grid units: the conversion and the power-of-two scaling are exact
                tot = (double)v * ((vi & 1) ? GRID : 0x1p-22);
                s_sum[vi] = 0;                       // (split passes add into these slots)
            }
            put_value(Pg + 2 * ((size_t)(b.gen & 1u) * SW_VMAX * a.Wpad + (size_t)vi * a.Wpad + wg), b.gen + 1u, tot);
        }
        b.nr = nv;
#endif
        b.nv = nv;
        PA(4);
    };

    // reduce role: workgroup v sums row v of generation b.gen over all workgroups
    auto reduce_role = [&](const Batch& b) __attribute__((always_inline)) -> bool {
        bool bad = false;
        if (wg < b.nr) {
            const unsigned long long* Pb = Pg + 2 * (size_t)(b.gen & 1u) * SW_VMAX * a.Wpad;
            unsigned long long* Tb = Ttg + 2 * (size_t)(b.gen & 1u) * SW_VMAX;
            const int rows = (b.nr - wg + W - 1) / W;         // rows wg, wg + W, ... of this workgroup (uniform)
            if constexpr (LONGB) {
                // packed rows (put_packed): both exact parts of a marker arrive in one granule pair; the reducer sums each part
                // (exact: any order) and publishes ONE total, the parts added with one rounding -- what the walk did with them.
                // Wavefront 0 ALONE reduces (four granule pairs per lane and look): the loader wavefronts' tile loads have to
                // flow for ~4 us per round (57 KB per workgroup, all workgroups at once), a load of theirs comes back behind
                // whatever they requested before it, and the two exchange hops are the only stretch of the round in which
                // nothing else needs them.  They start loading right behind the publish; wavefront 0 reduces and polls.
                const unsigned tag24 = (b.gen + 1u) & 0xFFFFFFu;
                // a row's two sums (exact integers as doubles) become its total: one rounding of the exact sum -- or, in the kernel that
                // crosses stops, the exact sum itself (put_total_x; rows behind nb: the two stops' genotype products, packed)
                auto put_row = [&](int v, double x1, double x2) __attribute__((always_inline)) {
                    if constexpr (CONT) {
                        const long long p1 = (long long)x1, p2 = (long long)x2;
                        if (v <= b.nb) put_total_x(Tb + 2 * v, tag24, p1 + (p2 >> 22), (unsigned)(p2 & 0x3FFFFFll));
                        else put_total_x(Tb + 2 * v, tag24, p1 | (p2 << 24), 0u);
                    } else put_value(Tb + 2 * v, b.gen + 1u, x1 * 0x1p-22 + x2 * GRID);
                };
                if (rows > 2) {
                    // far fewer workgroups than rows (N = 50 000: 49 workgroups, up to 241 rows): row after row on one wavefront
                    // is a memory round trip per row; all four wavefronts take granule pairs t, t + 256, ... of the rows x W this
                    // workgroup needs and add each to its row's LDS accumulators (ds_add_f64: exact values, any order)
                    double* s_rows = reinterpret_cast<double*>(smem + L_TOT);       // free until wavefront 0 polls the totals
                    for (int r = tid; r < 2 * rows; r += SW_TPB) s_rows[r] = 0.0;
                    lds_barrier();
                    const int total = rows * W;
                    for (int g = tid; g < total; g += SW_TPB) {
                        const int r = g / W, i = g - r * W;
                        Spin sp;
                        sp.start(spin_limit);
                        double x1 = 0.0, x2 = 0.0;
                        const unsigned long long* gp = Pb + 2 * ((size_t)(wg + r * W) * a.Wpad + i);
                        while (!get_packed(gp, tag24, x1, x2)) {
                            if (sp.expired(abort_word)) { bad = true; x1 = 0.0; x2 = 0.0; break; }
                        }
                        __hip_atomic_fetch_add(&s_rows[2 * r], x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&s_rows[2 * r + 1], x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    lds_barrier();
                    for (int r = tid; r < rows; r += SW_TPB) put_row(wg + r * W, s_rows[2 * r], s_rows[2 * r + 1]);
                    lds_barrier();
                    return bad;
                }
                if (wave != 0) return false;
                for (int r = 0; r < rows; r++) {              // (uniform; one row, two where there are fewer than 241 workgroups)
                    const int v = wg + r * W;
                    const unsigned long long* rowp = Pb + 2 * (size_t)v * a.Wpad;
                    double x1 = 0.0, x2 = 0.0;
                    unsigned got = 0u;
                    const unsigned want = (lane < W ? 1u : 0u) | (64 + lane < W ? 2u : 0u) | (128 + lane < W ? 4u : 0u) | (192 + lane < W ? 8u : 0u);
                    Spin sp;
                    sp.start(spin_limit);
                    for (;;) {
#ifdef GM_SWEEP_PROF
                        if (lane == 0) reinterpret_cast<unsigned long long*>(smem + L_M + 64)[4]++;      // looks at the row of partial sums
#endif
                        get_packed4(rowp, lane, tag24, W, x1, x2, got);
                        if (__all(got == want)) break;
                        if (sp.expired(abort_word)) { bad = true; break; }
                    }
                    const double r2 = reduce2(x1, x2);        // lanes 0-31: sum of x1 over the wavefront, lanes 32-63: of x2
                    const double t1 = readlane64(r2, 0), t2 = readlane64(r2, 32);
                    if (lane == 0) put_row(v, t1, t2);
                }
                return __any(bad);
            }
#if GM_PACK_ROWS
            // The other kernels: packed rows as well (compute_publish), all four wavefronts in the reduce role (their batches are
            // short, the loads few: measured c2 22.7 against 23.6 ms, c6 148 against 154, c5 equal with wavefront 0 alone).  A row's
            // two sums become the totals of the slots the walk reads: 2 r and 2 r + 1 for the pairs, one packed integer for the
            // rows of a crossed stop.
            {
                const unsigned tag24 = (b.gen + 1u) & 0xFFFFFFu;
                const int nd_ = MODE == 2 ? b.nb : __popcll(b.dm0) + __popcll(b.dm1);
                const int npair = b.nb + 1 + nd_, nv0 = 2 * npair;
                auto emit_row = [&](int r, double x1, double x2) __attribute__((always_inline)) {
                    if (r < npair) {
                        put_value(Tb + 2 * (2 * r), b.gen + 1u, x1 * 0x1p-22);
                        put_value(Tb + 2 * (2 * r + 1), b.gen + 1u, x2 * GRID);
                    } else {
                        const int k = r - npair;
                        if constexpr (CK == 2) {
                            if (k == 0) { put_value(Tb + 2 * nv0, b.gen + 1u, x1); put_value(Tb + 2 * (nv0 + 1), b.gen + 1u, x2); }
                            else put_value(Tb + 2 * (nv0 + 1 + k), b.gen + 1u, (double)((long long)x1 | ((long long)x2 << 26)));
                        } else put_value(Tb + 2 * (nv0 + k), b.gen + 1u, (double)((long long)x1 | ((long long)x2 << 24)));
                    }
                };
                if (rows == 1) {
                    double x1 = 0.0, x2 = 0.0;
                    if (tid < W) {
                        Spin sp;
                        sp.start(spin_limit);
                        const unsigned long long* gp = Pb + 2 * ((size_t)wg * a.Wpad + tid);
                        while (!get_packed(gp, tag24, x1, x2)) {
                            if (sp.expired(abort_word)) { bad = true; x1 = 0.0; x2 = 0.0; break; }
                        }
                    }
                    const double r2 = reduce2(x1, x2);        // lanes 0-31: sum of x1 over the wavefront, lanes 32-63: of x2 (exact values: any order)
                    if (lane == 0) s_red[wave] = r2;
                    if (lane == 32) s_red[4 + wave] = r2;
                    lds_barrier();
                    if (tid == 0) emit_row(wg, s_red[0] + s_red[1] + s_red[2] + s_red[3], s_red[4] + s_red[5] + s_red[6] + s_red[7]);
                    lds_barrier();
                } else {
                    // several rows (few workgroups, or a long tail behind a crossed stop): every thread takes granule pairs t, t + 256, ...
                    // of the rows x W this workgroup needs and adds each to its row's LDS accumulators (ds_add_f64: exact values, any order)
                    double* s_rows = reinterpret_cast<double*>(smem + L_TOT);       // free until wavefront 0 polls the totals
                    for (int r = tid; r < 2 * rows; r += SW_TPB) s_rows[r] = 0.0;
                    lds_barrier();
                    const int total = rows * W;
                    for (int g = tid; g < total; g += SW_TPB) {
                        const int r = g / W, i = g - r * W;
                        Spin sp;
                        sp.start(spin_limit);
                        double x1 = 0.0, x2 = 0.0;
                        const unsigned long long* gp = Pb + 2 * ((size_t)(wg + r * W) * a.Wpad + i);
                        while (!get_packed(gp, tag24, x1, x2)) {
                            if (sp.expired(abort_word)) { bad = true; x1 = 0.0; x2 = 0.0; break; }
                        }
                        __hip_atomic_fetch_add(&s_rows[2 * r], x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&s_rows[2 * r + 1], x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    lds_barrier();
                    for (int r = tid; r < rows; r += SW_TPB) emit_row(wg + r * W, s_rows[2 * r], s_rows[2 * r + 1]);
                    lds_barrier();
                }
                return bad;
            }
#else
            if (!a.reduce4 && rows <= 2) {
                if (wave != 0) return false;
                for (int r = 0; r < rows; r++) {              // (uniform)
                    const int v = wg + r * W;
                    const unsigned long long* rowp = Pb + 2 * (size_t)v * a.Wpad;
                    double x = 0.0;
                    unsigned got = 0u;
                    const unsigned want = (lane < W ? 1u : 0u) | (64 + lane < W ? 2u : 0u) | (128 + lane < W ? 4u : 0u) | (192 + lane < W ? 8u : 0u);
                    Spin sp;
                    sp.start(spin_limit);
                    for (;;) {
                        u32x4 d[4];
                        get_row4(rowp, lane, d);
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (((want & ~got) >> k) & 1u) {
                                if (d[k].y == b.gen + 1u && d[k].w == b.gen + 1u) {
                                    x += __longlong_as_double((long long)(((unsigned long long)d[k].z << 32) | d[k].x));   // exact values: any order
                                    got |= 1u << k;
                                }
                            }
                        if (__all(got == want)) break;
                        if (sp.expired(abort_word)) { bad = true; break; }
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                    if (lane == 0) put_value(Tb + 2 * v, b.gen + 1u, x);
                }
                return __any(bad);
            }
            if (rows == 1) {
                const int v = wg;
                double x = 0.0;
                if (tid < W) {
                    Spin sp;
                    sp.start(spin_limit);
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
            } else if (rows == 2) {
                // a long batch (or one with markers behind a crossed stop, or many dirty markers) has more rows than there are
                // workgroups: rows wg and wg + W in one pass, both granules of a thread in flight together
                double x0 = 0.0, x1 = 0.0;
                if (tid < W) {
                    Spin sp;
                    sp.start(spin_limit);
                    const unsigned long long* gp0 = Pb + 2 * ((size_t)wg * a.Wpad + tid);
                    const unsigned long long* gp1 = Pb + 2 * ((size_t)(wg + W) * a.Wpad + tid);
                    bool ok0 = false, ok1 = false;
                    for (;;) {
                        get_value2(gp0, gp1, b.gen + 1u, x0, ok0, x1, ok1);
                        if (ok0 && ok1) break;
                        if (sp.expired(abort_word)) { bad = true; x0 = 0.0; x1 = 0.0; break; }
                    }
                }
                const double r2 = reduce2(x0, x1);            // lanes 0-31: sum of x0 over the wavefront, lanes 32-63: of x1 (exact values: any order)
                if (lane == 0) s_red[wave] = r2;
                if (lane == 32) s_red[4 + wave] = r2;
                lds_barrier();
                if (tid == 0) put_value(Tb + 2 * wg, b.gen + 1u, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
                if (tid == 64) put_value(Tb + 2 * (wg + W), b.gen + 1u, s_red[4] + s_red[5] + s_red[6] + s_red[7]);
                lds_barrier();
            } else {
                // Fewer workgroups than rows (small N: 49 workgroups, up to 482 rows): all rows of this workgroup in ONE
                // pass -- thread t waits for granules t, t + 256, ... of the rows x W it needs and adds each to its row's
                // LDS accumulator (ds_add_f64; the values are exact, so the order does not matter).  One row after the
                // other cost a memory round trip and two barriers per row.
                double* s_rows = reinterpret_cast<double*>(smem + L_TOT);       // free until wavefront 0 polls the totals
                for (int r = tid; r < rows; r += SW_TPB) s_rows[r] = 0.0;
                lds_barrier();
                const int total = rows * W;
                for (int g = tid; g < total; g += SW_TPB) {
                    const int r = g / W, i = g - r * W;
                    Spin sp;
                    sp.start(spin_limit);
                    double x = 0.0;
                    const unsigned long long* gp = Pb + 2 * ((size_t)(wg + r * W) * a.Wpad + i);
                    while (!get_value(gp, b.gen + 1u, x)) {
                        if (sp.expired(abort_word)) { bad = true; x = 0.0; break; }
                    }
                    __hip_atomic_fetch_add(&s_rows[r], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                lds_barrier();
                for (int r = tid; r < rows; r += SW_TPB) put_value(Tb + 2 * (wg + r * W), b.gen + 1u, s_rows[r]);
                lds_barrier();
            }
#endif
        }
        return bad;
    };

    Batch cur{0, 0, 0, 0u, false, 0ull, 0ull, 0, 0, 0, 0, 0ull, 0ull};
    const int tdelay = LONGB ? a.totals_delay : a.totals_delay2;    // wavefront 0: s_sleep units (64 clocks) before the first look at the totals (see there)
    LaneIn li_cur0{0, 0, 0.0, 0.0, 1.0}, li_cur1{0, 0, 0.0, 0.0, 1.0};
    bool bad = false;
    ensure_meta(META_POS < a.M ? META_POS : a.M);
    __syncthreads();
    while (pos < a.M) {
        TRACE(0);
        PT(0);
        // ---- the batch: what the run-length estimate asks for, as far as tiles and sampling inputs are on chip
        bool fill_round = false;                      // (uniform) nothing to walk yet: only request tiles (start-up; a walk that ran through a short window)
        {
            int nb0 = ctl[C_NBNEXT];
            if (nb0 > a.M - pos) nb0 = a.M - pos;
            int have = 16 * t_hi - pos;               // (tiles beyond the block's end do not exist: nb0 <= M - pos)
            if (have < nb0 && have < 16) fill_round = true;
            if (!fill_round && mhi - pos < 16 && mhi < a.M) ensure_meta(pos + 64 < a.M ? pos + 64 : a.M);   // (slow path; the refill keeps up in practice)
            if (mhi - pos < have) have = mhi - pos;
            if (have < nb0) { nb0 = have; n_short++; }
            // Tiles are 16-aligned order positions and go to wavefront T & 3: a batch that straddles 4 k + 1 tiles gives ONE wavefront
            // a tile more than the others (64 markers from the middle of a tile: five tiles, a pair pass for wavefront 0, one tile
            // for the others, who wait at the barrier).  Dropping the last, partial tile -- at most 15 markers, which the walk
            // seldom reaches -- evens the passes out.
            if (!LONGB && a.tile_trim) {           // (the long batches: 240 markers are 15 or 16 tiles either way, and measured no gain)
                const int off = pos & 15, tl = (off + nb0 + 15) >> 4;
                if (off && tl >= 5 && (tl & 3) == 1) {
                    const int nbt = 16 * (tl - 1) - off;
                    if (4 * nbt >= 3 * nb0) nb0 = nbt;
                }
            }
            cur.p0 = pos; cur.nb = nb0; cur.gen = gen_next;
        }
        if (fill_round) {
            // ONE inlined copy of the tile loads serves start-up, this slow path and the steady state (below): more copies made the
            // register allocator keep a slot's registers in different places at different sites, with copies of registers whose
            // loads were in flight in between (tools/check_prefetch_regs.py)
            n_short++;
        }
        PT(1);
        tiles_wait();                                 // the tiles requested a round ago are there (no wait in practice)
        PT(2);
        if (!fill_round) {
        gen_next++;
        compute_publish(cur, li_cur0, li_cur1);
        TRACE(1);
        PROF(1);   // dots + publish
        bad = reduce_role(cur);
        TRACE(2);
        PROF(0);   // reduce role

        }   // !fill_round
        // The walk before this one has released tiles and sampling inputs: request what the window and the meta ring can take
        // now (wavefronts 1-3; no gate: loads issued here, while the totals are on their way, are back before the next round's
        // top -- behind a gate that opened when wavefront 0 had the totals they landed ~1 us late, 2.3 % of a sweep).  What
        // is requested here is used from the NEXT round on.
        {
            int t_lim = (pos >> 4) + WIN;
            if (t_lim > ntiles) t_lim = ntiles;
            tile_issue(t_lim, fill_round ? 0 : pos + META_POS + cur.nb);
        }
        PROF(2);   // tile / meta requests
        if (fill_round) { lds_barrier(); continue; }  // (the loads are waited for at the top of the next round)

        // ---- sampling step of the current batch (every workgroup, identical inputs)
        if constexpr (LONGB) {
            // parallel passes (pass_eval / pass_commit above): wavefront 0 fetches the totals row and parks it in LDS -- the other
            // wavefronts have tile loads in flight, and a load of theirs would come back behind those --, every wavefront
            // evaluates its 64 positions, one barrier, then the commits
            const SampleOut so{a.acum, a.betas_out, a.comp};
            const int nbw = cur.nb - 64 * wave;                                  // positions of this wavefront's pass (<= 0: none)
            double* s_tot = reinterpret_cast<double*>(smem + L_TOT);
            int* s_res = reinterpret_cast<int*>(smem + L_AB);                    // int[8]: every pass's first stop, every pass's draws (free in this kernel)
            // draws of the passes in front of mine: one per marker whose group has sigmaG != 0 -- known without the totals
            int cursor_w = ctl[C_CURSOR];
            for (int v = 0; v < wave; v++) {
                const int gv = mr_g[(cur.p0 + 64 * v + lane) & (META_POS - 1)];
                cursor_w += __popcll(__ballot(64 * v + lane < cur.nb && s_tab[gv] != 0.0));
            }
            bool okw = true;
            if (wave == 0) {
                const unsigned long long* Tb = Ttg + 2 * (size_t)(cur.gen & 1u) * SW_VMAX;
                Totals t0{0.0, 0.0, 0.0, 0.0}, t1{0.0, 0.0, 0.0, 0.0};
                // The totals cannot be complete before two memory round trips have passed (the other reducers' looks, then mine), and a
                // look that fails is not free: 245 workgroups looking at the same 4 KB 3.7 times per round slow the round trips of
                // everybody (measured: a wait of 1.2 us before the first look, -1.7 % on 500k x 1M; two looks in flight, +4 %:
                // profiles/r04_ab_totals_delay.txt, r04_ab_poll_two_looks.txt).  The host picks the wait from the grid's size.
                // (a short batch has few rows, few reducers and its totals early: 770 -> 758 ms for the first sweep of a chain without the wait)
                const int tdl = cur.nv >= 128 ? tdelay : (cur.nv >= 64 ? tdelay / 2 : 0);
                for (int i = 0; i < tdl; i++) __builtin_amdgcn_s_sleep(1);
                int nlooks = 0;
                if constexpr (CONT) okw = poll_totals_x(cur.nv, Tb, (cur.gen + 1u) & 0xFFFFFFu, smem, abort_word, spin_limit);
                else okw = poll_totals<true>(cur.nb, cur.nv, 0ull, 0ull, Tb, cur.gen + 1u, smem, t0, t1, abort_word, spin_limit, nlooks);

                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the row is in LDS before the word that says so
                if (lane == 0) *reinterpret_cast<volatile int*>(&ctl[C_TOTF]) = okw ? (int)(cur.gen + 1u) : -1;
                bad |= !okw;
            } else if (nbw > 0) {
                Spin sp;
                sp.start(spin_limit);
                for (;;) {
                    const int f = *reinterpret_cast<const volatile int*>(&ctl[C_TOTF]);
                    if (f == (int)(cur.gen + 1u)) break;
                    if (f == -1 || sp.expired(abort_word)) { okw = false; break; }
                }
            }
            TRACE(3);
            PROF(4);   // wait for the totals
            if constexpr (CONT) {
                const long long* s_toth = reinterpret_cast<const long long*>(smem + L_TOT);
                const int* s_totl = reinterpret_cast<const int*>(smem + L_SUM);
                if (K == 4) long_cont_body<4>(cur.nb, wave, okw && nbw > 0, okw, cursor_w, li_cur0, s_toth, s_totl, s_res, a.sigmae, a.inv2sige, a.nm1, G, smem, (TabLds)s_tab,
                                              so, wg == 0, a.lds_cass, bcap_rt, a.nb_factor16, cur.ns, cur.ps0, cur.ps1);
                else long_cont_other_k(K, cur.nb, wave, okw && nbw > 0, okw, cursor_w, li_cur0, s_toth, s_totl, s_res, a.sigmae, a.inv2sige, a.nm1, G, smem, (TabLds)s_tab,
                                       so, wg == 0, a.lds_cass, bcap_rt, a.nb_factor16, cur.ns, cur.ps0, cur.ps1);
            } else {
            const DirtyMasks dkm{cur.dm0, cur.dm1, cur.dm2, cur.dm3};
            if (K == 4) long_batch_body<4, MODE == 1>(cur.nb, wave, okw && nbw > 0, okw, cursor_w, li_cur0, s_tot, dkm, s_res, a.sigmae, a.inv2sige, a.nm1, G, smem, (TabLds)s_tab,
                                           so, wg == 0, a.lds_cass, bcap_rt, a.nb_factor16);
            else long_batch_other_k<MODE == 1>(K, cur.nb, wave, okw && nbw > 0, okw, cursor_w, li_cur0, s_tot, dkm, s_res, a.sigmae, a.inv2sige, a.nm1, G, smem, (TabLds)s_tab,
                                    so, wg == 0, a.lds_cass, bcap_rt, a.nb_factor16);
            }
        } else
        if (wave == 0) {
            const SampleOut so{a.acum, a.betas_out, a.comp};
            const unsigned long long* Tb = Ttg + 2 * (size_t)(cur.gen & 1u) * SW_VMAX;
            Totals tot0{0.0, 0.0, 0.0, 0.0}, tot1{0.0, 0.0, 0.0, 0.0};
            const Draws draws = sample_prepare(cur.nb, smem, (TabLds)s_tab, li_cur0.g, li_cur1.g);
            for (int i = 0; i < tdelay; i++) __builtin_amdgcn_s_sleep(1);               // (as in the long-batch kernel: see there)
            int nlooks = 0;
            const bool okw = poll_totals<LONGB>(cur.nb, cur.nv, cur.dm0, cur.dm1, Tb, cur.gen + 1u, smem, tot0, tot1, abort_word, spin_limit, nlooks);

            if (lane == 0) *reinterpret_cast<volatile int*>(&ctl[C_TOTF]) = (int)(cur.gen + 1u);   // the loaders may start
            TRACE(3);
            PROF(4);   // wait for the totals
            bad |= !okw;
            if (okw) {
                const int mpos0 = cur.p0 & (META_POS - 1);
                // the tables are read through a pointer of known address space (LDS)
                auto run_step = [&](auto tabq) __attribute__((always_inline))  {
                    if (K == 4) {
                        sample_batch_body<4, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, li_cur0, li_cur1, tot0, tot1, draws, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1);
                    } else {
                        // out-of-line copies take their inputs by address: hand them copies, so that the
                        // loop-carried lane inputs themselves stay in registers (no scratch round trips)
                        const LaneIn lc0 = li_cur0, lc1 = li_cur1;
                        const Totals tc0 = tot0, tc1 = tot1;
                        switch (K) {
                            case 2: sample_batch<2, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, lc0, lc1, tc0, tc1, draws.p0, draws.p1, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1); break;
                            case 3: sample_batch<3, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, lc0, lc1, tc0, tc1, draws.p0, draws.p1, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1); break;
                            case 5: sample_batch<5, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, lc0, lc1, tc0, tc1, draws.p0, draws.p1, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1); break;
                            case 6: sample_batch<6, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, lc0, lc1, tc0, tc1, draws.p0, draws.p1, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1); break;
                            case 7: sample_batch<7, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, lc0, lc1, tc0, tc1, draws.p0, draws.p1, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1); break;
                            default: sample_batch<8, CK, LONGB>(cur.nb, mpos0, BCAP, a.nb_factor16, G, smem, tabq, lc0, lc1, tc0, tc1, draws.p0, draws.p1, a.sigmae, a.inv2sige, a.nm1, so, wg == 0, a.lds_cass, cur.ns, cur.ps0, cur.ps1); break;
                        }
                    }
                };
                run_step((TabLds)s_tab);
            }
        }
        PROF(7);   // sampling step
        if (bad) ctl[C_BAD] = 1;
        lds_barrier();                                // (tile loads stay in flight)
        if (ctl[C_BAD] || ctl[C_RNGERR] || ctl[C_RANGE]) { ok = false; break; }
        TRACE(5);
        PROF(5);   // barrier after sampling

        // ---- phase C: residual updates of the round (the markers the walk crossed and the one it stopped at) ----
        const int n_done = ctl[C_NDONE];
        const int nupd = ctl[C_UPD];
        const bool upd = nupd != 0;
        n_planned += ctl[C_PLN];
        n_cross += ctl[C_SUPD];
        n_fastb += (cur.dm0 | cur.dm1 | cur.dm2 | cur.dm3) == 0ull ? 1 : 0;
        n_stale += cur.nb - n_done;                   // dots computed behind the stop: thrown away
        if (upd) {
            n_upd += nupd;
            const UpdList ul = upd_list(smem);
            if (stage_slices(nupd, [&](int u) __attribute__((always_inline))  { return pos + ul.pos[u]; })) lds_barrier();      // register-home markers: their owners hand the slices over
#pragma unroll 1
            for (int u = 0; u < nupd; u++) {
                const int ps = pos + ul.pos[u];
                const double* uv = ul.val + 4 * u;
                // this thread's ND dwords of the slice: field o_fld of each byte is one of its individuals (device codes)
                const SliceAt sa = slice_at(ps, u);
                const char* own = sa.base + 16 * (o_chunk ^ sa.swz) + o_jb;
#pragma unroll
                for (int d = 0; d < ND; d++) {
                    const uint32_t cd = (*reinterpret_cast<const uint32_t*>(own + 4 * d) | na_or[d]) >> (2 * o_fld);
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) eps[4 * d + bb] += uv[(cd >> (8 * bb)) & 3u];
                }
            }
            refresh_planes();
        }
        pos += n_done;
        n_batch++;
        TRACE(6);
        PROF(6);   // residual update + plane refresh
        PT_FROM_LAST();
        meta_commit();
        if (ctl[C_CURSOR] >= 624) block_advance(s_rng0, s_rng1, ctl, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may be in flight when the workgroup ends

    if (ok) {                                         // a range violation in the last update is seen here
        __syncthreads();
        if (ctl[C_RANGE]) ok = false;
    }
    if (!ok) {
        if (tid == 0) {
            st_u32(abort_word, 1u);
            atomicMax(a.err, ctl[C_RANGE] ? 4 : (ctl[C_RNGERR] ? 2 : 1));
        }
        return;
    }
    if (ctl[C_CURSOR] >= 624) block_advance(s_rng0, s_rng1, ctl, true);
    if (valid) {
#pragma unroll
        for (int q = 0; q < NI; q++) a.eps[4 * (o_byte + (size_t)q) + (size_t)o_fld] = eps[q];
    }
    for (int i = tid; i < G * K; i += SW_TPB)             // component counts: each workgroup counted the positions it wrote
        if (s_cass[i] != 0) atomicAdd(&a.cass[i], s_cass[i]);
    if (wg == 0) {
        for (int i = tid; i < 624; i += SW_TPB) a.rng_state[i] = s_rng0[i];
        if (tid == 0) {
            *a.rng_index = ctl[C_CURSOR];
            a.stats[0] = n_upd; a.stats[1] = n_batch; a.stats[2] = max_nb; a.stats[3] = n_short;
            a.stats[29] = n_planned; a.stats[30] = n_stale; a.stats[31] = n_fastb; a.stats[32] = n_cross;
            a.stats[33] = ctl[C_NSCRT]; a.stats[34] = ctl[C_NSCR];
        }
    }
#ifdef GM_SWEEP_PROF
    if (tid == GM_PROF_TID && (wg == 0 || wg == W / 2)) {
        for (int i = 0; i < 8; i++) a.stats[(wg == 0 ? 4 : 12) + i] = (long long)prof[i];
        if (wg != 0) for (int i = 0; i < 4; i++) a.stats[20 + i] = (long long)reinterpret_cast<unsigned long long*>(smem + L_M + 64)[i];
        if (wg != 0) for (int i = 0; i < 5; i++) a.stats[24 + i] = (long long)pa[i];
        if (wg != 0) for (int i = 0; i < 3; i++) a.stats[35 + i] = (long long)pt[i];
        if (wg != 0) for (int i = 0; i < 2; i++) a.stats[38 + i] = (long long)reinterpret_cast<unsigned long long*>(smem + L_M + 64)[4 + i];
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

template <int R, int MODE, bool CONT, bool LONG> static hipError_t launch_RF(const SweepArgs& a0, hipStream_t st, int grid) {
    const int lds = L_TOTAL;
    SweepArgs a = a0;
    const Carve cv = carve_for<R, CONT, LONG>(a.G, a.K);
    a.lds_cass = cv.cass; a.lds_tab = cv.tab; a.lds_pln = cv.pln; a.lds_ring = cv.ring; a.nl = cv.nl; a.nl_magic = cv.nl_magic; a.win = cv.win;
    if (cv.win < 4) return hipErrorInvalidValue;     // (cannot happen for G <= 64, K <= 8: the tables leave room for several tiles)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<R, MODE, CONT, LONG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sweep<R, MODE, CONT, LONG>), dim3(grid), dim3(SW_TPB), lds, st, a);
    return hipGetLastError();
}
#ifdef GM_ONE_KERNEL
// experiments on code generation (tools/one_kernel.sh): only one k_sweep is instantiated -- not a product build
#if defined(GM_ONE_KERNEL_CONT)
hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st, int grid) { return launch_RF<2, 0, true, true>(a, st, grid); }
#elif defined(GM_ONE_KERNEL_MIXED)
hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st, int grid) { return launch_RF<2, 1, false, true>(a, st, grid); }
#else
hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st, int grid) { return launch_RF<2, 0, false, true>(a, st, grid); }
#endif
hipError_t sweep_occupancy(int, int* blocks_per_cu) { *blocks_per_cu = 1; return hipSuccess; }
#else
template <int R> static hipError_t launch_R(const SweepArgs& a, hipStream_t st, int grid) {
    switch (a.miss_mode) {
        case 0:
            if (a.cross > 0) return a.long_cross ? launch_RF<R, 0, true, true>(a, st, grid) : launch_RF<R, 0, true, false>(a, st, grid);
            return launch_RF<R, 0, false, true>(a, st, grid);
        case 2: return a.cross > 0 ? launch_RF<R, 2, true, false>(a, st, grid) : launch_RF<R, 2, false, false>(a, st, grid);
        default: return a.long_mixed ? launch_RF<R, 1, false, true>(a, st, grid) : launch_RF<R, 1, false, false>(a, st, grid);
    }
}

// `grid` is a.W except in the fault-injection test (one workgroup short: the grid-wide wait must time out).
hipError_t launch_sweep(const SweepArgs& a, int R, hipStream_t st, int grid) {
    if (a.W > SW_TPB || grid < 1 || grid > a.W) return hipErrorInvalidValue;
    switch (R) {
        case 1: return launch_R<1>(a, st, grid);
        case 2: return launch_R<2>(a, st, grid);
        case 4: return launch_R<4>(a, st, grid);
        default: return hipErrorInvalidValue;
    }
}

// Resident workgroups per compute unit for the sweep kernel at this R (the smaller of the two exchange
// layouts' answers): the kernel's workgroups wait for each other, so a launch is only legal when the
// whole grid is resident.  The LDS request (> 80 KiB) makes this 1.
template <int R> static hipError_t occupancy_R(int* out) {
    const int lds = L_TOTAL;
    int n0 = 0, n1 = 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<R, 0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<R, 1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, k_sweep<R, 0, true, false>, SW_TPB, (size_t)lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_sweep<R, 1, false, false>, SW_TPB, (size_t)lds);
    if (e != hipSuccess) return e;
    *out = n0 < n1 ? n0 : n1;
    return hipSuccess;
}
hipError_t sweep_occupancy(int R, int* blocks_per_cu) {
    switch (R) {
        case 1: return occupancy_R<1>(blocks_per_cu);
        case 2: return occupancy_R<2>(blocks_per_cu);
        case 4: return occupancy_R<4>(blocks_per_cu);
        default: return hipErrorInvalidValue;
    }
}
#endif

}  // namespace gm

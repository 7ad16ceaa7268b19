// gm_host.h -- host-side structures behind the opaque handles of include/gmrm_hip.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace gm {

// One phenotype's device-resident chain state (reference: class Phenotype,
// src/phenotype.hpp:12-55) plus the sweep kernel's workspace.
struct Trait {
    double* eps = nullptr;          // Phenotype::epsilon_   [4*stride]
    double* eps_start = nullptr;    // residual at sweep start (multi-GPU exchange)
    uint8_t* namask2 = nullptr;     // Phenotype::mask4 expanded to 2 bits per individual [stride]
    double* mave = nullptr;         // Phenotype::mave [M]
    double* msig = nullptr;         // Phenotype::msig [M]
    uint8_t* nomiss = nullptr;      // [M] marker has no missing genotype among phenotyped individuals
    double* betas[2] = {nullptr, nullptr};   // Phenotype::betas [M], double-buffered per sweep
    int cur = 0;
    int* comp = nullptr;            // Phenotype::comp [M]
    double* acum = nullptr;         // Phenotype::acum [M]
    int nonas = 0;
    bool have_trait = false, have_stats = false, in_flight = false, empty = false;
    long long n_dirty = 0;          // markers of the block with a missing genotype among the phenotyped individuals (gmrm_marker_stats)
    int part_next = 0;              // a sweep in parts (gmrm_sweep_in.first / count): the position the next part starts at (0: a new sweep)
    bool part_last = true;          // the part in flight ends the sweep: its finish makes the new effects current
    bool holds_devlock = false;     // this sweep holds the per-device advisory lock against other processes (capi.cpp)
    bool poisoned = false;          // a sweep failed inside the kernel: comp / acum partly written, sweeps refused until re-upload
    long long in_model = 0;         // markers with a non-zero effect (betas[cur]): from the last sweep's component counts / gmrm_set_betas
    int miss_mode = 2;              // markers of the block with nomiss == 0: 0 none, 1 some, 2 all (or unknown)
    int G = 0, K = 0;
    // sweep workspace
    int* order = nullptr;
    int* o_g = nullptr; double* o_beta = nullptr; double* o_mave = nullptr; double* o_msig = nullptr; uint8_t* o_nm = nullptr;   // sampling inputs in visit order
    double* tab = nullptr;
    uint32_t* rng_state = nullptr;
    int* rng_index = nullptr;
    int* cass = nullptr;
    long long* stats = nullptr;
    int* err = nullptr;
    double* P = nullptr;
    double* Tt = nullptr;
    unsigned* cnt = nullptr;
    double* scratch = nullptr;
    unsigned long long* trace = nullptr;   // diagnostic stamps (GMRM_SWEEP_TRACE)
    hipStream_t stream = nullptr, launch_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

int fail(int code, const std::string& msg);

}  // namespace gm

// One GPU, one contiguous block of markers, T phenotypes (reference: class Bayes,
// src/bayes.hpp:84-105).
struct gmrm_ctx {
    int device = 0, N = 0, M = 0, Mt = 0, S = 0, T = 0;
    size_t mbytes = 0, stride = 0;
    uint8_t* bed = nullptr;         // Bayes::bed_data, column stride padded to 16 bytes
    int* group = nullptr;           // Bayes::group_index[S .. S+M)
    uint8_t* colbuf = nullptr;      // one column of another shard (per-step schedule, gmrm_update_eps_from)
    std::vector<gm::Trait> tr;
    int num_cu = 0, R = 0, W = 0, Wpad = 0, conc = 1;   // conc: chains that sweep side by side
    int max_resident_wg = 0;        // occupancy query x num_cu for the sweep kernel at this R
    int spin_timeout_ms = 4000;     // bound of every grid-wide wait inside the kernel (env GMRM_SPIN_TIMEOUT_MS)
    bool concurrent = true, have_bed = false, have_groups = false;
    double cross_density = 0.01;    // launch the kernel that crosses stops when at least this fraction of the block's markers is in the model (env GMRM_CROSS_DENSITY)
    int cross_frac16 = 9;           // the walk crosses a marker with a non-zero effect when at least this many sixteenths of the batch lie behind it (env GMRM_CROSS_FRAC16)
    int long_cross = 0;                                           // the long-batch kernel that crosses stops (env GMRM_LONG_CROSS: 1 in sparse models, 2 in dense models too; measured slower: DESIGN.md 9)
    int long_cross_frac16 = 5;                                    // ... when at least this many sixteenths of the batch lie behind the marker (env GMRM_LONG_CROSS_FRAC16)
    int batch_cap = 0;                                            // long-batch kernels: longest batch (env GMRM_BATCH_CAP; 0: the kernel's cap, 240)
    int batch_init = 16, nb_factor16 = 24;                        // sweep schedule knobs (env GMRM_NB_FACTOR16)
    int screen_min_run16 = 16 * 48;                               // the sampling screen is tried from this recent run length on (env GMRM_SCREEN_MIN_RUN16)
};

namespace gm {
int ctx_check_t(const gmrm_ctx* c, int t);
}

// gmrm_main.cpp -- command-line host program: gmrm's option surface and file formats on top
// of the C ABI of libgmrm_hip.so (include/gmrm_hip.h).  Everything numerical happens behind
// that ABI; this file is argument parsing, input decoding and output records.
//
// Mirrors, in the build's own code (file:line relative to /root/reference/):
//   src/main.cpp:8-24            driver
//   src/options.cpp:16-286       flags, validation, --group-mixture-file parser
//   src/dimensions.cpp:8-29      --dim-file
//   src/bayes.cpp:830-853        --group-index-file
//   src/bayes.cpp:867-900        .bed block read (here: chunked pread -> gmrm_upload_bed)
//   src/phenotype.cpp:587-621    --phen-files tokenising
//   src/bayes.cpp:659-669, src/xfiles.hpp:14-38, src/xfiles.cpp:6-47   .csv/.bet/.cpn records
//   src/bayes.cpp:16-284         --predict (run_predict below: posterior means from .bet, g = Z beta, .mlma records)
#include "../../include/gmrm_hip.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Opts {
    std::string bed_file, dim_file, bim_file, ref_bim_file, group_index_file, group_mixture_file, out_dir;
    std::vector<std::string> phen_files;
    int verbosity = 0;
    bool shuffle = true, mimic_hydra = false, predict = false;
    unsigned seed = 0, iterations = 1, truncm = 0, thin = 1;
    int device = 0;
    std::vector<int> devices;     // --devices a,b,...: one marker shard per entry (this build only)
    bool no_rccl = false;
    unsigned ckp_every = 0;       // --checkpoint-every n: write <out-dir>/gmrm.<shard>.ckp after every n-th iteration (this build only)
    bool resume = false;          // --resume: continue from those checkpoints instead of starting over (this build only)
    unsigned sync_every = 0;      // --sync-every 1: the reference's per-step exchange (bayes.cpp:495-553); 0 = once per sweep; k > 1: every k markers (this build only)
};

[[noreturn]] void fatal(const std::string& m) {
    std::cout << m << std::endl;
    std::exit(EXIT_FAILURE);
}
void need(int rc, const char* what) {
    if (rc < 0) fatal(std::string("FATAL  : ") + what + ": " + gmrm_last_error());
}
double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string tok;
    while (ss >> tok) out.push_back(tok);
    return out;
}
std::string stem_of(const std::string& path) {               // fs::path::stem()
    size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot != 0) name = name.substr(0, dot);
    return name;
}

// options.cpp:16-160: same flags, same messages
Opts parse(int argc, char** argv) {
    Opts o;
    std::stringstream ss;
    ss << "\nardyh command line options:\n";                  // options.cpp:22 (the program's former name)
    auto last = [&](int i) {
        if (i == argc - 1)
            fatal(std::string("FATAL  : missing argument for last option \"") + argv[i] + "\". Please check your input and relaunch.");
    };
    auto positive = [&](int i, const char* name, int minv) {
        if (atoi(argv[i + 1]) < minv)
            fatal(std::string("FATAL  : option ") + name + " has to be a " + (minv == 0 ? "positive" : "strictly positive")
                  + " integer! (" + argv[i + 1] + " was passed)");
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--bed-file") { last(i); o.bed_file = argv[++i]; ss << "--bed-file " << o.bed_file << "\n"; }
        else if (a == "--dim-file") { last(i); o.dim_file = argv[++i]; ss << "--dim-file " << o.dim_file << "\n"; }
        else if (a == "--phen-files") {
            last(i);
            std::string cslist = argv[++i];
            ss << "--phen-files " << cslist << "\n";
            std::stringstream sl(cslist);
            std::string fp;
            while (getline(sl, fp, ',')) {
                std::ifstream f(fp);
                if (!f.is_open()) fatal("FATAL: file " + fp + " not found");
                o.phen_files.push_back(fp);
            }
        }
        else if (a == "--group-index-file") { last(i); o.group_index_file = argv[++i]; ss << "--group-index-file " << o.group_index_file << "\n"; }
        else if (a == "--group-mixture-file") { last(i); o.group_mixture_file = argv[++i]; ss << "--group-mixture-file " << o.group_mixture_file << "\n"; }
        else if (a == "--verbosity") { last(i); o.verbosity = atoi(argv[++i]); ss << "--verbosity " << o.verbosity << "\n"; }
        else if (a == "--shuffle-markers") { last(i); o.shuffle = (bool)atoi(argv[++i]); ss << "--shuffle-markers " << o.shuffle << "\n"; }
        else if (a == "--mimic-hydra") { o.mimic_hydra = true; ss << "--mimic-hydra " << o.mimic_hydra << "\n"; }
        else if (a == "--seed") { last(i); positive(i, "--seed", 0); o.seed = (unsigned)atoi(argv[++i]); ss << "--seed " << o.seed << "\n"; }
        else if (a == "--iterations") { last(i); positive(i, "--iterations", 1); o.iterations = (unsigned)atoi(argv[++i]); ss << "--iterations " << o.iterations << "\n"; }
        else if (a == "--trunc-markers") { last(i); positive(i, "--trunc-markers", 1); o.truncm = (unsigned)atoi(argv[++i]); ss << "--trunc-markers " << o.truncm << "\n"; }
        else if (a == "--S") { last(i); ++i; ss << "--S " << argv[i] << "\n"; }           // parsed, never used upstream (options.cpp:105-119)
        else if (a == "--out-dir") {
            last(i);
            o.out_dir = argv[++i];
            struct stat st;
            if (stat(o.out_dir.c_str(), &st) != 0) mkdir(o.out_dir.c_str(), 0777);       // single level, options.cpp:123-126
            ss << "--out-dir " << o.out_dir << "\n";
        }
        else if (a == "--output-thin-rate") { last(i); positive(i, "--output-thin-rate", 1); o.thin = (unsigned)atoi(argv[++i]); ss << "--output-thin-rate " << o.thin << "\n"; }
        else if (a == "--predict") { o.predict = true; ss << "--predict " << o.predict << "\n"; }
        else if (a == "--bim-file") { last(i); o.bim_file = argv[++i]; ss << "--bim-file " << o.bim_file << "\n"; }
        else if (a == "--ref-bim-file") { last(i); o.ref_bim_file = argv[++i]; ss << "--ref-bim-file " << o.ref_bim_file << "\n"; }
        else if (a == "--device") { last(i); o.device = atoi(argv[++i]); ss << "--device " << o.device << "\n"; }   // this build only
        else if (a == "--devices") {                                       // this build only: marker shards, one per listed GPU
            last(i);
            std::stringstream sl(argv[++i]);
            std::string d;
            while (getline(sl, d, ',')) o.devices.push_back(atoi(d.c_str()));
            ss << "--devices " << argv[i] << "\n";
        }
        else if (a == "--gpus") {                                          // this build only: --devices 0,1,...,n-1
            last(i);
            const int n = atoi(argv[++i]);
            for (int d = 0; d < n; d++) o.devices.push_back(d);
            ss << "--gpus " << n << "\n";
        }
        else if (a == "--no-rccl") { o.no_rccl = true; ss << "--no-rccl 1\n"; }   // this build only: stage the exchange through the host
        else if (a == "--checkpoint-every") { last(i); positive(i, "--checkpoint-every", 1); o.ckp_every = (unsigned)atoi(argv[++i]); ss << "--checkpoint-every " << o.ckp_every << "\n"; }
        else if (a == "--resume") { o.resume = true; ss << "--resume 1\n"; }
        else if (a == "--sync-every") {
            last(i);
            const int v = atoi(argv[++i]);
            if (v < 0) fatal("FATAL  : --sync-every takes 1 (exchange after every marker step, as upstream), 0 (once per sweep) or k > 1 (every k markers of a shard's block)");
            o.sync_every = (unsigned)v;
            ss << "--sync-every " << o.sync_every << "\n";
        }
        else fatal("FATAL: option \"" + a + "\" unknown");
    }
    std::cout << ss.str() << std::endl;
    // options.cpp:175-220
    if (o.bed_file.empty()) fatal("FATAL  : no bed file provided! Please use the --bed-file option.");
    if (o.dim_file.empty()) fatal("FATAL  : no dim file provided! Please use the --dim-file option.");
    if (o.phen_files.empty()) fatal("FATAL  : no phen file(s) provided! Please use the --phen-files option.");
    if (!o.predict && (o.group_index_file.empty() != o.group_mixture_file.empty()))
        fatal("FATAL  : you need to activate BOTH --group-index-file and --group-mixture-file");
    if (o.mimic_hydra && o.phen_files.size() > 1)
        fatal("FATAL  : with --mimic-hydra, only a single phenotype can be processed.");
    if (o.predict) {                                                       // options.cpp:205-214
        if (o.bim_file.empty()) fatal("FATAL  : you need to pass a bim file with --bim-file when activating --predict");
        if (o.ref_bim_file.empty()) fatal("FATAL  : you need to pass a reference bim file with --ref-bim-file when activating --predict");
    }
    return o;
}

// options.cpp:222-286
void read_mixtures(const std::string& path, std::vector<double>& cva, int& G, int& K) {
    std::ifstream f(path);
    if (!f.is_open()) { printf("FATAL  : can not open the mixture file %s. Use the --group-mixture-file option!\n", path.c_str()); std::exit(1); }
    std::cout << "INFO   : Reading group mixtures from [" + path + "]." << std::endl;
    std::string line;
    G = 0; K = -1;
    while (getline(f, line)) {
        std::vector<std::string> tok = split_ws(line);
        if (tok.empty()) continue;
        if (K < 0) K = (int)tok.size();
        if ((int)tok.size() != K) {
            printf("FATAL  : check your mixture file. The same number of mixtures is expected for all groups.\n");
            printf("       : got %d mixtures for group %d, while first group had %d.\n", (int)tok.size(), G, K);
            std::exit(1);
        }
        for (int j = 0; j < K; j++) {
            const double v = std::stod(tok[j]);
            if (j == 0 && v != 0.0) { printf("FATAL  : First element of group mixture must be 0.0! Check your input file %s.\n", path.c_str()); std::exit(1); }
            if (j > 0 && v <= cva.back()) { printf("FATAL  : Mixtures must be given in ascending order! Check your input file %s.\n", path.c_str()); std::exit(1); }
            cva.push_back(v);
        }
        G++;
    }
}

struct HistFile {                 // xfiles.hpp:14-38 via POSIX pwrite
    int fd = -1;
    void open_fresh(const std::string& p) {
        unlink(p.c_str());                                        // bayes.cpp:323 delete_output_files
        fd = open(p.c_str(), O_CREAT | O_WRONLY | O_EXCL, 0644);  // MPI_MODE_CREATE|WRONLY|EXCL
        if (fd < 0) fatal("FATAL  : cannot create output file " + p);
    }
    void open_existing(const std::string& p) {                    // --resume: records are rewritten in place from the restart point
        fd = open(p.c_str(), O_WRONLY, 0644);
        if (fd < 0) fatal("FATAL  : --resume: cannot open the existing output file " + p);
    }
    void put(const void* buf, size_t n, off_t off) {
        if (pwrite(fd, buf, n, off) != (ssize_t)n) fatal("FATAL  : short write on an output file");
    }
};

// fs::path::replace_extension on the file-name part of p
std::string replace_ext(const std::string& p, const std::string& ext) {
    const size_t slash = p.find_last_of('/');
    const size_t name0 = slash == std::string::npos ? 0 : slash + 1;
    const size_t dot = p.find_last_of('.');
    if (dot != std::string::npos && dot > name0) return p.substr(0, dot) + ext;
    return p + ext;
}

// Bayes::cross_bim_files (bayes.cpp:289-316): row number = marker index
void cross_bim_files(const Opts& opt, std::vector<std::string>& rsid, std::unordered_map<std::string, int>& refrsid) {
    printf("INFO   : bim file:     %s\n", opt.bim_file.c_str());
    printf("INFO   : ref bim file: %s\n", opt.ref_bim_file.c_str());
    std::ifstream in(opt.bim_file);
    if (!in) fatal("Error: can not open the file [" + opt.bim_file + "] to read.");
    std::string id, a1, a2;
    unsigned chr, pos;
    float gpos;
    while (in >> chr >> id >> gpos >> pos >> a1 >> a2) rsid.push_back(id);
    printf("INFO   : found %d ids in bim file\n", (int)rsid.size());
    std::ifstream refin(opt.ref_bim_file);
    if (!refin) fatal("Error: can not open the file [" + opt.ref_bim_file + "] to read.");
    int idx = 0;
    while (refin >> chr >> id >> gpos >> pos >> a1 >> a2) refrsid[id] = idx++;
    printf("INFO   : found %d ids in reference bim file\n", idx);
}

// Bayes::predict (bayes.cpp:16-284) on one GPU = one rank: posterior-mean effects from <stem>.bet,
// g = Z beta (k_predict_g), leave-other-ranks-out correction of y (nothing to leave out with one
// rank, bayes.cpp:141-142), per-marker OLS beta / t / se / p (k_assoc + host arithmetic), <stem>.mlma.
// Bayes::predict (bayes.cpp:16-284) over the marker shards (upstream: MPI tasks): every shard computes the genetic values
// of ITS markers (g_k), the shards' g_k are summed (MPI_Allreduce, bayes.cpp:136), every shard removes the OTHER shards'
// markers from the phenotype (y_k = y - (g - g_k), bayes.cpp:141-142), tests its own markers against that and writes its
// records behind those of the shards before it (bayes.cpp:246-252).
void run_predict(const Opts& opt, const std::vector<gmrm_ctx*>& ctxs, const std::vector<int>& S_, const std::vector<int>& M_, int N, int Mt,
                 const std::vector<std::string>& stems, const std::vector<std::vector<double>>& eps0, const std::vector<int>& nonas) {
    const double ts = now();
    const int nsh = (int)ctxs.size();
    std::vector<std::string> rsid;
    std::unordered_map<std::string, int> refrsid;
    cross_bim_files(opt, rsid, refrsid);
    if ((int)rsid.size() < Mt) fatal("FATAL  : bim file has fewer rows than markers in the dim file.");
    const int T = (int)stems.size();
    for (int t = 0; t < T; t++) {
        for (int r = 0; r < nsh; r++) need(gmrm_marker_stats(ctxs[r], t), "gmrm_marker_stats");
        const std::string base = opt.out_dir.empty() ? stems[t] : opt.out_dir + "/" + stems[t];
        const std::string inbet = replace_ext(base, ".bet"), outmlma = base + ".mlma";   // phenotype.cpp:115-127
        unlink(outmlma.c_str());                                                         // bayes.cpp:33
        const int fb = open(inbet.c_str(), O_RDONLY);
        if (fb < 0) fatal("FATAL  : cannot open " + inbet);
        HistFile fm;
        fm.open_fresh(outmlma);
        struct stat sb;
        fstat(fb, &sb);
        unsigned Mtot_ = 0;
        if (pread(fb, &Mtot_, 4, 0) != 4) fatal("FATAL  : " + inbet + " is empty");
        if (Mtot_ != refrsid.size()) {                                                   // bayes.cpp:48-51
            printf("Mismatch between expected and Mtot read from .bet file: %lu vs %d\n", (unsigned long)rsid.size(), Mtot_);
            std::exit(1);
        }
        const size_t rec = (size_t)Mtot_ * 8 + 4;
        if (((size_t)sb.st_size - 4) % rec != 0) fatal("FATAL  : " + inbet + " is truncated (size is not 4 + k * (4 + 8 * Mtot)).");
        const unsigned niter = (unsigned)(((size_t)sb.st_size - 4) / rec);
        printf("INFO   : Number of recorded iterations in .bet file %d: %u\n", t, niter);
        std::vector<double> beta_sum(Mtot_, 0.0), beta_it(Mtot_);
        for (unsigned i = 0; i < niter; i++) {                                           // bayes.cpp:68-76
            const off_t off = 4 + (off_t)rec * i + 4;
            if (pread(fb, beta_it.data(), (size_t)Mtot_ * 8, off) != (ssize_t)((size_t)Mtot_ * 8)) fatal("FATAL  : short read on " + inbet);
            for (unsigned j = 0; j < Mtot_; j++) beta_sum[j] += beta_it[j];
        }
        for (unsigned j = 0; j < Mtot_; j++) beta_sum[j] /= double(niter);
        close(fb);

        // bayes.cpp:93-122 per shard.  beta_sum is indexed by the row of the CURRENT bim (mglo), as upstream does.
        std::vector<int> rm(Mt, -1);
        std::vector<std::vector<double>> g_k(nsh, std::vector<double>(N));
        for (int r = 0; r < nsh; r++) {
            std::vector<double> beta_local(M_[r], 0.0);
            for (int m = 0; m < M_[r]; m++) {
                const int mglo = S_[r] + m;
                auto f = refrsid.find(rsid[mglo]);
                if (f == refrsid.end()) continue;
                rm[mglo] = f->second;
                if ((unsigned)mglo >= Mtot_) fatal("FATAL  : marker row beyond the .bet file's Mtot (upstream reads out of bounds here).");
                beta_local[m] = beta_sum[mglo];
            }
            need(gmrm_predict_g(ctxs[r], t, beta_local.data(), g_k[r].data()), "gmrm_predict_g");
        }
        std::vector<double> g(N, 0.0);                                                   // MPI_Allreduce(SUM), bayes.cpp:136: shard order
        for (int r = 0; r < nsh; r++)
            for (int i = 0; i < N; i++) g[i] += g_k[r][i];
        const int LLEN = 123 + 1;
        off_t at = 0;                                                                    // records of the shards before this one, bayes.cpp:246-252
        for (int r = 0; r < nsh; r++) {
            const int M = M_[r];
            std::vector<double> y_k(eps0[t].begin(), eps0[t].begin() + N);
            for (int i = 0; i < N; i++) y_k[i] -= (g[i] - g_k[r][i]);                    // bayes.cpp:141-142
            double sigma = 0.0;
            for (int i = 0; i < N; i++) sigma += y_k[i] * y_k[i];
            sigma /= nonas[t];
            std::vector<double> xtx(std::max(1, M)), xty(std::max(1, M));
            need(gmrm_assoc(ctxs[r], t, y_k.data(), xtx.data(), xty.data()), "gmrm_assoc");
            std::vector<char> todump((size_t)LLEN * (size_t)std::max(1, M));
            int n_rem = 0;
            for (int m = 0; m < M; m++) {
                const int mglo = S_[r] + m;
                if (rm[mglo] < 0) { printf("WARNING: marker id %s excluded -- no match\n", rsid[mglo].c_str()); n_rem++; continue; }
                const double beta = xty[m] / xtx[m];                                     // bayes.cpp:198-205
                const double tdist = xty[m] / sqrt(sigma * xtx[m]);
                const double se = beta / tdist;
                const double pval = 1.0 - erf(sqrt(tdist * tdist * 0.5));               // 1 - gamma_p(1/2, t^2/2)
                const int cx = snprintf(&todump[(size_t)(m - n_rem) * (LLEN - 1)], LLEN, "%20s %8d %8d %20.15f %20.15f %20.15f %20.15f\n",
                                        rsid[mglo].c_str(), mglo, rm[mglo], beta, tdist, se, pval);
                if (cx < 0 || cx >= LLEN) fatal("FATAL  : .mlma record longer than 123 bytes (marker id over 20 characters or a value over 4 integer digits).");   // bayes.cpp:235 assert
            }
            if (M - n_rem > 0) fm.put(todump.data(), (size_t)(LLEN - 1) * (size_t)(M - n_rem), at);
            at += (off_t)(LLEN - 1) * (off_t)(M - n_rem);
        }
        close(fm.fd);
    }
    printf("INFO   : Time to compute the predictions: %.2f seconds.\n", now() - ts);
}

}  // namespace

int main(int argc, char** argv) {
    const Opts opt = parse(argc, argv);

    // dimensions.cpp:8-29, dimensions.hpp:10-14
    int N = 0, Mt = 0;
    {
        std::ifstream f(opt.dim_file);
        if (!f.is_open()) fatal("FATAL: could not open dim file: " + opt.dim_file);
        std::string line;
        getline(f, line);
        std::vector<std::string> tok = split_ws(line);
        if (tok.size() != 2) fatal("FATAL: dim file should contain a single line with 2 integers");
        N = atoi(tok[0].c_str());
        Mt = atoi(tok[1].c_str());
        if (opt.truncm > 0 && opt.truncm < (unsigned)Mt) Mt = (int)opt.truncm;
    }
    std::vector<double> cva;
    int G = 0, K = 0;
    if (!opt.predict) {                                                      // options.hpp:12-13
        if (opt.group_mixture_file.empty())
            fatal("FATAL  : can not open the mixture file . Use the --group-mixture-file option!");   // options.cpp:259-261
        read_mixtures(opt.group_mixture_file, cva, G, K);
    }

    // Limits of this build, said out loud (the reference accepts any number of groups and mixtures, options.cpp:222-286, and
    // any N, phenotype.cpp:22): the sampling step keeps the per-group tables in LDS and spreads a marker's component search
    // over one wavefront; the sweep kernel keeps the whole residual on chip, one slice of individuals per compute unit.
    if (!opt.predict && K > GMRM_KMAX)
        fatal("FATAL  : " + std::to_string(K) + " mixture components per group in " + opt.group_mixture_file + ": this build supports at most "
              + std::to_string(GMRM_KMAX) + " (limit of the GPU sampling step; upstream gmrm has none).");
    if (!opt.predict && K < 2)
        fatal("FATAL  : a group mixture needs at least 2 components (0.0 and one variance); found " + std::to_string(K) + ".");
    if (!opt.predict && G > 64)
        fatal("FATAL  : " + std::to_string(G) + " groups in " + opt.group_mixture_file + ": this build supports at most 64 "
              "(limit of the GPU sampling step's on-chip tables; upstream gmrm has none).");
    if (N > 1048576)
        fatal("FATAL  : " + std::to_string(N) + " individuals: this build supports at most 1048576 (256 compute units x 256 threads x 16 "
              "individuals: the residual stays on chip during a sweep; upstream gmrm has no limit).");
    if (gmrm_device_count() < 1) fatal("FATAL  : no HIP device visible; this build has no CPU path.");
    const int T = (int)opt.phen_files.size();
    if (T > 64) fatal("FATAL  : " + std::to_string(T) + " phenotype files: this build supports at most 64 per run.");
    // Marker shards, one per GPU, by the reference's block rule (Bayes::set_block_of_markers,
    // bayes.cpp:903-925): what MPI ranks are upstream.  `ctx` / `smp` below are shard 0.
    std::vector<int> devs = opt.devices.empty() ? std::vector<int>{opt.device} : opt.devices;
    const int nsh = (int)devs.size();
    std::vector<int> S_(nsh), M_(nsh);
    {
        const int modu = Mt % nsh, size = Mt / nsh, Mm = modu != 0 ? size + 1 : size;
        int at = 0;
        for (int r = 0; r < nsh; r++) {
            M_[r] = r < modu ? size + 1 : size; S_[r] = at; at += M_[r];
            printf("INFO   : rank %4d has %d markers over tot Mt = %d, max Mm = %d, starting at S = %d\n", r, M_[r], Mt, Mm, S_[r]);
        }
    }
    std::vector<gmrm_ctx*> ctxs(nsh, nullptr);
    for (int r = 0; r < nsh; r++) need(gmrm_ctx_create(&ctxs[r], devs[r], N, M_[r], Mt, S_[r], T), "gmrm_ctx_create");

    // bayes.cpp:867-900: marker-major block after 3 magic bytes (validated by the library; upstream skips
    // them unchecked).  Parallel chunked pread -> pinned ring -> copy engine, gmrm_amd/csrc/ingest.cpp.
    const size_t mbytes = ((size_t)N + 3) / 4;
    {
        printf("INFO   : rank %4d has allocated %zu bytes (%.3f GB) for raw data.\n", 0, (size_t)Mt * mbytes, double((size_t)Mt * mbytes) / 1.0E9);
        gmrm_ingest_stats st, tot{};
        const unsigned hw = std::thread::hardware_concurrency();
        for (int r = 0; r < nsh; r++) {
            if (gmrm_load_bed_file(ctxs[r], opt.bed_file.c_str(), (size_t)S_[r], (int)std::min(16u, std::max(1u, hw)), &st) != GMRM_OK)
                fatal(std::string("FATAL  : ") + gmrm_last_error());
            tot.bytes += st.bytes; tot.seconds += st.seconds; tot.read_seconds += st.read_seconds; tot.threads = st.threads;
        }
        st = tot;
        printf("INFO   : time to load genotype data = %.2f seconds.\n", st.seconds);
        printf("INFO   : genotype ingest %.3f GB at %.2f GB/s (%d reader threads, %.2f s inside pread).\n",
               double(st.bytes) / 1.0E9, st.seconds > 0 ? double(st.bytes) / st.seconds / 1.0E9 : 0.0, st.threads, st.read_seconds);
    }

    // phenotype.cpp:587-673
    std::vector<std::string> stems;
    std::vector<std::vector<double>> eps0;                                   // predict: the centred, scaled phenotypes
    std::vector<int> nonas_t;
    for (int t = 0; t < T; t++) {
        std::ifstream f(opt.phen_files[t]);
        if (!f.is_open()) fatal("FATAL: could not open phenotype file: " + opt.phen_files[t]);
        std::vector<double> y;
        std::vector<uint8_t> isna;
        std::string line;
        while (getline(f, line)) {
            std::vector<std::string> tok = split_ws(line);
            if (tok.size() < 3) continue;
            if (tok[2] == "NA") { y.push_back(0.0); isna.push_back(1); }
            else { y.push_back(atof(tok[2].c_str())); isna.push_back(0); }
        }
        if ((int)y.size() != N) {                                            // bayes.cpp:856-864
            std::cout << "Fatal: N = " << N << " while phen file " << opt.phen_files[t] << " has " << y.size() << " individuals!" << std::endl;
            std::exit(1);
        }
        std::vector<double> eps(4 * mbytes);
        std::vector<uint8_t> mask4(mbytes);
        int nonas = 0;
        need(gmrm_phen_prepare(y.data(), isna.data(), N, eps.data(), mask4.data(), &nonas), "gmrm_phen_prepare");
        if (N % 4 != 0) std::cout << "Setting last " << 4 - N % 4 << " bits to NAs" << std::endl;
        for (int r = 0; r < nsh; r++) need(gmrm_upload_trait(ctxs[r], t, eps.data(), mask4.data(), nonas), "gmrm_upload_trait");
        printf("INFO   : %s has %d NAs and %d non-NAs.\n", opt.phen_files[t].c_str(), N - nonas, nonas);
        stems.push_back(stem_of(opt.phen_files[t]));
        if (opt.predict) { eps0.push_back(eps); nonas_t.push_back(nonas); }
    }
    printf("INFO   : output directory: %s\n", opt.out_dir.c_str());
    if (opt.predict) {                                                       // main.cpp:15-16, bayes.cpp:794
        const double ts = now();
        run_predict(opt, ctxs, S_, M_, N, Mt, stems, eps0, nonas_t);
        (void)ts;
        for (int r = 0; r < nsh; r++) gmrm_ctx_destroy(ctxs[r]);
        return 0;
    }

    // bayes.cpp:830-853 (only column 2 is used; the upstream check is `group > G`, off by one)
    std::vector<int> group_index;
    {
        std::ifstream f(opt.group_index_file);
        if (!f) fatal("Error: can not open the group file [" + opt.group_index_file + "] to read. Use the --group-index-file option!");
        std::cout << "INFO   : Reading groups from " + opt.group_index_file + "." << std::endl;
        std::string label;
        int group;
        while (f >> label >> group) {
            if (group >= G || group < 0) {
                printf("FATAL  : group index file contains a value that exceeds the number of groups given in group mixture file.\n");
                printf("       : check the consistency between your group index and mixture input files.\n");
                std::exit(1);
            }
            group_index.push_back(group);
        }
        if ((int)group_index.size() < Mt) fatal("FATAL  : group index file has fewer lines than markers.");
        group_index.resize(Mt);
    }

    std::vector<gmrm_sampler*> smps(nsh, nullptr);
    {
        const double ts = now();
        for (int r = 0; r < nsh; r++) {
            gmrm_sampler_opts so{};
            so.seed = opt.seed; so.rank = r; so.nranks = nsh; so.shuffle = opt.shuffle; so.mimic_hydra = opt.mimic_hydra;
            so.G = G; so.K = K; so.cva = cva.data(); so.group_index = group_index.data();
            need(gmrm_sampler_create(&smps[r], ctxs[r], &so), "gmrm_sampler_create");   // computes the markers' statistics
        }
        printf("INFO   : Time to compute the markers' statistics: %.2f seconds.\n", now() - ts);
    }
    for (int r = 0; r < nsh; r++) need(gmrm_sampler_init(smps[r]), "gmrm_sampler_init");
    // Checkpoint / restart (upstream cannot resume: it deletes its outputs at start, bayes.cpp:323)
    auto ckp_path = [&](int r) { return (opt.out_dir.empty() ? std::string("") : opt.out_dir + "/") + "gmrm." + std::to_string(r) + ".ckp"; };
    unsigned it_first = 1;
    if (opt.resume) {
        int it_ck = -1;
        for (int r = 0; r < nsh; r++) {
            int it_r = 0;
            need(gmrm_sampler_load(smps[r], ckp_path(r).c_str(), &it_r), "gmrm_sampler_load");
            if (r > 0 && it_r != it_ck) fatal("FATAL  : --resume: the shards' checkpoints are from different iterations");
            it_ck = it_r;
        }
        it_first = (unsigned)it_ck + 1;
        printf("INFO   : resuming after iteration %d from %s\n", it_ck, ckp_path(0).c_str());
    }
    gmrm_sampler* smp = smps[0];
    gmrm_group* grp = nullptr;
    if (opt.sync_every == 1) {                                               // bayes.cpp:374-553 as upstream runs it
        need(gmrm_group_create(&grp, nsh, ctxs.data(), smps.data(), G, K, 0), "gmrm_group_create");
        printf("INFO   : %d marker shard%s on the reference's per-step schedule: every changed marker is applied to every residual\n"
               "       : replica before the next step (bayes.cpp:495-553, 681-706).  This is the chain of `mpiexec -n %d gmrm`; it costs\n"
               "       : one kernel launch and one device-to-host copy per marker and shard -- use it to validate, not to produce.\n",
               nsh, nsh > 1 ? "s" : "", nsh);
    } else if (nsh > 1) {                                                    // replaces the MPI calls of Bayes::process
        need(gmrm_group_create(&grp, nsh, ctxs.data(), smps.data(), G, K, opt.no_rccl ? 0 : 1), "gmrm_group_create");
        if (opt.sync_every > 1)
            printf("INFO   : %d marker shards, residual exchange every %u markers of a shard's block through %s.\n", nsh, opt.sync_every,
                   gmrm_group_uses_rccl(grp) ? "RCCL (ncclAllReduce)" : "host memory");
        else
        printf("INFO   : %d marker shards, residual exchange once per sweep through %s.\n", nsh,
               gmrm_group_uses_rccl(grp) ? "RCCL (ncclAllReduce)" : "host memory");
        // the sweep-synchronous schedule is an approximation of the sequential scan, not the
        // reference's per-step exchange (bayes.cpp:495-553) -- say so instead of running it silently.
        printf("WARNING: %d shards sweep their blocks against per-shard residual replicas that are reconciled ONCE per sweep (or every\n"
               "       : --sync-every k markers).\n"
               "       : This is not the Markov chain of `mpiexec -n %d gmrm` (exchange after every marker step) nor the 1-shard chain;\n"
               "       : markers in LD that sit in different shards see each other's updates one sweep late.  Use 1 shard for the\n"
               "       : reference's 1-task chain, or --sync-every 1 for its %d-task chain (slow: one exchange per marker step).\n",
               nsh, nsh, nsh);
    }

    // phenotype.cpp:129-143: <out_dir>/<phen stem>.{bet,cpn,csv}
    std::vector<HistFile> fbet(T), fcpn(T), fcsv(T);
    for (int t = 0; t < T; t++) {
        std::string base = opt.out_dir.empty() ? stems[t] : opt.out_dir + "/" + stems[t];
        if (opt.resume) { fbet[t].open_existing(base + ".bet"); fcpn[t].open_existing(base + ".cpn"); fcsv[t].open_existing(base + ".csv"); }
        else { fbet[t].open_fresh(base + ".bet"); fcpn[t].open_fresh(base + ".cpn"); fcsv[t].open_fresh(base + ".csv"); }
    }
    std::vector<double> betas(Mt);
    std::vector<int> comp(Mt);
    std::vector<char> line(50000);                                          // const.hpp:3 LENBUF
    const unsigned Mtot = (unsigned)Mt;
    for (unsigned it = it_first; it <= opt.iterations; it++) {
        const double ts = now();
        printf("\n\n@@@ ITERATION %5d\n", it);
        if (grp && opt.sync_every == 1) need(gmrm_group_iterate_steps(grp, (int)it), "gmrm_group_iterate_steps");
        else if (grp && opt.sync_every > 1) need(gmrm_group_iterate_parts(grp, (int)it, (int)opt.sync_every), "gmrm_group_iterate_parts");
        else if (grp) need(gmrm_group_iterate(grp, (int)it), "gmrm_group_iterate");
        else need(gmrm_sampler_iterate(smp, (int)it), "gmrm_sampler_iterate");
        for (int t = 0; t < T; t++) {
            gmrm_hyper h;
            need(gmrm_sampler_get(smp, t, &h), "gmrm_sampler_get");
            double sg = 0.0;
            for (int g = 0; g < G; g++) sg += h.sigmag[g];
            printf("RESULT : i:%d r:%d p:%d  sum sigmaG = %20.15f  sigmaE = %20.15f\n", it, 0, t, sg, h.sigmae);
        }
        printf("RESULT : It %d  total proc time = %7.3f sec, with sync time = %7.3f\n", it, now() - ts, 0.0);
        if (it % opt.thin == 0) {                                           // bayes.cpp:659-669
            const unsigned nth = it / opt.thin - 1;
            for (int t = 0; t < T; t++) {
                const int n = gmrm_sampler_csv_line(smp, t, (int)it, line.data(), line.size());
                need(n, "gmrm_sampler_csv_line");
                fcsv[t].put(line.data(), (size_t)n, (off_t)nth * n);        // xfiles.cpp:45
                for (int r = 0; r < nsh; r++) {                             // each shard's block at its global offset S
                    need(gmrm_get_betas(ctxs[r], t, betas.data() + S_[r]), "gmrm_get_betas");
                    need(gmrm_get_comp(ctxs[r], t, comp.data() + S_[r]), "gmrm_get_comp");
                }
                if (nth == 0) { fbet[t].put(&Mtot, 4, 0); fcpn[t].put(&Mtot, 4, 0); }
                const off_t ob = 4 + (off_t)nth * (4 + (off_t)Mtot * 8), oc = 4 + (off_t)nth * (4 + (off_t)Mtot * 4);
                fbet[t].put(&it, 4, ob); fbet[t].put(betas.data(), (size_t)Mtot * 8, ob + 4);
                fcpn[t].put(&it, 4, oc); fcpn[t].put(comp.data(), (size_t)Mtot * 4, oc + 4);
            }
        }
        if (opt.ckp_every > 0 && it % opt.ckp_every == 0)
            for (int r = 0; r < nsh; r++) need(gmrm_sampler_save(smps[r], ckp_path(r).c_str(), (int)it), "gmrm_sampler_save");
        fflush(stdout);
    }
    for (int t = 0; t < T; t++) { close(fbet[t].fd); close(fcpn[t].fd); close(fcsv[t].fd); }
    if (grp) gmrm_group_destroy(grp);
    for (int r = 0; r < nsh; r++) { gmrm_sampler_destroy(smps[r]); gmrm_ctx_destroy(ctxs[r]); }
    return 0;
}

// extract_non_zero_beta.cpp -- sparse extraction of the non-zero effects of a .bet history
// (SURVEY 8f-4).  Upstream ships this tool as a binary only (example/extract_non_zero_betaAll; its
// source is not in the checkout); its behaviour was pinned by running it on small .bet files
// (tests/golden/ref_extract.*, tools/make_golden_extract.py):
//
//   extract_non_zero_betaAll <path to .bet> <min record> <max record>
//
// For every saved record r in [min, max] (0-based position in the file, NOT the iteration number
// stored in the record) and every marker m with a non-zero effect, one line
//   printf("%7d %7d %20.12f\n", r, m, beta)
// Record layout: src/xfiles.hpp:14-38 (uint32 Mtot, then per record uint32 iteration + Mtot doubles).
//
// One deliberate difference: the upstream binary does not check its reads and prints the stale buffer
// for records past the end of the file; this one stops at the last complete record.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 4) {
        std::printf("Wrong number of arguments passed: %d; expected 3 (path to .bet file, min iteration, max iteration to convert)!\n", argc - 1);
        return 1;
    }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::printf("Error opening file: %s\n", argv[1]); return 1; }
    const long rmin = std::atol(argv[2]), rmax = std::atol(argv[3]);
    uint32_t M = 0;
    if (std::fread(&M, 4, 1, f) != 1) { std::fclose(f); return 0; }
    std::vector<double> beta(M);
    for (long r = rmin < 0 ? 0 : rmin; r <= rmax; r++) {
        const long long off = 4 + (long long)r * (4 + 8ll * M) + 4;       // skip the record's iteration number
        if (fseeko(f, (off_t)off, SEEK_SET) != 0) break;
        if (std::fread(beta.data(), 8, M, f) != M) break;                  // past the last complete record
        for (uint32_t m = 0; m < M; m++)
            if (beta[m] != 0.0) std::printf("%7d %7d %20.12f\n", (int)r, (int)m, beta[m]);
    }
    std::fclose(f);
    return 0;
}

// oracle/ref_harness/lut_dump.cpp -- TEST INFRASTRUCTURE.
// Our own harness; it #includes the reference's generated table headers IN PLACE
// (-I/root/reference/src, nothing copied) and dumps them as raw little-endian doubles:
//   dotp_lut_a[1024] | dotp_lut_b[1024] | dotp_lut_ab[2048] | na_lut[64]
// (src/dotp_lut.hpp:3,1030,2057; src/na_lut.hpp:3).  tests/golden/ref_luts.bin is
// this program's output; tools/make_golden.py runs it.
#include <cstdio>
#include "dotp_lut.hpp"
#include "na_lut.hpp"

int main(int argc, char** argv) {
    if (argc != 2) { std::fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    std::FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 1;
    std::fwrite(dotp_lut_a, sizeof(double), 1024, f);
    std::fwrite(dotp_lut_b, sizeof(double), 1024, f);
    std::fwrite(dotp_lut_ab, sizeof(double), 2048, f);
    std::fwrite(na_lut, sizeof(double), 64, f);
    std::fclose(f);
    return 0;
}

// oracle/ref_harness/xfiles_harness.cpp -- TEST INFRASTRUCTURE.
// Our own harness around the reference's OWN output writers, compiled in place from
// /root/reference/src/xfiles.cpp + utilities.cpp (oracle/Makefile, target ref_xfiles):
//   write_ofile_csv  (xfiles.cpp:6-47)   -> <out>.csv
//   write_ofile_h1<T>(xfiles.hpp:14-38)  -> <out>.bet (double) and <out>.cpn (int)
// It reads a small text spec (iterations, G, K, Mtot, then per iteration the values)
// so that tests/golden/ref_xfiles_* are records produced by the reference's code from
// inputs the tests also feed to the build's writers.
//
// spec format (whitespace separated):
//   n_it G K Mtot
//   then n_it blocks:  it  sigmaG[G]  sigmaE  m0_sum  pi[G*K]  betas[Mtot]  comp[Mtot]
#include <mpi.h>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "xfiles.hpp"

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    if (argc != 3) { std::fprintf(stderr, "usage: %s spec.txt out_stem\n", argv[0]); MPI_Finalize(); return 2; }
    std::ifstream in(argv[1]);
    const std::string stem = argv[2];
    unsigned n_it; int G, K; unsigned Mtot;
    in >> n_it >> G >> K >> Mtot;
    MPI_File fcsv, fbet, fcpn;
    const int amode = MPI_MODE_CREATE | MPI_MODE_WRONLY;
    MPI_File_open(MPI_COMM_WORLD, (stem + ".csv").c_str(), amode, MPI_INFO_NULL, &fcsv);
    MPI_File_open(MPI_COMM_WORLD, (stem + ".bet").c_str(), amode, MPI_INFO_NULL, &fbet);
    MPI_File_open(MPI_COMM_WORLD, (stem + ".cpn").c_str(), amode, MPI_INFO_NULL, &fcpn);
    for (unsigned n = 0; n < n_it; n++) {
        unsigned it; in >> it;
        std::vector<double> sigmag(G);
        for (auto& v : sigmag) in >> v;
        double sigmae; int m0_sum; in >> sigmae >> m0_sum;
        std::vector<std::vector<double>> pi(G, std::vector<double>(K));
        for (auto& row : pi) for (auto& v : row) in >> v;
        std::vector<double> betas(Mtot);
        for (auto& v : betas) in >> v;
        std::vector<int> comp(Mtot);
        for (auto& v : comp) in >> v;
        // bayes.cpp:659-669 with rank 0, S = 0, M = Mtot
        write_ofile_csv(fcsv, it, &sigmag, sigmae, m0_sum, n, &pi);
        write_ofile_h1(fbet, 0, Mtot, it, n, 0, Mtot, betas.data(), MPI_DOUBLE);
        write_ofile_h1(fcpn, 0, Mtot, it, n, 0, Mtot, comp.data(), MPI_INTEGER);
    }
    MPI_File_close(&fcsv); MPI_File_close(&fbet); MPI_File_close(&fcpn);
    MPI_Finalize();
    return 0;
}

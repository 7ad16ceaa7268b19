// oracle/ref_harness/stl_shuffle.cpp -- TEST INFRASTRUCTURE.
// Pins the two pieces of the marker shuffle that this image CAN pin: the reference shuffles
// Phenotype::midx with boost::range::random_shuffle(midx, generator) (src/phenotype.cpp:314-323), which is
// std::random_shuffle(first, last, rand) of the C++ standard library, driven by a boost::mt19937 -- the same
// engine as std::mt19937 by specification.  This harness runs libstdc++'s OWN std::random_shuffle and
// std::mt19937 (real library code of this image, compiled as C++14: the function left the standard in C++17)
// with the oracle's restatement of Boost's uniform_int rule as the generator (bucket rejection on one 32-bit
// output; Boost itself is absent), and prints the permutations.  tests/golden/stl_shuffle.txt.gz is its output;
// orc_rng_shuffle and the product's gm::shuffle must reproduce it.
//   usage: stl_shuffle n seed [n seed ...]     prints: "n seed : p0 p1 ... p(n-1)" per pair
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct BoostUniformIntRule {           // boost/random/uniform_int_distribution.hpp generate_uniform_int, range < 2^32 - 1
    std::mt19937& eng;
    std::ptrdiff_t operator()(std::ptrdiff_t n) {      // in [0, n-1], as variate_generator< mt19937&, uniform_int<> >(n)
        const uint32_t range = (uint32_t)n - 1u;
        if (range == 0) return 0;
        uint32_t bucket = 0xFFFFFFFFu / (range + 1u);
        if (0xFFFFFFFFu % (range + 1u) == range) ++bucket;
        for (;;) {
            const uint32_t r = (uint32_t)eng() / bucket;
            if (r <= range) return (std::ptrdiff_t)r;
        }
    }
};

int main(int argc, char** argv) {
    for (int a = 1; a + 1 < argc; a += 2) {
        const int n = std::atoi(argv[a]);
        const uint32_t seed = (uint32_t)std::strtoul(argv[a + 1], nullptr, 10);
        std::mt19937 eng(seed);
        std::vector<int> v(n);
        for (int i = 0; i < n; i++) v[i] = i;
        BoostUniformIntRule gen{eng};
        std::random_shuffle(v.begin(), v.end(), gen);
        std::printf("%d %u :", n, seed);
        for (int i = 0; i < n; i++) std::printf(" %d", v[i]);
        std::printf("\n");
    }
    return 0;
}

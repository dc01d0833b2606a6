// CPU test of sg_kmix / sg_kunmix (alaz_amd/csrc/sg_hash.h): the key mix of the narrow-record K1 path must be a BIJECTION of the
// 2 nb-bit endpoint pair (pass B recovers (from, to) from partition number + remainder) and must spread structured keys —
// consecutive pod ids x a Zipf head of service ids — evenly over the partitions.
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>
#include "../../alaz_amd/csrc/sg_hash.h"

int main() {
    // exhaustive at nb = 10: every output pair is hit exactly once
    {
        const uint32_t nb = 10, m = (1u << nb) - 1;
        std::vector<uint8_t> seen(1u << (2 * nb), 0);
        for (uint32_t f = 0; f <= m; f++) for (uint32_t t = 0; t <= m; t++) {
            uint32_t L, R; sg_kmix(f, t, m, &L, &R);
            if (L > m || R > m) { std::printf("out of range\n"); return 1; }
            if (seen[(L << nb) | R]++) { std::printf("collision at %u %u\n", f, t); return 1; }
            uint32_t a, b; sg_kunmix(L, R, m, &a, &b);
            if (a != f || b != t) { std::printf("unmix mismatch\n"); return 1; }
        }
    }
    std::mt19937 rng(5);
    for (uint32_t nb : {12u, 14u, 18u, 21u, 24u}) {
        const uint32_t m = (1u << nb) - 1;
        for (int k = 0; k < 2000000; k++) {
            const uint32_t f = rng() & m, t = rng() & m;
            uint32_t L, R, a, b; sg_kmix(f, t, m, &L, &R); sg_kunmix(L, R, m, &a, &b);
            if (a != f || b != t || L > m || R > m) { std::printf("nb %u: round trip failed\n", nb); return 1; }
        }
    }
    // balance: 10 000 sources x the 100 most popular of 5 000 destinations (dense id ranges), 512 partitions
    {
        const uint32_t nb = 14, m = (1u << nb) - 1, pb = 9;
        std::vector<uint32_t> cnt(1u << pb, 0);
        for (uint32_t f = 0; f < 10000; f++) for (uint32_t t = 10000; t < 10100; t++) { uint32_t L, R; sg_kmix(f, t, m, &L, &R); cnt[L >> (nb - pb)]++; }
        const double mean = 1e6 / 512.0; double var = 0; uint32_t mx = 0;
        for (uint32_t c : cnt) { var += (c - mean) * (c - mean); mx = c > mx ? c : mx; }
        const double sigma = std::sqrt(var / 512.0);
        std::printf("balance: mean %.1f sigma %.1f max %u\n", mean, sigma, mx);
        if (sigma > 3.0 * std::sqrt(mean) || mx > 1.25 * mean) { std::printf("unbalanced\n"); return 1; }
    }
    std::printf("ok kmix\n");
    return 0;
}

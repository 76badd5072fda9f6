// tests/median_net_check.cpp -- host check of lmono_amd/csrc/median_net.hpp (built and run by tests/test_median_net.py)
#include "../lmono_amd/csrc/median_net.hpp"
#include <algorithm>
#include <cstdio>
int main()
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad)
    for (long x = 0; x < (1L << 25); x++) {          // every binary input: the median is 1 iff 13 or more inputs are 1
        int v[25], ones = 0;
        for (int i = 0; i < 25; i++) { v[i] = (int)((x >> i) & 1); ones += v[i]; }
        if (mednet::median25(v) != (ones >= 13 ? 1 : 0)) bad++;
    }
    unsigned s = 1;
    long badr = 0;
    for (int it = 0; it < 300000; it++) {            // and random byte windows against std::sort
        int v[25], w[25];
        for (int i = 0; i < 25; i++) { s = s * 1664525u + 1013904223u; v[i] = w[i] = (int)((s >> 10) % (it % 3 == 0 ? 4u : 256u)); }
        std::sort(w, w + 25);
        if (mednet::median25(v) != w[12]) badr++;
    }
    std::printf("BINARY_FAILURES %ld RANDOM_FAILURES %ld\n", bad, badr);
    return bad || badr ? 1 : 0;
}

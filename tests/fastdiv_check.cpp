// Host check of qk_common.h make_fastdiv / fast_div (the row decode of the fp32 backward-weight kernel): floor(n / d) for 0 <= n < 2^31,
// every d in 1..5000 around multiples and at the ends, 2 M random (d, n) pairs, powers of two +- 1.  Built and run by tests/test_api_cpu.py.
#include "qk_common.h"
#include <cstdio>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    unsigned long long bad = 0, n_checked = 0;
    auto check = [&](unsigned d, unsigned n) {
        unsigned mul, sh; qk::make_fastdiv(d, &mul, &sh);
        ++n_checked;
        if ((unsigned)qk::fast_div((int)n, mul, sh) != n / d) { if (bad++ < 5) printf("bad n=%u d=%u\n", n, d); }
    };
    const unsigned nmax = 0x7fffffffu;
    for (unsigned d = 1; d <= 5000; ++d) {
        for (unsigned k = 0; k < 64; ++k) { unsigned long long q = rng() % (nmax / d + 1); for (int e = -1; e <= 1; ++e) { long long n = (long long)q * d + e; if (n >= 0 && n <= nmax) check(d, (unsigned)n); } }
        check(d, nmax); check(d, nmax - 1); check(d, 0); check(d, d - 1); check(d, d);
    }
    for (int i = 0; i < 2000000; ++i) { unsigned d = (unsigned)(rng() % nmax) + 1; unsigned n = (unsigned)(rng() % ((unsigned long long)nmax + 1)); check(d, n); unsigned long long q = nmax / d; check(d, (unsigned)(q * d)); if (q * d >= 1) check(d, (unsigned)(q * d - 1)); }
    for (unsigned s = 0; s < 31; ++s) for (int e = -1; e <= 1; ++e) { long long d = (1ll << s) + e; if (d >= 1) for (int f = -2; f <= 2; ++f) for (unsigned k = 1; k < 40; ++k) { long long n = d * k + f; if (n >= 0 && n <= nmax) check((unsigned)d, (unsigned)n); } check((unsigned)(d >= 1 ? d : 1), nmax); }
    printf("checked %llu bad %llu\n", n_checked, bad);
    return bad ? 1 : 0;
}

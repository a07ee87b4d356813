/* Test infrastructure (see oracle/__init__.py): checks on the CPU that the reciprocal-multiply
 * quotient the register DTW kernel uses (subgnn_amd/csrc/similarity.hip: dtw_cost_rcp) is the
 * correctly rounded IEEE-754 quotient -- i.e. bit-identical to the `/` of the reference's
 * gamma.calc_dist (SubGNN/gamma.py:51-52) -- over the operands that kernel can meet: value + 1 for
 * degrees and their pairwise averages (dyadic rationals).
 *   q0 = RN(mx * r); rem = fma(-q0, mn, mx); q = fma(rem, r, q0)   with r = RN(1 / mn)
 * The kernel forms BOTH quotients this way, a / b and b / a, and takes the larger: checked as well -- the larger one
 * must be the IEEE quotient max / min (the smaller only has to stay <= 1).
 * usage: division_check <max_int> <n_random>     prints "bad <count>" */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static inline double rcp_div(double mx, double mn, double r)
{
    const double q0 = mx * r;
    const double rem = fma(-q0, mn, mx);
    return fma(rem, r, q0);
}

int main(int argc, char** argv)
{
    const int max_int = argc > 1 ? atoi(argv[1]) : 3000;
    const long n_random = argc > 2 ? atol(argv[2]) : 10000000L;
    long bad = 0, n = 0;
    for (int b = 1; b <= max_int; ++b) {
        const double db = b, r = 1.0 / db;
        for (int a = b; a <= max_int; ++a, ++n) {
            const double da = a;
            if (rcp_div(da, db, r) != da / db) ++bad;
            if (fmax(rcp_div(da, db, r), rcp_div(db, da, 1.0 / da)) != da / db) ++bad;
        }
    }
    uint64_t s = 88172645463325252ull;
    for (long i = 0; i < n_random; ++i, ++n) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t x = (uint32_t)s & 0x3fffff, y = (uint32_t)(s >> 32) & 0x3fffff;      /* up to 2^22 */
        const int L = (int)((s >> 59) & 7);                                                  /* halved up to 7 times */
        double a = (double)(x + 1) / (double)(1 << L), b = (double)(y + 1) / (double)(1 << L);
        if (a < b) { const double t = a; a = b; b = t; }
        if (rcp_div(a, b, 1.0 / b) != a / b) ++bad;
        if (fmax(rcp_div(a, b, 1.0 / b), rcp_div(b, a, 1.0 / a)) != a / b) ++bad;
    }
    printf("checked %ld\nbad %ld\n", n, bad);
    return bad != 0;
}

/* check_exp_restatement.c -- TEST INFRASTRUCTURE ONLY (oracle/): the proof behind libm_exp.
 *
 * The device-resident simplex (verifybamid_amd/csrc/resident_kernel.inc) computes
 * FullLLKFunc::InvLogit (ContaminationEstimator.h:119-122: e = exp(x); e / (1 + e)) on the GPU and
 * promises the SAME alpha, bit for bit, as the reference's host code gets from libm's exp() on an
 * FMA-capable x86-64 (glibc >= 2.28, the __exp_fma ifunc variant).  `libm_exp` there restates that
 * algorithm; this program is the same statement sequence on the host -- the table is the very
 * text the kernel includes (libm_exp_table.inc), every fma below is a hardware fma (-mfma,
 * -ffp-contract=off: no other contraction) exactly where the device routine has one -- compared
 * bit for bit with the libm this process links, over
 *   * N uniform arguments in (-512, 512) and N log-uniform magnitudes 2^-54 .. 2^9 of both signs,
 *   * N arguments in the logit range the search actually visits, (-40, 40),
 *   * the neighbourhoods (+-64 ulps) of every edge: +-512, +-2^-54, 0, the rounding boundaries
 *     (j + 1/2) ln2/128 of the table index for 4096 values of j, and small integers.
 * Outside 2^-54 <= |x| < 512 the device does not use the routine at all (resident_kernel.inc:
 * unpack_rows returns 1 + x below 2^-54, as libm does, and hands |x| >= 512 and NaN to the host);
 * the two branches are part of what is checked here.
 *
 * Usage: check_exp_restatement [N = 4000000] [seed = 1]     exit 0 = all equal, 1 = mismatch,
 *        77 = this CPU has no FMA (libm then runs its other variant: nothing to compare).
 * Built by `make -C oracle check_exp`; run by tests/test_oracle_golden.py (N = 4e6 -> 1.2e7
 * arguments + edges) -- `check_exp_restatement 50000000` is the 1.5e8-argument version of the
 * round-2 one-off.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const unsigned long long kTab[256] = {
#include "../verifybamid_amd/csrc/libm_exp_table.inc"
};

static inline double as_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t as_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* resident_kernel.inc: libm_exp, statement for statement */
static double device_exp_core(double x)
{
    const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8,
                 NegLn2loN = -0x1.cf79abc9e3b3ap-47, Shift = 0x1.8p52;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
                 C5 = 0x1.1111167a4d017p-7;
    double kd = __builtin_fma(InvLn2N, x, Shift);
    const uint64_t ki = as_bits(kd);
    kd -= Shift;
    const double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const double tail = as_double(kTab[idx]);
    const uint64_t sbits = kTab[idx + 1] + (ki << 45);
    const double r2 = r * r;
    const double tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C5, C4),
                                     __builtin_fma(r2, __builtin_fma(r, C3, C2), tail + r));
    const double scale = as_double(sbits);
    return __builtin_fma(scale, tmp, scale);
}

/* flatten_kernels.hip: libm_exp_any -- the routine above plus libm's special cases (tiny, huge, NaN/inf, and the
 * results near the subnormal range that libm's `specialcase` rounds once), statement for statement: the device flatten
 * computes a marker's alpha-free terms exp(c_other + D[g]) (context.cpp pass A) for sums that reach -700 and below on
 * markers of a thousand reads, and promises the host flatten's bytes. */
static double device_exp_any(double x)
{
    const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8,
                 NegLn2loN = -0x1.cf79abc9e3b3ap-47, Shift = 0x1.8p52;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
                 C5 = 0x1.1111167a4d017p-7;
    const uint64_t xb = as_bits(x);
    const unsigned abstop = (unsigned)(xb >> 52) & 0x7ffu;
    int special = 0;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {
        if (abstop < 0x3c9u) return 1.0 + x;                      /* |x| < 2^-54 */
        if (abstop >= 0x409u) {                                   /* |x| >= 1024, inf, NaN */
            if (xb == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (xb >> 63) ? 0.0 : INFINITY;
        }
        special = 1;
    }
    double kd = __builtin_fma(InvLn2N, x, Shift);
    const uint64_t ki = as_bits(kd);
    kd -= Shift;
    const double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const double tail = as_double(kTab[idx]);
    uint64_t sbits = kTab[idx + 1] + (ki << 45);
    const double r2 = r * r;
    const double tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C5, C4),
                                     __builtin_fma(r2, __builtin_fma(r, C3, C2), tail + r));
    if (!special) {
        const double scale = as_double(sbits);
        return __builtin_fma(scale, tmp, scale);
    }
    if ((ki & 0x80000000ull) == 0) {                              /* k > 0: the exponent of scale may have overflowed */
        sbits -= 1009ull << 52;
        const double scale = as_double(sbits);
        return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                                       /* k < 0: take care in the subnormal range */
    const double scale = as_double(sbits);
    double y = scale + scale * tmp;
    if (y < 1.0) {
        double lo = scale - y + scale * tmp;
        const double hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;                                    /* (no -0 in round-to-nearest) */
    }
    return 0x1p-1022 * y;
}

/* resident_kernel.inc: DeviceSimplex::unpack_rows' use of it (status 3 = "the host's job") */
static int device_exp(double x, double* out)
{
    const double ax = fabs(x);
    if (!(ax < 512.0)) return 3;
    *out = ax < 0x1p-54 ? 1.0 + x : device_exp_core(x);
    return 0;
}

static uint64_t rng_next(uint64_t* s)          /* splitmix64 */
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_unit(uint64_t* s) { return (double)(rng_next(s) >> 11) * 0x1p-53; }

static long g_checked = 0, g_bad = 0, g_routed = 0;
static void check(double x)
{
    double got;
    if (device_exp(x, &got)) { ++g_routed; return; }
    const double want = exp(x);
    ++g_checked;
    if (as_bits(got) != as_bits(want)) {
        if (g_bad < 10) fprintf(stderr, "MISMATCH x=%a device=%a libm=%a\n", x, got, want);
        ++g_bad;
    }
}
static long g_checked_any = 0;
static void check_any(double x)
{
    const double got = device_exp_any(x), want = exp(x);
    ++g_checked_any;
    if (as_bits(got) != as_bits(want) && !(got != got && want != want)) {
        if (g_bad < 10) fprintf(stderr, "MISMATCH (any) x=%a device=%a libm=%a\n", x, got, want);
        ++g_bad;
    }
}
static void check_around(double x)
{
    double lo = x, hi = x;
    check(x);
    check_any(x);
    for (int i = 0; i < 64; ++i) {
        lo = nextafter(lo, -INFINITY);
        hi = nextafter(hi, INFINITY);
        check_any(lo);
        check_any(hi);
    }
    lo = hi = x;
    for (int i = 0; i < 64; ++i) {
        lo = nextafter(lo, -INFINITY);
        hi = nextafter(hi, INFINITY);
        check(lo);
        check(hi);
    }
}

int main(int argc, char** argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 4000000;
    uint64_t seed = argc > 2 ? (uint64_t)atoll(argv[2]) : 1;
    if (!__builtin_cpu_supports("fma")) {
        printf("no FMA on this CPU: libm runs its non-FMA exp variant, nothing to compare\n");
        return 77;
    }
    for (long i = 0; i < N; ++i) check((rng_unit(&seed) * 2 - 1) * 512.0);
    for (long i = 0; i < N; ++i) {
        const double mag = ldexp(1.0 + rng_unit(&seed), (int)(rng_next(&seed) % 63) - 54);   /* 2^-54 .. 2^9 */
        check((rng_next(&seed) & 1) ? mag : -mag);
    }
    for (long i = 0; i < N; ++i) check((rng_unit(&seed) * 2 - 1) * 40.0);
    /* the full-range routine: the flatten's arguments are sums of logarithms of probabilities, (-inf, 0] */
    for (long i = 0; i < N; ++i) check_any(-rng_unit(&seed) * 800.0);
    for (long i = 0; i < N / 2; ++i) check_any(-700.0 - rng_unit(&seed) * 50.0);      /* results around and below DBL_MIN */
    for (long i = 0; i < N / 4; ++i) check_any(500.0 + rng_unit(&seed) * 230.0);
    for (long i = 0; i < N / 4; ++i) {
        const double mag = ldexp(1.0 + rng_unit(&seed), (int)(rng_next(&seed) % 80) - 60);    /* 2^-60 .. 2^20 */
        check_any((rng_next(&seed) & 1) ? mag : -mag);
    }
    const double edges[] = {1024.0, -1024.0, -0x1.6232bdd7abcd2p+9 /* log(DBL_MIN) */, -0x1.74385446d71c3p+9 /* -> 0 */,
                            -0x1.74910d52d3051p+9 /* log(2^-1075) */, 0x1.62e42fefa39efp+9 /* overflow */, -708.0, -745.0, 709.0,
                            512.0, -512.0, 0x1p-54, -0x1p-54, 0.0, 1.0, -1.0, 0.5, -0.5,
                            0x1.62e42fefa39efp-1 /* ln 2 */, -0x1.62e42fefa39efp-1, 511.9999, -511.9999,
                            -0x1.c7ede1c2b1f12p+1 /* logit(0.03), the search's start */};
    for (size_t i = 0; i < sizeof(edges) / sizeof(edges[0]); ++i) check_around(edges[i]);
    for (int j = -2048; j < 2048; ++j) check_around(((double)j * 16 + 0.5) * (0x1.62e42fefa39efp-1 / 128));
    for (int j = -511; j <= 511; ++j) check((double)j);
    check(NAN);
    check(INFINITY);
    check(-INFINITY);
    check_any(NAN);
    check_any(INFINITY);
    check_any(-INFINITY);
    check_any(-0.0);
    check_any(-1e300);
    check_any(1e300);
    printf("checked %ld arguments against this process's libm exp(): %ld mismatches; %ld routed to the host "
           "(|x| >= 512 or NaN); the full-range routine on %ld more\n", g_checked, g_bad, g_routed, g_checked_any);
    return g_bad ? 1 : 0;
}

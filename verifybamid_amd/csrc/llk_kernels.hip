// llk_kernels.hip -- gfx950 kernels for the genotype-mixture log-likelihood
// (reference: FullLLKFunc::ComputeMixLLKs, ContaminationEstimator.h:194-314).
//
// Work decomposition:
//   * a wave = 16 markers x 4 candidate slots: lane (m, g) walks marker m's reads for the BTL
//     candidate points of slot g (MODE 2: 8 points per group); 4-point launches use two
//     16-marker micro-tiles x 2 slots x 2 points instead (MODE 3), 1- and 2-point launches four (MODE 4, 5).  Markers are sorted by depth
//     at context creation and grouped in 16-marker micro-tiles so all lanes of a wave run about
//     the same number of steps;
//   * a marker's reads are run-length coded over the (class x quality) dictionary: one 32-bit run
//     word per distinct code (low half = LDS byte offset of the code's table row, high half = the
//     top 16 bits of double(count), count <= 31), two per uint2, stored [micro-tile][step/2][marker]:
//     a wave load is one contiguous 128-byte row, rows are prefetched 8 deep in cohort steps (HBM), 2-4 deep otherwise;
//   * the per-alpha log-likelihood table (h:213-229) is rebuilt per launch in LDS,
//     restricted to the codes that occur in the data and to the six OFF-diagonal
//     genotype pairs: the diagonal (g1==g2) and the "other base" class do not
//     depend on alpha or the PCs and were summed once at context creation;
//   * each lane keeps 6 FP64 accumulators per candidate point in registers and
//     gathers its table row from LDS with ds_read_b128 (row = 6 values per point);
//   * epilogue per lane: UD*PC projection (h:251-267), HWE priors (h:186-192),
//     9-term exp-sum with the reference's `> 0` test (h:307-311); the marker likelihoods are
//     multiplied as (mantissa, exponent) and the log is taken once per (workgroup, point);
//   * persistent launch: one 1024-thread workgroup per CU (table built once per workgroup);
//     depth-sorted micro-tiles are dealt round-robin across workgroups, so every CU sees the
//     same depth mix, and pulled by the waves of a workgroup through an LDS work queue;
//   * deterministic reduction: 16-lane butterfly -> work-item slots -> per-workgroup partial;
//     the partials cross workgroups inside the launch, either collected by workgroup 0 from
//     tagged, self-validating sets (<= 16 points, cohorts, resident search) or summed by the last
//     workgroup to arrive at an agent-scope ticket (bigger batches), always in the same fixed
//     order (or, two-kernel mode, by llk_finalize_kernel);
//   * llk_resident_kernel keeps the same body on the CUs for a whole Nelder-Mead search and
//     takes its batches from a mailbox in mapped host memory.
//
// Two layouts of a sample (DeviceLayout::pd; Context::create decides):
//   * run words / sums of logarithms, as above: any depth;
//   * probability domain (round 6; template parameter PD): the table holds P(read | genotype pair, alpha)^n per quality,
//     n = 1 .. K, class ref only -- class alt is class ref with the genotypes mirrored (h:164-177) --, a marker's list is one
//     16-bit row offset per STEP (a run of count c = ceil(c / K) steps; ref steps, then alt steps), the six sums are six
//     PRODUCTS of table rows, and the epilogue needs no exponential: (6 exp + 6 log-table rows) per marker and point less,
//     at rounding-level differences.  Taken when every marker's likelihood is bound to stay far from the smallest doubles.
#include "llk_kernels.h"

#include <hip/hip_runtime.h>
#include <link.h>

#include "kernel_debug.h"
#include "tile_sched.h"
#include "tunables.h"
#ifndef VB2_SIMD_DEAL
#define VB2_SIMD_DEAL 1     // (0: wave w takes item w -- the A/B of the SIMD-balanced first deal, see eval_body)
#endif

#include <algorithm>
#include <atomic>
#include <cstring>
#include <type_traits>
#include <vector>

namespace vb2 {

// off-diagonal genotype pairs, in the reference's (g1 outer, g2 inner) order
__device__ __forceinline__ void pair_of(int p, int& g1, int& g2)
{
    // p: 0:(0,1) 1:(0,2) 2:(1,0) 3:(1,2) 4:(2,0) 5:(2,1)
    g1 = p >> 1;
    const int lo = p & 1;
    g2 = lo + (lo >= g1 ? 1 : 0);
}

// v[i] + v[i ^ 32], then ^ 16, 8, 4, 2, 1: the butterfly every final sum of partial LLKs goes through (the order is part of the
// result's bits).  Round 4: without ds_bpermute -- gfx950's v_permlane32_swap / v_permlane16_swap bring the other half / the
// other row next to a lane's own value (x + y = y + x: which of the two registers holds the lane's own does not matter), the
// distances 8 .. 1 are DPP moves inside a row.  The same additions in the same order: the same bits; a search round waits for
// this once (workgroup 0's four sums), a cross-lane LDS round trip per step before.
__device__ __forceinline__ double wave_sum(double v)
{
    {
        const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    {
        const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    auto dpp = [](double x, auto ctrl) -> double {
        constexpr int kCtrl = decltype(ctrl)::value;
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), kCtrl, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(0, __double2loint(x), kCtrl, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0x128>());                       // row_ror:8   -> lane i ^ 8
    {                                                                        // lane i ^ 4: quads 1, 3 <- i - 4, quads 0, 2 <- i + 4
        const int hi = __double2hiint(v), lo = __double2loint(v);
        const int h1 = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xa, false), l1 = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xa, false);
        const int h2 = __builtin_amdgcn_update_dpp(h1, hi, 0x12C, 0xf, 0x5, false), l2 = __builtin_amdgcn_update_dpp(l1, lo, 0x12C, 0xf, 0x5, false);
        v += __hiloint2double(h2, l2);
    }
    v += dpp(v, std::integral_constant<int, 0x4E>());                        // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0xB1>());                        // quad_perm [1,0,3,2]
    return v;
}

// 2^(j/64), j = 0..63, correctly rounded (generated with 60-digit decimal arithmetic).
__device__ const double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};

// exp(x) for x <= 0 (arguments here are sums of log-probabilities), FP64, <= 1.1 ulp:
// x = k*ln2/64 + r, |r| <= ln2/128; exp(x) = 2^(k>>6) * 2^((k&63)/64) * (1 + p(r)), p = r + r^2 q(r), q of degree 3, the
// 2^(j/64) factor from an LDS table.
// `etab_lane` = LDS byte address of this lane's column of the bank-replicated table: entry j sits
// 256*j bytes further, i.e. in LDS banks 2*(lane%32), 2*(lane%32)+1 whatever j is, so the
// 64 lanes' lookups of a wave never conflict (a 512-byte table would: random j's collide).
// v_max_f64 / v_min_f64 as single instructions: the compiler's fmax/fmin lowering first
// canonicalises an operand it cannot prove free of signalling NaNs (one more v_max_f64 x, x);
// the hardware instruction quiets them by itself.
//
// Two forms.  The one every marker x point takes six times applies 2^(k>>6) to the TABLE ENTRY before the last
// multiply-add, by ONE integer add on the entry's high word: the table holds the entries with j << 14 taken off their high
// words (exp_tab_entry), so that adding k << 14 = (k>>6) << 20 + j << 14 leaves exactly (k>>6) in the exponent field -- the
// scaling is exact and the result the same bits as scaling afterwards; the shift and the v_ldexp_f64 of the other form
// are gone, and so is the clamp (three of an exponential's sixteen instructions for one).  It holds as long
// as the scaled entry is a normal number, i.e. for x >= -708 (2^-1022 <= e^-708.39): the caller (marker_lk) takes the
// minimum of a marker's six arguments -- five v_min_f64, integer-rate -- and sends markers with an argument below that
// (deep pileups; -inf: a table entry that is log 0) through the EXACT form: clamp at -800, v_ldexp_f64, gradual underflow,
// 0 exactly where the reference's exp() reaches it (the `markerLK > 0` test, h:310).  Either way the same bits as before.
__device__ __forceinline__ double vmax_f64(double x, double c)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(c));
    return r;
}
__device__ __forceinline__ double vmin_f64(double x, double c)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(c));
    return r;
}
__device__ __forceinline__ double vmin2_f64(double x, double y)      // (both operands in vector registers)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

typedef __attribute__((address_space(3))) const double lds_cdouble;
typedef double __attribute__((ext_vector_type(2))) vdouble2;      // (a plain vector: loadable through address_space(3))
typedef __attribute__((address_space(3))) const vdouble2 lds_cdouble2;
__device__ __forceinline__ uint32_t lds_byte_addr(const void* p)      // generic pointer into LDS -> LDS address
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// what the LDS copy of the table holds for 2^(j/64): the double with j << 14 taken off its high word
__device__ __forceinline__ double exp_tab_entry(double t, int j)
{
    return __hiloint2double(__double2hiint(t) - (j << 14), __double2loint(t));
}

// ESH: log2 of the byte stride between two entries of the table -- 8: 32 copies of every entry, the conflict-free layout
// above (16 KiB); 6: eight copies (4 KiB; lookups of different entries in one column collide) for the one kernel that needs
// the 12 KiB for a third point group's table (llk_eval_passes_kernel).
// EXACT: see above.
template <int ESH = 8, bool EXACT = false>
__device__ __forceinline__ double exp_nonpos(double x, uint32_t& etab_lane)
{
    const double kInvStep = 0x1.71547652b82fep+6;        // 64/ln2
    const double kStepHi = 0x1.62e42fee00000p-7;         // ln2/64, 32 significant bits: k*hi is exact
    const double kStepLo = 0x1.a39ef35793c76p-39;
    const double kMagic = 0x1.8p52;                      // 2^52 + 2^51: x + kMagic rounds x to an integer
    // EXACT: below -800 the result is 0 anyway (2^-1154); the clamp also maps -inf to a finite
    // argument, so no special case is needed after the ldexp.  (The other form is only called with x >= -708.)
    if constexpr (EXACT) x = vmax_f64(x, -800.0);
    // k = rint(x * 64/ln2) by the magic-number addition: the low mantissa word of the sum IS the
    // integer (two's complement), so no v_rndne / v_cvt_i32 pair
    const double tk = fma(x, kInvStep, kMagic);
    const int k = __double2loint(tk);
    const double kd = tk - kMagic;
    double r = fma(-kd, kStepHi, x);
    r = fma(-kd, kStepLo, r);
    double t;
    if constexpr (ESH == 8) {
        // The conflict-free table: entry j of this lane's column sits at LDS address j << 8 | etab_lane, etab_lane < 256 -- the
        // index IS byte 1 of the address.  One SDWA instruction masks k to six bits and drops them into byte 1 of the
        // register that holds etab_lane, the other bytes kept (a shift and an and-or before).  The register is the lane's
        // for the whole kernel: byte 1 is rewritten by every call, byte 0 never.
        asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(etab_lane)
            : "v"(k), "s"(63));
        t = *reinterpret_cast<lds_cdouble*>(etab_lane);
    } else {
        t = *reinterpret_cast<lds_cdouble*>((((uint32_t)k << ESH) & (63u << ESH)) | etab_lane);
    }
    // exp(r) - 1 = r + r^2 q(r), q of degree 3 through the Chebyshev nodes of |r| <= ln2/128 (60-digit arithmetic; relative error
    // of exp 4.4e-18): worst error against expl over [-700, 0] 1.02 ulp
    double p = fma(r, 0x1.11111d8fbe766p-7, 0x1.55556b3304ec0p-5);
    p = fma(p, r, 0x1.5555555555255p-3);
    p = fma(p, r, 0x1.ffffffffff57fp-2);
    p = fma(p, r * r, r);
    if constexpr (EXACT) {
        t = __hiloint2double(__double2hiint(t) + ((k & 63) << 14), __double2loint(t));     // the entry itself again
        return ldexp(fma(t, p, t), k >> 6);
    } else {
        t = __hiloint2double(__double2hiint(t) + (int)((uint32_t)k << 14), __double2loint(t));   // 2^(k>>6) 2^(j/64), exactly
        return fma(t, p, t);
    }
}

// log(x) for x >= 0, FP64 (fdlibm's e_log algorithm with explicit FMAs, ~1 ulp):
// x = 2^e * m, m in [sqrt(1/2), sqrt(2)); f = m-1; s = f/(2+f);
// log(m) = f - hfsq + s*(hfsq + R(s^2)).
// log(0) = -inf (table entries whose probability is exactly 0, e.g. q = 0 / hom-ref).
// The library log() is not used: built without FMA contraction (this file's
// -ffp-contract=off, needed for reference-order rounding elsewhere) it expands to
// ~115 instructions, four times this.
__device__ __forceinline__ double log_nonneg(double x)
{
    const double kLn2Hi = 6.93147180369123816490e-01, kLn2Lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1), subnormals included
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);                  // 1/d, refined twice (Newton)
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double sq = f * r;
    sq = fma(fma(-d, sq, f), r, sq);                     // s = f/d to ~0.5 ulp
    const double z = sq * sq;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    const double res = fma(dk, kLn2Hi, -((hfsq - fma(sq, hfsq + R, dk * kLn2Lo)) - f));
    return x == 0.0 ? -__builtin_huge_val() : res;
}

// log(x) for 0 <= x <= 1 and normal (the per-alpha table's entries: logarithms of probabilities), table-driven -- round 4,
// second session: the six tables of a 48-point launch were 3.9 us of its 65 (the launch with the table build compiled out),
// nearly all of it the fdlibm-style logarithm above: ~40 dependent instructions, with a reciprocal and two Newton steps.
// Here: x = 2^k z, the interval i of z (seven mantissa bits after an offset that puts 1 INSIDE an interval) gives
// {1 / c_i, log c_i} from a 2 KiB table in LDS; r = fma(z, 1/c, -1) is exact and |r| <= 2^-8;
// log x = k ln2 + log c + (r + r^2 P(r)), P of degree 4.  ~22 instructions, none of them a transcendental.
// The table and the polynomial: log_table.inc, generated with 60-digit arithmetic by tools/gen_log_table.py (the interval
// around 1 has c = 1 exactly: log(1) = 0, and log(1 - p_err) -- most entries of a pileup's table -- keeps full relative
// accuracy; the other reciprocals are chosen so that log c is a double to within 0.001 ulp).  oracle/check_log_table.c is
// this routine statement for statement on the host, held to <= 1.6 ulp against the C library's long-double logarithm
// (tests/test_oracle_golden.py; typical < 0.8, the 1.5 where k ln2 + log c sits a binade above the result).
#include "log_table.inc"
constexpr int kLogTabDoubles = 256;
__device__ const double2 kLogTabRows[128] = {VB2_LOG_TAB_ROWS};
// (the routine proper: x normal and positive, hi its high word, kadj what the caller scaled x by)
__device__ __forceinline__ double log_tab_core(double x, uint32_t hi, int kadj, uint32_t ltab_addr)
{
    const double kLn2Hi = 0x1.62e42fefa3800p-1, kLn2Lo = 0x1.ef35793c76730p-45;
    constexpr double kP[5] = {VB2_LOG_POLY};
    const uint32_t tmp = hi - 0x3FE5F000u;
    const uint32_t i = (tmp >> 13) & 127u;
    const int k = ((int)tmp >> 20) - kadj;
    const double z = __hiloint2double((int)(hi - (tmp & 0xFFF00000u)), __double2loint(x));
    const vdouble2 t = *reinterpret_cast<lds_cdouble2*>(ltab_addr + i * 16u);
    const double r = fma(z, t.x, -1.0);
    const double kd = (double)k;
    const double w = fma(kd, kLn2Hi, t.y);
    const double r2 = r * r;
    // (the polynomial in two halves and k ln2's low part next to r: five dependent steps behind r instead of seven)
    const double q0 = fma(r, kP[1], kP[0]);
    double q1 = fma(r, kP[3], kP[2]);
    q1 = fma(r2, kP[4], q1);
    const double p = fma(r2, q1, q0);
    const double rl = fma(kd, kLn2Lo, r);
    const double y = fma(r2, p, rl);
    return w + y;
}
__device__ __forceinline__ double log_tab(double x, uint32_t ltab_addr /* LDS byte address of the table's copy */)
{
    const uint32_t hi = (uint32_t)__double2hiint(x);
    double res = log_tab_core(x, hi, 0, ltab_addr);
    // Outside the table's domain (one compare on the high word, and a branch no wave of a search ever takes --
    // a real branch: as selects folded into the routine the four cases cost every entry a dozen instructions):
    // zero -> -inf (the reference's log(0): entries whose probability is exactly 0); a positive subnormal (alpha below
    // 2^-1022: a logit under -708) -> scaled into the normal range, 64 ln2 taken off again through k; NaN, a negative
    // "probability" (alpha outside [0, 1]) -> NaN; +inf -> +inf: libm's answers, so that a NaN likelihood reaches the
    // caller as NaN and not as a plausible number.
    if (__builtin_expect(hi - 0x00100000u >= 0x7FE00000u, 0)) {
        if (x == 0.0) res = -__builtin_huge_val();
        else if (!(x > 0.0)) res = __builtin_nan("");
        else if (hi >= 0x7FF00000u) res = x;
        else {
            const double xs = x * 0x1p64;
            res = log_tab_core(xs, (uint32_t)__double2hiint(xs), 64, ltab_addr);
        }
    }
    return res;
}

// A positive value as (mantissa in [0.5,1), binary exponent): products of likelihoods are
// kept in this form so that a marker's log is never taken -- one log per (workgroup, point)
// replaces one per (marker, point).
struct ScaledProd {
    double m;     // mantissa product
    double e;     // exponent sum (exact integer)
};
__device__ __forceinline__ void sp_renorm(ScaledProd& p)
{
    p.e += (double)__builtin_amdgcn_frexp_exp(p.m);
    p.m = __builtin_amdgcn_frexp_mant(p.m);
}
// The per-lane running product of the static deal: integer exponent (one v_add_u32 per marker)
// and a mantissa that is only renormalised every kLazyRenorm factors -- each factor is a
// mantissa in [0.5, 1), so 256 of them cannot underflow.
struct LaneProd {
    double m;
    int e;
};
constexpr int kLazyRenorm = 256;

// One table entry, with the reference's expression order (h:223-225).
// perr_signed = +pErr(q) for class ref, -pErr(q) for class alt (one load per code;
// the alt class is the ref class with genotypes mirrored, g -> 2-g: h:164-177).
__device__ __forceinline__ double table_entry(double alpha, double perr_signed, int g1, int g2, uint32_t ltab_addr)
{
    const bool alt = perr_signed < 0.0;
    const double p_err = fabs(perr_signed);
    const double p_ok = 1.0 - p_err;
    if (alt) { g1 = 2 - g1; g2 = 2 - g2; }
    // class ref: P(ref | g, error) = {0, 1/6, 1/3}[g], P(ref | g, no error) = {1, .5, 0}[g].
    // As arithmetic on g (exact: fl(1/3) = 2*fl(1/6), the same mantissa one binade up; 1 - g/2
    // is exact), which is cheaper than selecting among 64-bit constants:
    const double e1 = (double)g1 * (1.0 / 6.0), e2 = (double)g2 * (1.0 / 6.0);
    const double n1 = 1.0 - 0.5 * (double)g1, n2 = 1.0 - 0.5 * (double)g2;
    const double one_minus_alpha = 1.0 - alpha;
    const double val = (alpha * e1 + one_minus_alpha * e2) * p_err +
                       (alpha * n1 + one_minus_alpha * n2) * p_ok;
    return log_tab(val, ltab_addr);
}

// The same entry BEFORE its logarithm, class ref (probability-domain contexts: llk_kernels.h, kMaxPow): the value the
// reference takes the logarithm of (h:223-225, the same expression order); the table build multiplies it up to P^2 .. P^K.  A
// negative "probability" (alpha outside [0, 1]) or a NaN becomes NaN, like the reference's log() of it: the marker's
// likelihood is then NaN, fails `markerLK > 0` and the marker is left out.
__device__ __forceinline__ double prob_entry(double alpha, double p_err, int g1, int g2)
{
    const double p_ok = 1.0 - p_err;
    const double e1 = (double)g1 * (1.0 / 6.0), e2 = (double)g2 * (1.0 / 6.0);
    const double n1 = 1.0 - 0.5 * (double)g1, n2 = 1.0 - 0.5 * (double)g2;
    const double one_minus_alpha = 1.0 - alpha;
    double val = (alpha * e1 + one_minus_alpha * e2) * p_err +
                 (alpha * n1 + one_minus_alpha * n2) * p_ok;
    return val >= 0.0 ? val : __builtin_nan("");
}

__device__ __forceinline__ void initial_gf(double af, double* gf)   // h:186-192
{
    // (v_max_f64 / v_min_f64 instead of compare + select; a NaN allele frequency -- NaN
    // parameters -- becomes min_af here where the reference keeps the NaN)
    af = vmin_f64(vmax_f64(af, 0.00005), 0.99995);
    gf[0] = (1 - af) * (1 - af);
    gf[1] = 2 * (af) * (1 - af);
    gf[2] = af * af;
}

// Row stride of the LDS table: L.row_bytes (llk_kernels.h: kRowBytesWide = 6 values x 8 points
// + 2 doubles of padding, or kRowBytesNarrow = 6 x 4 + 2), the same for every wave shape of a
// context because the run words of the pileup carry PRE-MULTIPLIED row offsets.  Either stride
// is an odd number of 16-byte slots -> 16 consecutive codes land in 16 distinct 4-dword bank
// slots for ds_read_b128 (bank = (addr/4) % 64).

constexpr int kExpTabDoubles = 64 * 32;    // exp_nonpos's table in LDS (16 KiB)
constexpr int kPrefetch = 8;               // rows of run dwords in flight per lane: cohort steps (lists from HBM)
constexpr int kPrefetchL2 = 2;             // ... a single sample's launches and search rounds (lists in L2 / LDS)

// Lane -> (marker m in the micro-tile, candidate slot g): the 16 lanes that ds_read_b128 services in one LDS pass
// (lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32; MI355X_MICROARCH.md, LDS) share one candidate
// slot, so within a pass the addresses differ only by the code.
__device__ __forceinline__ void lane_map(int lane, int& m, int& g)
{
    const int q = (lane >> 2) & 7;                       // quad within the 32-lane half
    const int nib = (0x76452310u >> (4 * q)) & 7;        // (rank<<1 | group) of quad q
    g = (nib & 1) + 2 * (lane >> 5);
    m = ((nib >> 1) << 2) + (lane & 3);
}
__device__ __forceinline__ int lane_of(int m, int g)
{
    const int idx = ((g & 1) << 2) + (m >> 2);           // group*4 + quad-rank
    const int q = (0x74216530u >> (4 * idx)) & 7;
    return ((g >> 1) << 5) + (q << 2) + (m & 3);
}

// The value of lane `partner` = the lane whose marker index differs in bit `off` (same slot).  The lane map keeps the low
// two marker bits in the low two lane bits, so the exchanges over off = 1 and 2 stay inside a quad: DPP quad permutes -- VALU
// moves, no trip through the LDS crossbar (a ds_bpermute per dword and ~100 cycles before the dependent multiply).  Marker
// bit 2 selects between the quads {0,3}, {1,2} of a 16-lane row (lane_of: ranks 0,1,2,3 of slot-parity 0 sit in quads
// 0,3,5,6, of parity 1 in 1,2,4,7), i.e. quad j <-> 3 - j with the lane in the quad kept: a row rotation by 4 for the
// quads 0 and 2, by 12 for 1 and 3 (tools/ubench/dpp_map.hip: row_ror:n gives lane i the value of lane i - n of its row)
// -- two DPP moves with bank masks.  The exchange over 8 crosses rows and stays a ds_bpermute (as three VALU moves per dword
// -- lane ^ 4, then v_permlane16_swap -- it was measured slower: HISTORY.md, round 4).  Multiplication commutes: the same bits
// whatever the route.
__device__ __forceinline__ int lane_xchg_i32(int x, int off, int partner)
{
    if (off == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    if (off == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    if (off == 4) {
        const int a = __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0x5, false);       // row_ror:4  -> quads 0 and 2 of every row
        return __builtin_amdgcn_update_dpp(a, x, 0x12C, 0xf, 0xa, false);              // row_ror:12 -> quads 1 and 3
    }
    return __shfl(x, partner, 64);
}
__device__ __forceinline__ double lane_xchg_f64(double x, int off, int partner)
{
    if (off == 1 || off == 2 || off == 4)
        return __hiloint2double(lane_xchg_i32(__double2hiint(x), off, partner), lane_xchg_i32(__double2loint(x), off, partner));
    return __shfl(x, partner, 64);
}

// Occupancy target: one 1024-thread workgroup per CU (4 waves/SIMD, <=128 VGPRs).  Two
// smaller workgroups per CU were measured slower: the per-alpha table is built once per
// workgroup, and that redundant work is 9 % of a launch's transcendentals at one
// workgroup per CU but 18 % at two.
template <int MODE> struct Geom { static constexpr int kMaxWaves = 16, kBlocksPerCU = 1, kWavesPerSimd = 4; };
// points per group of a wave shape
template <int MODE> struct ModeNp { static constexpr int value = MODE == 2 ? 8 : MODE == 4 ? 1 : MODE == 5 ? 2 : 4; };

// MODE picks the wave shape.  BTL = candidate points per lane, SLOTS = candidate slots per wave:
//   MODE 2: 16 markers x 4 slots x 2 points  (NP = 8)
//   MODE 3: 2 x 16 markers x 2 slots x 2 points (NP = 4): the wave takes TWO micro-tiles, so a
//           4-point launch (a Nelder-Mead iteration) amortises the per-run bookkeeping over two
//           points per lane like MODE 2 does, instead of one.
//   MODE 4: 4 x 16 markers x 1 slot x 1 point (NP = 1): a single-point evaluation (a caller's own
//           optimiser, Initialize, LLK0) does a quarter of the work of a 4-point launch instead of
//           evaluating the point four times.
//   MODE 5: 4 x 16 markers x 1 slot x 2 points (NP = 2): cohort steps of a search that speculates on
//           one extra point per iteration (multi-sample kernel only).
// A launch evaluates groups of NP points (num_valid of them real; the rest replicate the last).
// The body is shared by the single-sample kernel (blk = blk, nblk = nblk) and
// the multi-sample kernel (blk/nblk = this workgroup's index among its sample's workgroups).
// Parameter rows passed inside the kernel-argument segment (host-pointer path with few
// points): saves the PCIe round trip of reading them from mapped host memory.
struct InlinePoints {
    int count;                       // doubles valid in v (0 = read from the pointer)
    double v[kInlinePointDoubles];
};

// A wave of the workgroup that spends the tile phase on something else (the resident kernel's control wave:
// resident_kernel.inc).  on_block: this workgroup has such a wave (wave 0); mine: this wave is it.  Only with
// the dynamic work queue (eval_is_dynamic): the other waves then start on items 0..nwave-2 and pull the rest.
struct NoHook {
    __device__ bool on_block() const { return false; }
    __device__ bool mine() const { return false; }
    __device__ void run() {}
    __device__ bool keep_etab() const { return false; }     // (the 2^(j/64) table of an earlier call is still in LDS)
    __device__ bool prim_in_lds() const { return false; }   // the primary-code records live in LDS across calls ...
    __device__ bool keep_prim() const { return false; }     // ... and an earlier call left them there
    __device__ double* sum_stage() const { return nullptr; } // where workgroup 0 stages the partial sums (nullptr: over the dead tables)
    __device__ uint32_t cache_rec() const { return 0u; }    // LCACHE: LDS byte address of this workgroup's tile records
    __device__ unsigned int* sync_word() const { return nullptr; }   // a zeroed LDS word: the hook wave skips the prologue (see eval_body)
    __device__ bool collect_only() const { return false; }  // this workgroup owns no tiles: it collects the other workgroups' sums (and hosts the hook wave)
    __device__ bool no_collect() const { return false; }    // the tagged sets are collected by such a workgroup, not by workgroup 0
};
// The waves of a workgroup pull (tile, group) items through the LDS queue (per-item result slots) when there
// are at most dyn_limit of them per wave; else the static deal (cohort launches).
__host__ __device__ __forceinline__ bool eval_is_dynamic(const DeviceLayout& L, uint32_t nblk, int nwave, int ngrp)
{
    const uint32_t max_tiles_blk = owned_most(L.pd ? 1 : 0, (uint32_t)L.num_mt, nblk);
    return max_tiles_blk * (uint32_t)ngrp <= (uint32_t)(L.dyn_limit * nwave);
}

// STREAM: the launch reads its samples' run lists from HBM (cohort steps) rather than from L2: the prefetch never
// runs past a tile's own rows then (see the read loop).
// QUEUE: 1 = the launch is known (on the host, by eval_is_dynamic) to take its work items through the LDS queue, 0 = the
// static deal, -1 = decided in the kernel.  The single-sample kernels are compiled for both: with the other way's code
// gone a 48-point launch is 1.8 % shorter and a search round 5 % (fewer scalar registers spilled to vector lanes).
// KSEL: --NumPC when it is 2 (the reference's default) or 4 (its usual setting) AND the context has no known-allele-
// frequency column (the usual case: AF from UD x PC), else 0 = both read from the layout: the guards and address
// multiples of the projection and the known-AF tests go at compile time (+2.6 % on the 48-point launch).
// LCACHE (the resident search kernel): this workgroup's run lists and tile records were copied to LDS when the kernel
// started (a workgroup owns the same micro-tiles in every round), so a round's read loops begin without the two dependent
// trips to L2 (tile record, then its first rows) and never wait for a row again.
// ESH: log2 of the byte stride between the entries of exp_nonpos's table (8: conflict-free; 6: the compact copy).
// (Measured and dropped over the rounds, all bit-identical: a software-pipelined item loop, a run-ahead ring of table reads,
// the next item drawn a whole item early, two half-sized workgroups per CU -- HISTORY.md.)
// PD: a probability-domain context (DeviceLayout::pd; llk_kernels.h, kMaxPow): the table holds P^n rows of class ref only,
// a marker's list is one 16-bit row offset per step, ref steps first and alt steps behind them; the six sums are PRODUCTS,
// class alt multiplies them with the row read the other way round (g -> 2 - g, h:164-177: T[alt][q][g1][g2] is T[ref][q][2-g1][2-g2]),
// and the epilogue needs no exponential.  Only for contexts whose markers cannot underflow that way (Context::create).
// SPLIT (a probability-domain launch of more points than a workgroup's LDS holds tables for -- a dictionary of ~100 rows and 48
// points): the workgroups come in PAIRS that share their tiles and split the point groups -- workgroup 2v takes the first
// half of the groups, 2v + 1 the second, both over the tiles of the VIRTUAL blocks v and v + grid / 2, i.e. over exactly the
// tiles two workgroups of a plain launch own.  A tile's product still has its own slot, a virtual block's slots are
// multiplied in the same order and its sum goes to the same word of the partial sums as in a plain launch of this grid:
// the same bits, and no second pass over the tiles (llk_eval_passes_kernel: +4 us per pass at C3).
template <int MODE, bool W16 = false, class Hook = NoHook, bool STREAM = false, int QUEUE = -1, int KSEL = 0,
          bool LCACHE = false, int ESH = 8, bool PD = false, int SPLIT = 0>
__device__ __forceinline__ void
eval_body(const DeviceLayout& L, const double* ip_v, const int ip_count, const double* __restrict__ points, int num_valid,
          double* __restrict__ partials, double* __restrict__ llk_out,
          unsigned int* __restrict__ ticket, unsigned long long* __restrict__ done_flag,
          unsigned long long done_seq, const uint32_t blk, const uint32_t nblk,
          unsigned int* __restrict__ batch_done, unsigned int batch_active, const int ngrp_in,
          const unsigned long long tag, const Schedule sch, const bool coherent_points = false,
          const int tid_in = -1, const double* lds_rows = nullptr /* the parameter rows, already in LDS */,
          Hook hook = Hook())
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    static_assert(MODE >= 2 && MODE <= 5, "wave shapes 2..5");
    static_assert(!PD || ESH == 8, "probability-domain contexts have no exp table");
    static_assert(!(PD && W16) || ((MODE == 4 || MODE == 5) && STREAM), "8-bit step lists: cohort steps of one and two points");
    static_assert(!SPLIT || (PD && MODE == 2 && QUEUE == 1 && !STREAM && !LCACHE), "split launches: the 8-point shape on the queue");
    constexpr int OSH = PD ? 1 : 0;             // a workgroup owns pairs of neighbouring micro-tiles (owned_tile)
    // the launch is known to carry ONE group of points (every shape but the 8-point one always does; cohort steps too):
    // the group loops and the item -> (group, unit) division go at compile time
    constexpr bool ONEGRP = MODE != 2 || STREAM;
    constexpr int KAF = KSEL > 0 ? 0 : -1;      // 0: no known-AF column, known at compile time
    constexpr int BTL = MODE == 4 ? 1 : 2;
    constexpr int TPW = MODE == 3 ? 2 : (MODE == 4 || MODE == 5) ? 4 : 1;   // micro-tiles per wave
    constexpr int SLOTS = 4 / TPW;                           // candidate slots per wave
    constexpr int NP = SLOTS * BTL;
    const int row_bytes = MODE == 2 ? kRowBytesWide : L.row_bytes;      // (the 8-point shape needs the wide rows: the launcher sees to it)
    const int RS = row_bytes >> 3;              // doubles per table row (>= 6 * NP + 2)
    const int nrow = L.num_code + 1;
    const int nthread = blockDim.x;
    const int nwave = nthread >> 6;
    const int k = KSEL > 0 ? KSEL : L.num_pc;
    const int stride = 2 * k + 1;
    // A launch carries ngrp groups of NP points; each group has its own table and the
    // (tile, group) pairs are the work items, so a bigger batch re-reads the pileup from
    // L2 per group but pays launch, prologue and reduction once.
    // (SPLIT = S workgroups share their tiles: this workgroup's share of the launch's groups, the points before them, and its S
    // virtual blocks vb0 + j * vstep)
    constexpr int NVS = SPLIT ? SPLIT : 1;      // virtual blocks per workgroup
    const int g_per = SPLIT ? (ngrp_in + NVS - 1) / NVS : ngrp_in;
    const int g_lo = SPLIT ? (int)(blk % (uint32_t)NVS) * g_per : 0;
    const int ngrp = ONEGRP ? 1 : SPLIT ? (ngrp_in - g_lo < g_per ? ngrp_in - g_lo : g_per) : ngrp_in;      // (the launcher sees to it that none is empty)
    const int p_off = g_lo * NP;
    const uint32_t vstep = SPLIT ? nblk / (uint32_t)NVS : 0u, vb0 = SPLIT ? blk / (uint32_t)NVS : blk;
    const double* const known_af_p = KAF == 0 ? nullptr : L.known_af;
    const int NPT = NP * ngrp;                  // points of this launch
    constexpr int kEtabCopies = (1 << ESH) / 8, kEtabDoubles = PD ? 0 : 64 * kEtabCopies;
    constexpr int kLtabDoubles = PD ? 0 : kLogTabDoubles;
    double* etab = lds;                         // [64][32] exp_nonpos's 2^(j/64), bank-replicated; at LDS address 0
    double* ltab = lds + kEtabDoubles;          // [128] {1 / c, log c} of log_tab
    double* tab = ltab + kLtabDoubles;          // [ngrp][nrow][RS]  (PD: neither table above: this one is at address 0)
    double* red = tab + ngrp * nrow * RS;       // [virtual blocks][NPT] block sums; then the work-queue counter
    unsigned int* queue = reinterpret_cast<unsigned int*>(red + NVS * NPT);
    double* pts = red + NVS * NPT + 2;          // [NPT][2k+1] this launch's parameter rows
    // the PCs again, as the epilogue reads them: [group][slot][2k][BTL] -- a lane's BTL points side by
    // side, so one ds_read_b128 (4 LDS cycles) brings a coefficient of both points where two
    // strided 8-byte reads (ds_read2_b64: 8 cycles) did
    double* ptq = pts + (size_t)NPT * stride + ((NPT * stride) & 1);
    const size_t prim_off = (size_t)(ptq - lds) + (size_t)NPT * 2 * k;
    double2* prim_lds = reinterpret_cast<double2*>(lds + prim_off + (prim_off & 1));   // [num_prim], 16-B aligned
    double* tile_llk = reinterpret_cast<double*>(prim_lds + L.num_prim);   // [work items or waves][NP] {mantissa, exponent}

    // (tid_in: the resident kernel passes its thread index through an opaque register each round,
    // so that nothing derived from it is hoisted out of the round loop and kept alive across it)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and the compiler knows it)
    int m, g4;
    lane_map(lane, m, g4);
    const int g = g4 & (SLOTS - 1);              // candidate slot
    const int half = TPW == 1 ? 0 : g4 / SLOTS;  // which of the item's TPW micro-tiles
    // profiling aid: 100 MHz wall-clock stamps per workgroup (stamps pointer null normally)
    // (kernel_debug.h: VB2_STAMP_ROUND; the accumulated figures cover every round either way)
    const bool stamp_this = VB2_STAMP_ROUND == 0 || (unsigned)(tag & 0xffffffffull) == (unsigned)VB2_STAMP_ROUND;
#ifdef VB2_ITEM_PROF
    unsigned long long* stamps = nullptr;                 // (the per-workgroup slots hold the item profile's sums in this build)
#else
    unsigned long long* stamps = (VB2_STAMPS_OF(L) && stamp_this) ? VB2_STAMPS_OF(L) + (size_t)blk * 8 : nullptr;
#endif
    if (stamps && tid == 0) { stamps[0] = wall_clock64(); stamps[4] = 0; }

    // A workgroup that owns no tiles (the resident kernel's extra workgroup for the control wave: Hook::collect_only): nothing of
    // the evaluation -- its hook wave does its work, then all of it collects the tile workgroups' sums below.
    const bool collect_only = hook.collect_only();
    if (collect_only && hook.mine()) hook.run();
    if (!collect_only) {
    const bool hook_blk = hook.on_block(), hook_mine = hook.mine();
    // A workgroup with a hook wave (the resident kernel's control wave: wave 0 of workgroup 0) in a probability-domain search
    // round: that wave starts on its own work AT ONCE -- it builds no table entries and does not come to the prologue's barriers,
    // which the other waves replace by a counter in LDS (hook.sync_word(), zero at the round's start).  Its work is the round's
    // longest dependent chain (resident_kernel.inc: control_tile_phase, ~6 us): begun behind the table build it ended 1.5 us
    // after every other workgroup had its sums in.
    unsigned int* const soft_word = PD ? hook.sync_word() : nullptr;
    const bool soft = hook_blk && soft_word != nullptr;
    if (soft && hook_mine) hook.run();
    const int ptid = !soft ? tid : hook_mine ? 0x3fffffff : tid - 64, pnthread = soft ? nthread - 64 : nthread;
    unsigned int soft_gen = 0;
    auto prologue_sync = [&]() {
        if (!soft) { __syncthreads(); return; }
        if (hook_mine) return;
        soft_gen += (unsigned int)(nwave - 1);
        // (a wave's LDS operations are performed in order: its table entries are in place when its arrival is counted)
        if (lane == 0) __hip_atomic_fetch_add(soft_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(soft_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < soft_gen) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    if (ptid == 0) *queue = (unsigned int)(nwave - (hook_blk ? 1 : 0));      // waves start on tiles 0..nwave-1
    // parameter rows -> LDS with one coalesced load (they may live in mapped host memory)
    for (int e = ptid; e < NPT * stride; e += pnthread) {
        const int b = e / stride;
        const int src = p_off + b < num_valid ? p_off + b : num_valid - 1;
        const int idx = src * stride + (e - b * stride);
        // resident mode: the rows were just written by another workgroup -> L1-bypassing loads
        const double v = (kAblate & kAblNoMap) ? 0.01 * (double)(1 + idx % 7)
                         : lds_rows ? lds_rows[idx]
                         : ip_count > 0 ? ip_v[idx]
                         : coherent_points ? __hip_atomic_load(&points[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : points[idx];
        pts[e] = v;
        const int c = e - b * stride;                       // 0..2k-1: a PC coordinate, 2k: alpha
        if (c < 2 * k) ptq[((b / BTL) * 2 * k + c) * BTL + (b % BTL)] = v;
    }
    // 2^(j/64) table of exp_nonpos, one copy per pair of LDS banks (see there)
    // (entry j = e / 32 is the same for a half-wave: two SCALAR loads per step, no vector-memory
    // round trip before the first barrier)
    if (!PD && !hook.keep_etab() && tid < kLogTabDoubles / 2)
        reinterpret_cast<double2*>(ltab)[tid] = kLogTabRows[tid];
    if (!PD && !hook.keep_etab())
        for (int jb = 2 * wave; jb < 64; jb += 2 * nwave) {
            const double t0 = kExp2Tab[jb], t1 = kExp2Tab[jb + 1];
            const double tv = lane < 32 ? exp_tab_entry(t0, jb) : exp_tab_entry(t1, jb + 1);
            if constexpr (ESH == 8) etab[jb * 32 + lane] = tv;
            else if ((lane & 31) < kEtabCopies) etab[(jb + (lane >> 5)) * kEtabCopies + (lane & 31)] = tv;
        }
    // With several groups a thread builds several table entries: the primary-code records (a
    // few dozen) go to LDS first so that the loop below does not wait on a global load per
    // entry.  With one group each thread builds about one entry and loads its record directly.
    // (the resident kernel keeps them in LDS from its first round on: a round's table then starts without a trip to L2)
    const bool prim_kept = hook.prim_in_lds() && hook.keep_prim();
    const bool staged = ngrp > 1 || (hook.prim_in_lds() && !prim_kept) || (PD && L.num_pair > 0 && !prim_kept);      // (pair rows: their records from LDS)
    if (staged)
        for (int e = ptid; e < L.num_prim; e += pnthread) prim_lds[e] = L.prim[e];
    // A search round (its rows are in LDS since the round's staging barrier) builds its one table without waiting
    // for the copies above: the table needs the alphas only, and takes them from the staged rows (-0.6 us per
    // round; the same for launches whose rows come with the kernel arguments: no gain, not kept).
    const bool early_table = lds_rows != nullptr && !staged;
    if (!early_table) prologue_sync();
    if (stamps && (soft ? ptid : tid) == 0) stamps[1] = wall_clock64();

    // ---- per-alpha table, off-diagonal pairs only (h:213-229) ----
    // Class alt is class ref with the genotypes mirrored (g -> 2-g, h:164-177): the entry of
    // (alt, q) for pair p IS the entry of (ref, q) for pair 5-p, the same expression on the same
    // numbers.  So only the "primary" codes are computed -- every ref code, and alt codes whose
    // quality has no ref code in the data -- and the thread that computes a ref entry also stores
    // it into the alt twin's row: half the logarithms, and no barrier in between.
    // (a thread's entries for the different groups are independent: computed side by side so
    // that their long dependent logarithm chains overlap)
    const uint32_t ltab_addr = lds_byte_addr(ltab);
    const int num_single = PD ? L.num_prim - L.num_pair : L.num_prim;
    for (int e = ptid; e < ((kAblate & kAblNoTable) ? 0 : num_single * 6 * NP); e += pnthread) {
        const int pi = e / (6 * NP);
        const int bp = e - pi * (6 * NP);
        const int bb = bp / 6, p = bp - bb * 6;
        const double2 rec = (staged || prim_kept) ? prim_lds[pi] : L.prim[pi];       // {signed pErr, code | twin << 16}
        const uint32_t pr = (uint32_t)__double_as_longlong(rec.y);
        const int dc = (int)(pr & 0xffffu), twin = (int)(pr >> 16);
        int g1, g2;
        pair_of(p, g1, g2);
        // W16 (cohort steps on the 16-bit run lists): the run word's count field decodes to the double 2 * n with ONE
        // byte permute (see the read loop), so the table holds T / 2 -- both scalings are exact (powers of two), and
        // fma(2n, T/2, acc) rounds the same real number as fma(n, T, acc): bit-identical to the 32-bit lists.
        // (the groups one after the other in a rolled loop: side by side -- six logarithms' chains interleaved -- the
        // unrolled body held every group's temporaries and constants at once and spilled scalar registers to vector lanes;
        // it cost 430 vector instructions per thread of a 48-point launch where this costs 285: headline +0.8 %, 118 codes +2.2 %)
        if constexpr (PD && !ONEGRP) {
            // (probability domain, several groups: an entry is a dozen dependent FP64 operations and K dependent multiplies -- no
            // logarithm's forty --, so THREE groups' entries side by side cost few registers and hide each other's latencies:
            // the tables of a split launch's workgroup 1.8 -> 1.6 us (127 KB of LDS stores: ~0.75 us at the LDS's store rate),
            // the 48-point launch 51.35 -> 50.6 us on one box)
            constexpr int kSide = 3;
            for (int g0 = 0; g0 < ngrp; g0 += kSide) {
                double v3[kSide], r3[kSide];
                double* cell3[kSide];
#pragma unroll
                for (int u = 0; u < kSide; ++u) {
                    const int ge = g0 + u < ngrp ? g0 + u : ngrp - 1;      // (past the last group: the last one again, the same values)
                    v3[u] = prob_entry(pts[(ge * NP + bb) * stride + 2 * k], rec.x, g1, g2);
                    r3[u] = v3[u];
                    cell3[u] = tab + (size_t)ge * nrow * RS + dc * RS + bp;
                    *cell3[u] = r3[u];
                }
                const int kq = twin & 0xff, pstride = RS * (int)(int8_t)(twin >> 8);      // (record: first row | K << 16 | rows from P^n to P^(n+1), a signed byte, << 24)
                for (int n = 1; n < kq; ++n) {
#pragma unroll
                    for (int u = 0; u < kSide; ++u) {
                        r3[u] *= v3[u];
                        cell3[u] += pstride;
                        *cell3[u] = r3[u];
                    }
                }
            }
        } else
#pragma clang loop unroll(disable)
        for (int grp_e = 0; grp_e < ngrp; ++grp_e) {
            const double alpha_e = early_table ? lds_rows[(bb < num_valid ? bb : num_valid - 1) * stride + 2 * k]
                                               : pts[(grp_e * NP + bb) * stride + 2 * k];
            double* gtab = tab + (size_t)grp_e * nrow * RS;
            if constexpr (PD) {
                // record pi = a QUALITY: {pErr, first row | K << 16}: its rows P^1 .. P^K follow each other, one multiply
                // apiece (class alt reads the same rows mirrored)
                const double v = prob_entry(alpha_e, rec.x, g1, g2);
                double r = v;
                double* cell = gtab + dc * RS + bp;
                // (shapes of at most four points: the row's second half holds its mirror image, pair p <-> 5 - p, where the
                // alt steps' offsets point -- kPdAltOffset)
                constexpr int kMir = kPdAltOffset / 8;
                const int mir = kMir + bb * 6 + (5 - p) - bp;
                *cell = r;
                if constexpr (NP <= 4 && STREAM) cell[mir] = r;
                const int kq = twin & 0xff, pstride = RS * (int)(int8_t)(twin >> 8);      // (record: first row | K << 16 | rows from P^n to P^(n+1), a signed byte, << 24)
                for (int n = 1; n < kq; ++n) {
                    r *= v;
                    cell += pstride;
                    *cell = r;
                    if constexpr (NP <= 4 && STREAM) cell[mir] = r;
                }
            } else {
                const double v = table_entry(alpha_e, rec.x, g1, g2, ltab_addr);
                const double tv = W16 ? 0.5 * v : v;
                gtab[dc * RS + bp] = tv;
                if (twin != 0xffff) gtab[twin * RS + bb * 6 + (5 - p)] = tv;
            }
        }
    }
    for (int e = ptid; e < ngrp * RS; e += pnthread) {                // padding code: zero rows (PD: ones)
        const int grp_e = e / RS;
        tab[((size_t)grp_e * nrow + L.num_code) * RS + (e - grp_e * RS)] = PD ? 1.0 : 0.0;
    }
    if constexpr (PD) {
        // ---- pair rows (PdDict): the product of two of the rows above, entry by entry -- one record per row
        // {bits: row a | row b << 16, bits: the row}, from LDS ----
        for (int level = 0; level < 2 && L.num_pair > 0 && !(kAblate & kAblNoTable); ++level) {
            // (level 1: products of two rows the records of this kind made -- windows of three and four qualities)
            const int nrec = level == 0 ? L.num_pair - L.num_pair2 : L.num_pair2;
            if (nrec == 0) break;
            prologue_sync();
            const double2* const prec = prim_lds + num_single + (level == 0 ? 0 : L.num_pair - L.num_pair2);
            for (int e = ptid; e < nrec * 6 * NP; e += pnthread) {
                const int pi = e / (6 * NP);
                const int bp = e - pi * (6 * NP);
                const double2 rec = prec[pi];
                const uint32_t ab = (uint32_t)__double_as_longlong(rec.x), dst = (uint32_t)__double_as_longlong(rec.y);
                const int ra = (int)(ab & 0xffffu) * RS + bp, rb = (int)(ab >> 16) * RS + bp, rd = (int)dst * RS + bp;
                constexpr int kMir = kPdAltOffset / 8;
                const int bb = bp / 6, p = bp - bb * 6;
                const int mir = kMir + bb * 6 + (5 - p) - bp;
#pragma clang loop unroll(disable)
                for (int grp_e = 0; grp_e < ngrp; ++grp_e) {
                    double* gtab = tab + (size_t)grp_e * nrow * RS;
                    const double v = gtab[ra] * gtab[rb];
                    gtab[rd] = v;
                    if constexpr (NP <= 4 && STREAM) gtab[rd + mir] = v;
                }
            }
        }
    }
    prologue_sync();
    if (stamps && (soft ? ptid : tid) == 0) stamps[2] = wall_clock64();

    // Work distribution.  Workgroup b owns micro-tiles b, b+grid, b+2*grid, ... (the tiles are
    // depth-sorted, so every workgroup -- hence every CU, and every XCD's L2 at every launch --
    // gets the same depth mix of the same data).  The (tile, group) work items of the workgroup
    // are dealt to its waves in snake order (static: a lane then keeps ONE running product per
    // group in registers and a work item costs no cross-lane traffic at all), or -- L.dyn_limit,
    // an A/B knob -- pulled longest-first through an LDS counter, each item's product going to its
    // own LDS slot.  Either way the multiplication order is fixed, so the schedule does not
    // change a single bit of the result.
    const uint32_t nt_v0 = owned_count(OSH, (uint32_t)L.num_mt, vb0, nblk);
    const uint32_t nt_v1 = SPLIT ? owned_count(OSH, (uint32_t)L.num_mt, vb0 + vstep, nblk) : 0u;
    const uint32_t nt_v2 = SPLIT > 2 ? owned_count(OSH, (uint32_t)L.num_mt, vb0 + 2u * vstep, nblk) : 0u;
    auto nt_of = [&](uint32_t vs) { return vs == 0u ? nt_v0 : vs == 1u ? nt_v1 : nt_v2; };
    // (SPLIT: per virtual block, the largest -- virtual block vb0 has it: owned_count does not grow with the block index)
    const uint32_t ntile_blk = nt_v0;
    const size_t mp = L.m_pad;
    // work items: (tile, group), or (TPW consecutive owned tiles, group)
    const uint32_t nunit = SPLIT ? (uint32_t)NVS * ntile_blk : (ntile_blk + TPW - 1) / TPW;
    const uint32_t nitem = (kAblate & kAblNoItems) ? 0u : nunit * (uint32_t)ngrp;
    const float inv_nunit = 1.0f / (float)(nunit ? nunit : 1u);
    const uint32_t ngrp_magic = ngrp > 1 ? 0xFFFFFFFFu / (uint32_t)ngrp + 1u : 0u;      // ceil(2^32 / ngrp) for ngrp >= 2
    // Decided on ceil(tiles / workgroups), the same for every workgroup and exactly what
    // eval_shmem_np sized the result slots for (a workgroup with one tile fewer must not choose
    // differently: the queue needs a slot per item, the static deal only one per wave).
    const bool dyn = QUEUE < 0 ? eval_is_dynamic(L, nblk, nwave, ngrp) : QUEUE != 0;
    // The pileup arrays through GLOBAL-address-space pointers: the resident kernel passes its layout through
    // opaque registers every round, after which the compiler no longer knows where the pointers came from
    // and would use flat loads -- whose waits also cover the LDS counter, i.e. every per-marker load would
    // be waited for together with the table reads.
    typedef unsigned int __attribute__((ext_vector_type(2))) vuint2;          // (a plain vector: loadable through address_space(1))
    typedef __attribute__((address_space(1))) const vuint2 g_cuint2;
    typedef __attribute__((address_space(1))) const double g_cdouble;
    typedef __attribute__((address_space(1))) const char g_cchar;
    g_cuint2* const g_rec = (g_cuint2*)(W16 ? L.mt_rec16 : L.mt_rec);
    g_cuint2* const g_codes = (g_cuint2*)(W16 ? L.codes16 : L.codes);
    // (PD: a row of the list is 16 x 4 bytes -- two 16-bit steps per marker -- where the run words' rows are 16 x 8)
    typedef __attribute__((address_space(1))) const uint32_t g_cuint;
    typedef __attribute__((address_space(3))) const uint32_t lds_cuint;
    typedef typename std::conditional<PD, uint32_t, vuint2>::type RowWord;
    constexpr uint32_t kRowBytes = PD ? kMtMarkers * 4u : kMtMarkers * 8u, kLaneBytes = PD ? 4u : 8u;
    g_cdouble* const g_ediag = (g_cdouble*)L.ediag;
    g_cdouble* const g_ud = (g_cdouble*)L.ud;
    g_cdouble* const g_mu = (g_cdouble*)L.mu;
    g_cdouble* const g_kaf = (g_cdouble*)known_af_p;
    // The per-marker likelihoods of a work item are MULTIPLIED (mantissa x 2^exponent): over
    // the 16 markers of the tile by a butterfly, then slot by slot in the block reduction.
    auto tile_product = [&](ScaledProd* p, bool cross) {  // over the 16 lanes sharing slot g (+ the item's other tiles)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const int partner = lane_of(m ^ off, g4);
#pragma unroll
            for (int t = 0; t < BTL; ++t) {
                p[t].m *= lane_xchg_f64(p[t].m, off, partner);    // 16 factors in [0.5,1): no underflow
                p[t].e += lane_xchg_f64(p[t].e, off, partner);
            }
        }
        if (!cross) return;
#pragma unroll
        for (int off = SLOTS; off < 4; off <<= 1) {       // the item's other micro-tiles
            const int partner = lane_of(m, g4 ^ off);
#pragma unroll
            for (int t = 0; t < BTL; ++t) {
                p[t].m *= __shfl(p[t].m, partner, 64);    // factors >= 2^-16 each: no underflow
                p[t].e += __shfl(p[t].e, partner, 64);
            }
        }
    };
    LaneProd wave_prod[BTL];                              // static mode: this lane's running product
#pragma unroll
    for (int t = 0; t < BTL; ++t) wave_prod[t] = LaneProd{1.0, 0};
    int nfactor = 0;                                      // factors in wave_prod.m since its last renormalisation
    auto renorm_wave = [&]() {
#pragma unroll
        for (int t = 0; t < BTL; ++t) {
            wave_prod[t].e += __builtin_amdgcn_frexp_exp(wave_prod[t].m);
            wave_prod[t].m = __builtin_amdgcn_frexp_mant(wave_prod[t].m);
        }
        nfactor = 0;
    };
    uint32_t grp_wave = 0;                                // static mode: group wave_prod belongs to
    auto flush_wave = [&](uint32_t grp) {                 // static mode: one slot per (wave, group)
        renorm_wave();
        ScaledProd sp[BTL];
#pragma unroll
        for (int t = 0; t < BTL; ++t) sp[t] = ScaledProd{wave_prod[t].m, (double)wave_prod[t].e};
        tile_product(sp, true);
        if (m == 0 && half == 0) {
#pragma unroll
            for (int t = 0; t < BTL; ++t) {
                const size_t o = (((size_t)grp * nwave + wave) * NP + g * BTL + t) * 2;
                tile_llk[o] = sp[t].m;
                tile_llk[o + 1] = sp[t].e;
            }
        }
#pragma unroll
        for (int t = 0; t < BTL; ++t) wave_prod[t] = LaneProd{1.0, 0};
    };
    uint32_t etab_lane = (uint32_t)(lane & (kEtabCopies - 1)) * 8u;     // etab is at LDS address 0 (no static LDS in this file)
    const uint32_t tab_addr = lds_byte_addr(tab);
    const uint32_t ptq_addr = lds_byte_addr(ptq);
    if (L.stagger > 0 && wave >= (nwave >> 1))
        for (int i = 0; i < L.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    // static mode: the host-built schedule (LPT over the waves), or items dealt in snake order
    const bool have_sched = !dyn && sch.off != nullptr;
    uint32_t s_i = 0, s_end = 0;
    if (have_sched) {
        s_i = sch.off[blk * (uint32_t)nwave + (uint32_t)wave];
        s_end = sch.off[blk * (uint32_t)nwave + (uint32_t)wave + 1];
    }
    uint32_t round = 0;
    if (hook_mine && !soft) hook.run();                   // (this wave takes no work items; the others cover for it)
#ifdef VB2_STAMP_CTRL     // (profiling build: workgroup 0's slot 3 = the control wave is back from its tile-phase work)
    if (stamps && hook_mine && lane == 0) stamps[3] = wall_clock64();
#endif
    typedef __attribute__((address_space(3))) const vuint2 lds_cuint2v;
    // (PD, the 8-point shape: its window of rows moves up a row at a time -- kept short)
    constexpr int kPf = (STREAM && !(PD && MODE == 2)) ? kPrefetch : kPrefetchL2;
    RowWord w[kPf];                                       // this lane's run words, kPf rows in flight
    // PIPE (a cohort step under the static deal: every sample's lists come from HBM, one item = ~8 KB per wave, and a
    // wave that requests an item's bytes, waits ~2 us for ALL of them -- the compiler's vmcnt(0) at the head of the row
    // loop --, computes for ~2 us and only then requests the next item's spends half its time waiting, and four waves
    // per SIMD do not cover that).  So the next item's run words are requested BEFORE this item's epilogue and its tile
    // record an item earlier still; this item's per-marker constants are requested at its top and awaited at its epilogue;
    // the first kPf rows of an item are walked in straight-line code, where the compiler counts the loads in flight
    // instead of draining them.
    // (compiled for the static deal only -- QUEUE == 0 --: with the way of dealing decided at run time the carried
    // registers of the two ways meet in copies after every item, and a copy of a register that is being loaded drains the
    // loads: the 4-point cohort step, whose kernel decides at run time, went 199 -> 228 us that way.  The same for the
    // 8-point shape on the work queue was measured and dropped twice: HISTORY.md.)
    constexpr bool PIPE = STREAM && ONEGRP && !LCACHE && QUEUE == 0;
    // the item after position (s_pos, rnd) of this wave's static deal
    auto static_item = [&](uint32_t s_pos, uint32_t rnd) -> uint32_t {
        if (have_sched) return s_pos < s_end ? (uint32_t)sch.item[s_pos] : nitem;
        return rnd * (uint32_t)nwave + ((rnd & 1u) ? (uint32_t)(nwave - 1 - wave) : (uint32_t)wave);
    };
    // (one group) item -> this lane's micro-tile
    auto tile_of = [&](uint32_t idx_, bool& have_) -> uint32_t {
        const uint32_t it_ = TPW * idx_ + (uint32_t)half;
        have_ = idx_ < nitem && (TPW == 1 || it_ < ntile_blk);
        return have_ ? owned_tile(OSH, blk, nblk, it_) : owned_tile(OSH, blk, nblk, 0u);
    };
    auto draw_item = [&]() -> uint32_t {                             // the workgroup's next item, whichever wave asks first
        uint32_t nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane(nxt);
    };
    auto issue_rows = [&](const vuint2 rec_, bool have_) {           // the first kPf rows of a tile (clamped to its own)
        const uint32_t cb_ = rec_.x * kRowBytes + (uint32_t)m * kLaneBytes;       // (32-bit byte offsets: see load_row)
        const int rows_ = have_ ? (PD ? (int)(((rec_.y >> 16) + (W16 ? 3u : 1u)) >> (W16 ? 2 : 1)) : (int)rec_.y) : 0;
        const int last_ = rows_ > 0 ? rows_ - 1 : 0;
#pragma unroll
        for (int j = 0; j < kPf; ++j) {
            g_cchar* a_ = reinterpret_cast<g_cchar*>(g_codes) + (cb_ + (uint32_t)(j < last_ ? j : last_) * kRowBytes);
            if constexpr (PD) w[j] = *reinterpret_cast<g_cuint*>(a_);
            else w[j] = *reinterpret_cast<g_cuint2*>(a_);
        }
    };
    // ---- one uint2 of run words (2 runs, or 4 of the 16-bit lists) into the accumulators ----
    // (first_tag: the tile's first row under PEEL -- its first run starts the sums from `init`, the marker's "other base"
    // constant, instead of adding to accumulators that were set to it: the same FMAs, twelve register moves fewer per item)
    auto walk_word = [&](const vuint2 w_cur, double* acc, const uint32_t my_tab, const uint32_t my_tab_w16, auto first_tag,
                         const double init) {
        constexpr bool kFirst = decltype(first_tag)::value;
        if constexpr ((kAblate & kAblNoReads) != 0) {        // (ablation build: the run words are consumed, the table is not read)
            acc[0] += __hiloint2double((int)((w_cur.x ^ w_cur.y) & 0x000f0000u) | 0x3ff00000, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < (W16 ? 4 : 2); ++j) {
            // one run: `n` reads of the same (class, quality) -> n * table row.  The run
            // word is {low half: byte offset of the row, high half: the top 16 bits of the
            // double n} -- one add and one and, no multiply, no int -> double conversion.
            // W16 (cohort steps): {low byte: dictionary index, next byte: n} -- two field
            // extractions, a multiply-add and a conversion, for half the bytes from HBM
            double n;
            uint32_t row_addr;
            if constexpr (W16) {
                // 16-bit run word = dictionary index | count code << 8, count code = (top 16 bits of double(n)) -
                // 0x3ff0 = exponent offset << 4 | top four mantissa bits (n <= 31: that IS the double).  The
                // high dword of 2 * n is 0x40 | code | 00 | 00: ONE byte permute (every VALU instruction costs
                // the same issue slot on gfx950 -- tools/ubench/int_rates.hip -- so what counts is their number:
                // round 3 spent five per run on the decode, and - 1 point per lane - six on the arithmetic);
                // the row address is one 24-bit multiply of the index byte, the table's base rides in the
                // ds_read's immediate offset.
                const uint32_t w2 = (j & 2) ? w_cur.y : w_cur.x;
                const uint32_t idx = (j & 1) ? ((w2 >> 16) & 0xffu) : (w2 & 0xffu);
                const uint32_t hi = __builtin_amdgcn_perm(0x40000000u, w2, (j & 1) ? 0x07030c0cu : 0x07010c0cu);
                n = __hiloint2double((int)hi, 0);
                row_addr = my_tab_w16 + __umul24(idx, (uint32_t)row_bytes);
            } else {
                const uint32_t rw = j ? w_cur.y : w_cur.x;
                n = __hiloint2double((int)(rw & 0xffff0000u), 0);
                row_addr = my_tab + (rw & 0xffffu);
            }
            lds_cdouble2* row = reinterpret_cast<lds_cdouble2*>(row_addr);
#pragma unroll
            for (int i = 0; i < 3 * BTL; ++i) {
                const vdouble2 t = row[i];
                acc[2 * i] = fma(n, t.x, (kFirst && j == 0) ? init : acc[2 * i]);
                acc[2 * i + 1] = fma(n, t.y, (kFirst && j == 0) ? init : acc[2 * i + 1]);
            }
        }
    };
    // ---- PD: one word of the list = two steps.  A step multiplies the six products of every point by ONE table row; class alt
    // (the rows behind the tile's ref rows: alt_tag) by the same row read the other way round, pair p <-> 5 - p.  No count, no
    // conversion: one SDWA add per step and the multiplies ----
    auto walk_pd = [&](const uint32_t w_cur, double* acc, const uint32_t my_tab, auto alt0_tag, auto alt1_tag, auto first_tag) {
        constexpr bool kAlt0 = decltype(alt0_tag)::value, kAlt1 = decltype(alt1_tag)::value, kFirst = decltype(first_tag)::value;
        if constexpr ((kAblate & kAblNoReads) != 0) {        // (ablation build: the steps are consumed, the table is not read)
            acc[0] = (kFirst ? 1.0 : acc[0]) * __hiloint2double((int)((w_cur ^ (w_cur >> 16)) & 0x000fu) | 0x3ff00000, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool kAlt = j ? kAlt1 : kAlt0;        // (a compile-time value after unrolling)
            // (an alt step's offset points at the mirror image the narrow shapes keep in the row's second half: this shape
            // has eight points there -- it takes the offset off and names its products the other way round)
            const uint32_t row_addr = (kAlt ? my_tab - (uint32_t)kPdAltOffset : my_tab) + (j ? (w_cur >> 16) : (w_cur & 0xffffu));
            lds_cdouble2* row = reinterpret_cast<lds_cdouble2*>(row_addr);
#pragma unroll
            for (int i = 0; i < 3 * BTL; ++i) {
                const vdouble2 t = row[i];
                const int pt = i / 3, q2 = 2 * (i - 3 * pt);
                const int a0 = pt * 6 + (kAlt ? 5 - q2 : q2), a1 = pt * 6 + (kAlt ? 4 - q2 : q2 + 1);
                // (a tile's first step: the products START as the row -- the marker's constant is multiplied in once, at the end)
                if constexpr ((kAblate & kAblNoMul) != 0) {      // (ablation build: the rows are read and consumed, not multiplied)
                    asm volatile("" ::"v"(t.x), "v"(t.y));
                    if (kFirst && j == 0) { acc[a0] = t.x; acc[a1] = t.y; }
                    continue;
                }
                acc[a0] = (kFirst && j == 0) ? t.x : acc[a0] * t.x;
                acc[a1] = (kFirst && j == 0) ? t.y : acc[a1] * t.y;
            }
        }
    };
    // ---- PD, shapes of at most four points per row: ONE kind of step -- an alt step's offset points at the row's mirror image ----
    auto walk_pd1 = [&](const uint32_t w_cur, double* acc, const uint32_t my_tab, auto first_tag) {
        constexpr bool kFirst = decltype(first_tag)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t row_addr = my_tab + (j ? (w_cur >> 16) : (w_cur & 0xffffu));
            lds_cdouble2* row = reinterpret_cast<lds_cdouble2*>(row_addr);
#pragma unroll
            for (int i = 0; i < 3 * BTL; ++i) {
                const vdouble2 t = row[i];
                acc[2 * i] = (kFirst && j == 0) ? t.x : acc[2 * i] * t.x;
                acc[2 * i + 1] = (kFirst && j == 0) ? t.y : acc[2 * i + 1] * t.y;
            }
        }
    };
    // ---- PD, the cohort steps' 8-BIT lists (W16; DeviceLayout::codes16 of a probability-domain context): a word = FOUR steps, a
    // byte = the row's index; whether a step is an alt step -- its row's mirror image, kPdAltOffset into the row -- follows from
    // its position in the tile (the ref steps come first: step >= s1).  Half the list bytes of a step that is bound by what it
    // streams; three vector instructions more per step (index x row stride, the comparison, the choice of the base) ----
    auto walk_pd8 = [&](const uint32_t w_cur, double* acc, const uint32_t my_tab, auto first_tag, const int step0, const int s1) {
        constexpr bool kFirst = decltype(first_tag)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t idx = (w_cur >> (8 * j)) & 0xffu;
            const uint32_t base = (step0 + j >= s1) ? my_tab + (uint32_t)kPdAltOffset : my_tab;
            const uint32_t row_addr = base + __umul24(idx, (uint32_t)row_bytes);
            lds_cdouble2* row = reinterpret_cast<lds_cdouble2*>(row_addr);
#pragma unroll
            for (int i = 0; i < 3 * BTL; ++i) {
                const vdouble2 t = row[i];
                acc[2 * i] = (kFirst && j == 0) ? t.x : acc[2 * i] * t.x;
                acc[2 * i + 1] = (kFirst && j == 0) ? t.y : acc[2 * i + 1] * t.y;
            }
        }
    };
    // ---- per-marker epilogue: a marker's likelihood as (mantissa, exponent) per point, from its six sums per point ----
    auto marker_lk = [&](const bool live, const uint32_t pos, const double* acc, const double e0, const double e1, const double e2,
                         const double* udr, const double mur, const uint32_t my_ptq, double* lk_m, int* lk_e, const double cst_pd) {
#pragma unroll
        for (int t = 0; t < BTL; ++t) { lk_m[t] = 1.0; lk_e[t] = 0; }
        if constexpr ((kAblate & kAblNoEpi) != 0) {          // (ablation build: the sums and the constants are consumed, nothing is computed from them)
            double z = e0 + e1 + e2 + mur + udr[0] + udr[1] + udr[2] + udr[3];
            for (int i = 0; i < BTL * 6; ++i) z += acc[i];
            if (z == 12345.678) lk_m[0] = 0.75;
            return;
        }
        if (live) {
            double af1[BTL], af2[BTL];
            if (known_af_p) {
                const double a = g_kaf[pos];
#pragma unroll
                for (int t = 0; t < BTL; ++t) af1[t] = af2[t] = a;
            } else {
                // h:251-267: AF = (sum_k UD[i][k]*pc[k] + mean) / 2, same order of the k terms; each
                // term enters with one rounding (FMA) where the reference's x86-64 build rounds the
                // product first: AF differs by <= k/2 ulp, far inside the 1e-12 the tests hold the LLK to
#pragma unroll
                for (int t = 0; t < BTL; ++t) af1[t] = af2[t] = 0.;
                auto project = [&](int kk, double uu) {
                    if constexpr (BTL == 2) {              // both points' coefficient in one ds_read_b128
                        const vdouble2 c1 = *reinterpret_cast<lds_cdouble2*>(my_ptq + (uint32_t)kk * 16u);
                        const vdouble2 c2 = *reinterpret_cast<lds_cdouble2*>(my_ptq + (uint32_t)(k + kk) * 16u);
                        af1[0] = fma(uu, c1.x, af1[0]); af1[1] = fma(uu, c1.y, af1[1]);
                        af2[0] = fma(uu, c2.x, af2[0]); af2[1] = fma(uu, c2.y, af2[1]);
                    } else {
                        af1[0] = fma(uu, *reinterpret_cast<lds_cdouble*>(my_ptq + (uint32_t)kk * 8u), af1[0]);
                        af2[0] = fma(uu, *reinterpret_cast<lds_cdouble*>(my_ptq + (uint32_t)(k + kk) * 8u), af2[0]);
                    }
                };
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    if (kk < k) project(kk, udr[kk]);                       // rows loaded before the read loop
                for (int kk = 4; kk < k; ++kk) project(kk, g_ud[(size_t)kk * mp + pos]);
                const double mu = mur;
#pragma unroll
                for (int t = 0; t < BTL; ++t) {
                    af1[t] += mu; af1[t] /= 2.0;
                    af2[t] += mu; af2[t] /= 2.0;
                }
            }
#pragma unroll
            for (int t = 0; t < BTL; ++t) {
                double gf[3], gf2[3];
                initial_gf(af1[t], gf);
                initial_gf(af2[t], gf2);
                const double* a = acc + t * 6;
                // h:307-311: lk = sum_{g1,g2} exp(A[g1][g2]) GF[g1] GF2[g2], factored as
                // sum_g1 GF[g1] * (sum_g2 exp(A[g1][g2]) GF2[g2]) with FMAs (12 operations instead
                // of 27; the rounding differs from the reference's term-by-term sum at the 1e-16
                // level, like the marker summation order does).  The three g1==g2 exponentials do
                // not depend on (alpha, PC) and were taken at context creation.
                double x01, x02, x10, x12, x20, x21;
                const double a_min = PD ? 0.0 : vmin2_f64(vmin2_f64(vmin2_f64(a[0], a[1]), vmin2_f64(a[2], a[3])), vmin2_f64(a[4], a[5]));
                if constexpr (PD) {
                    // the products ARE the six likelihoods: no exponential (probability domain)
                    x01 = a[0]; x02 = a[1]; x10 = a[2]; x12 = a[3]; x20 = a[4]; x21 = a[5];
                } else
                if (__builtin_expect(a_min < -708.0, 0)) {      // (wave-divergent, and never taken on whole-genome depths: exp_nonpos)
                    x01 = exp_nonpos<ESH, true>(a[0], etab_lane); x02 = exp_nonpos<ESH, true>(a[1], etab_lane);
                    x10 = exp_nonpos<ESH, true>(a[2], etab_lane); x12 = exp_nonpos<ESH, true>(a[3], etab_lane);
                    x20 = exp_nonpos<ESH, true>(a[4], etab_lane); x21 = exp_nonpos<ESH, true>(a[5], etab_lane);
                } else {
                    x01 = exp_nonpos<ESH>(a[0], etab_lane); x02 = exp_nonpos<ESH>(a[1], etab_lane);
                    x10 = exp_nonpos<ESH>(a[2], etab_lane); x12 = exp_nonpos<ESH>(a[3], etab_lane);
                    x20 = exp_nonpos<ESH>(a[4], etab_lane); x21 = exp_nonpos<ESH>(a[5], etab_lane);
                }
                double lk;
                if constexpr (PD) {
                    // the six products lack the marker's constant exp(c_other) (`cst`): it multiplies their part of the sum
                    // once; the three g1 == g2 terms hold it since context creation
                    const double o0 = fma(x02, gf2[2], x01 * gf2[1]);
                    const double o1 = fma(x12, gf2[2], x10 * gf2[0]);
                    const double o2 = fma(x21, gf2[1], x20 * gf2[0]);
                    const double off = fma(o2, gf[2], fma(o1, gf[1], o0 * gf[0]));
                    const double dg = fma(e2 * gf2[2], gf[2], fma(e1 * gf2[1], gf[1], (e0 * gf2[0]) * gf[0]));
                    lk = fma(cst_pd, off, dg);
                    if (__builtin_expect(lk < 0x1p-960, 0)) {
                        x01 *= cst_pd; x02 *= cst_pd; x10 *= cst_pd; x12 *= cst_pd; x20 *= cst_pd; x21 *= cst_pd;
                    }
                } else {
                const double s0 = fma(x02, gf2[2], fma(x01, gf2[1], e0 * gf2[0]));
                const double s1 = fma(x12, gf2[2], fma(e1, gf2[1], x10 * gf2[0]));
                const double s2 = fma(e2, gf2[2], fma(x21, gf2[1], x20 * gf2[0]));
                lk = fma(s2, gf[2], fma(s1, gf[1], s0 * gf[0]));
                }
                // Near the bottom of the double range the factoring is no longer harmless: products with
                // the priors underflow at different places, and the reference's rule below (h:310-311,
                // "add log(lk) only if lk > 0") turns that into a marker counted or dropped, 745 units
                // of LLK apart (markers with ~1000 reads).  There the sum is redone term by term in the
                // reference's own order (g1 outer, g2 inner, h:307-309), as round 1 did everywhere;
                // wave-divergent, and never taken on whole-genome depths.
                if (__builtin_expect(lk < 0x1p-960, 0)) {
                    double r = 0;
                    r += e0 * gf[0] * gf2[0];
                    r += x01 * gf[0] * gf2[1];
                    r += x02 * gf[0] * gf2[2];
                    r += x10 * gf[1] * gf2[0];
                    r += e1 * gf[1] * gf2[1];
                    r += x12 * gf[1] * gf2[2];
                    r += x20 * gf[2] * gf2[0];
                    r += x21 * gf[2] * gf2[1];
                    r += e2 * gf[2] * gf2[2];
                    lk = r;
                }
                // the reference adds log(lk) only if lk > 0 (h:310-311): a dropped marker is
                // the factor 1
                lk = lk > 0 ? lk : 1.0;
                lk_m[t] = __builtin_amdgcn_frexp_mant(lk);
                lk_e[t] = __builtin_amdgcn_frexp_exp(lk);
            }
        }
    };
    vuint2 rec_nx;                                        // PIPE: the record of the item about to be processed ...
    rec_nx.x = rec_nx.y = 0u;
    double cst_nx = 0.0;                                  // ... and its markers' "other base" constants (the accumulators start from them)
    auto other_const = [&](uint32_t mt_, bool have_) -> double {
        const uint32_t pos_ = mt_ * (uint32_t)kMtMarkers + (uint32_t)m;
        const uint32_t boff_ = ((have_ && pos_ < (uint32_t)L.num_active) ? pos_ : 0u) * 8u;
        return *reinterpret_cast<g_cdouble*>(reinterpret_cast<__attribute__((address_space(1))) const char*>(g_ediag) + boff_);
    };
    // The workgroup of the resident kernel's control wave (wave 0, busy with the simplex during the tile phase): the waves w, w + 4,
    // w + 8, ... of a workgroup run on one SIMD (tools/ubench/wave_simd.hip), so the waves 4, 8, 12 share the control wave's.
    // They come LAST in the order in which the waves take their first items: a search round has fewer items than waves (13 for
    // 15 at C3), and the idle waves are then the control wave's neighbours instead of the last two.  Same items, same slots: the
    // same bits.  OptimizeLLK 6.09 -> 6.04 ms on the same box.
    const int spare_rank = (wave & 3) ? (wave >> 2) * 3 + (wave & 3) - 1 : (nwave - (nwave >> 2)) + (wave >> 2) - 1;
    // FEWER ITEMS THAN WAVES (a search round, the launches of one to four points): which wave takes which item decides how
    // the rows are spread over the CU's four SIMDs (waves w, w + 4, w + 8, w + 12 share one).  The items are depth-sorted, so
    // "wave w takes item w" gives SIMD 0 the deepest item of every four -- and a fourth item when there are thirteen -- and the
    // round ends with that SIMD's last wave.  Instead: one SIMD takes the ceil(n / 4) SHALLOWEST items (in the control wave's
    // workgroup: its SIMD, and one item fewer), the others share the rest in snake order; up to eight items: a snake over all
    // four.  On the queue a tile's product has its own slot whichever wave computes it: the same bits.  OptimizeLLK at C3
    // 5.86 -> 5.60 ms on one box (three alternating rounds; either workgroup kind alone: no gain -- the round ends with the
    // other kind), the control wave's SIMD with ONE item as before: no gain; tables for 13 items built by hand: the same 5.59-5.61.
    int deal_rank = hook_blk ? spare_rank : wave;
    if (VB2_SIMD_DEAL && dyn && nwave == 16 && ONEGRP && nitem <= (uint32_t)(hook_blk ? 15 : 16)) {
        const int n = (int)nitem, sd = wave & 3, sj = wave >> 2;
        int r;
        if (!hook_blk && n <= 8) r = sj == 0 ? sd : sj == 1 ? 7 - sd : n;
        else {
            const int c0 = ((n + 3) >> 2) - (hook_blk ? 1 : 0), nrest = n - c0;
            if (sd == 0) {
                const int j0 = sj - (hook_blk ? 1 : 0);
                r = (j0 >= 0 && j0 < c0) ? nrest + j0 : n;
            } else {
                r = 3 * sj + ((sj & 1) ? 3 - sd : sd - 1);
                if (r >= nrest) r = n;
            }
        }
        deal_rank = r < n ? r : n;
    }
    const uint32_t idx_first = hook_mine ? nitem
                               : have_sched ? (s_i < s_end ? (uint32_t)sch.item[s_i] : nitem)
                                            : (uint32_t)deal_rank;
    if (PIPE && idx_first < nitem) {
        bool h0;
        const uint32_t mt0 = tile_of(idx_first, h0);
        rec_nx = g_rec[mt0];
        cst_nx = other_const(mt0, h0);
        issue_rows(rec_nx, h0);
    }
#ifdef VB2_ITEM_PROF     // (profiling build, with VB2_WITH_STAMPS: where a wave's time per work item goes -- tools/item_prof.py)
    unsigned long long ip_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (uint32_t idx = idx_first; idx < nitem;) {
        VB2_IP_T(ip_t0);
        // grp = idx / nunit without the ~25-instruction integer division: float estimate (exact for
        // these magnitudes up to one) and a correction step
        uint32_t grp, unit;
        if (MODE == 2 && QUEUE == 1 && !ONEGRP) {
            // The queue walks the tiles deepest first and every tile's point groups side by side: the waves that hold a tile's items at
            // the same time read the same run words and per-marker constants (one trip to L2 instead of up to six), and the order is
            // still longest-first.  The slots and their order are those of a group-by-group walk: the same bits.
            // (idx is uniform and so is the reciprocal: a scalar multiply-high -- exact for idx < 2^32 / ngrp -- where the float
            // division of the uniform pair cost fifteen vector instructions per item)
            unit = ngrp > 1 ? __umulhi(idx, ngrp_magic) : idx;
            grp = idx - unit * (uint32_t)ngrp;
        } else {
            grp = ngrp == 1 ? 0u : (uint32_t)(((float)idx + 0.5f) * inv_nunit);
            if (ngrp != 1) {
                if (grp * nunit > idx) --grp;
                else if ((grp + 1) * nunit <= idx) ++grp;
            }
            unit = idx - grp * nunit;
        }
        // index in this workgroup's tile list (SPLIT: the units alternate between the two virtual blocks -- both lists are
        // in descending order of rows, so the queue still walks longest first)
        const uint32_t vs = SPLIT ? unit % (uint32_t)NVS : 0u;
        const uint32_t it = SPLIT ? unit / (uint32_t)NVS : TPW * unit + (uint32_t)half;
        const bool have_tile = SPLIT ? it < nt_of(vs) : (TPW == 1 || it < ntile_blk);   // TPW > 1: the list's end may leave lanes idle
        const uint32_t mt = owned_tile(OSH, vb0 + vs * vstep, nblk, have_tile ? it : 0u);
        const uint32_t my_tab = tab_addr + (grp * (uint32_t)nrow * (uint32_t)row_bytes + (uint32_t)g * (6 * BTL * 8));
        const uint32_t my_ptq = ptq_addr + (grp * SLOTS + (uint32_t)g) * (uint32_t)(2 * k * BTL * 8);
        // (one slot, one group: the table's address is the constant behind the exp table -- this file has no static LDS --
        // and the compiler folds it into the reads' immediate offsets)
        const uint32_t my_tab_w16 = (SLOTS == 1 && ONEGRP) ? (uint32_t)((kEtabDoubles + kLogTabDoubles) * sizeof(double)) : my_tab;
        while (!dyn && grp_wave < grp) {                     // wave-uniform
            flush_wave(grp_wave);
            ++grp_wave;
        }
        // {first row, rows}; one scalar load when TPW == 1.  LCACHE: {LDS byte address of the tile's first row, rows}
        vuint2 rec;
        if constexpr (LCACHE) rec = *reinterpret_cast<lds_cuint2v*>(hook.cache_rec() + (have_tile ? it : 0u) * 8u);
        else if (PIPE) rec = rec_nx;
        else rec = g_rec[mt];
        VB2_IP_USE(rec.x);
        VB2_IP_T(ip_t1);
        // PIPE: the next item's record, requested now, needed when this item's rows have been walked
        uint32_t idx_next = nitem;
        bool have_next = false;
        uint32_t mt_next = blk;
        vuint2 rec_n2;
        rec_n2.x = rec_n2.y = 0u;
        if (PIPE) {
            idx_next = static_item(s_i + 1, round + 1);
            mt_next = tile_of(idx_next, have_next);
            rec_n2 = g_rec[mt_next];
        }
        // per-marker constants: issued now, consumed after the read loop
        const uint32_t pos = mt * (uint32_t)kMtMarkers + (uint32_t)m;      // position in sorted order
        const bool live = have_tile && pos < (uint32_t)L.num_active;
        // One 32-bit byte offset for all of a marker's constants, each array's base a scalar: the loads take the
        // base-plus-offset form and cost no 64-bit vector address arithmetic (nine add pairs per item before).
        const uint32_t boff = (live ? pos : 0u) * 8u;        // (m_pad < 2^29: vb2_ctx_create refuses more markers)
        auto at = [&](g_cdouble* base_) -> double {
            // (the base through an opaque scalar register: else the compiler re-associates base + offset + column stride
            // and adds the uniform stride in 64-bit vector arithmetic)
            unsigned long long b_ = (unsigned long long)base_;
            asm volatile("" : "+s"(b_));
            return *reinterpret_cast<g_cdouble*>(reinterpret_cast<__attribute__((address_space(1))) const char*>(b_) + boff);
        };
        const double cst = PIPE ? cst_nx : at(g_ediag);
        const double e0 = at(g_ediag + mp), e1 = at(g_ediag + 2 * mp), e2 = at(g_ediag + 3 * mp);

        // (the panel row of the marker too: up to four UD columns and the mean)
        double udr[4], mur = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) udr[kk] = (!known_af_p && kk < k) ? at(g_ud + (size_t)kk * mp) : 0.0;
        if (!known_af_p) mur = at(g_mu);

        // the six off-diagonal sums start from the marker's "other base" constant (it is part of
        // every genotype pair's sum, h:299-303), so the epilogue needs no separate addition
        // PEEL (one tile per wave, lists from L2: the launches of many points): a tile's row count is uniform, and its
        // first row is walked outside the row loop
        constexpr bool PEEL = TPW == 1 && !W16 && !PIPE && !LCACHE && QUEUE == 1 && (kAblate & kAblNoReads) == 0;
        double acc[BTL * 6];
        if constexpr (!PEEL && !PD) {
#pragma unroll
            for (int i = 0; i < BTL * 6; ++i) acc[i] = cst;
        }

        // ---- per-read accumulate (h:288-303), one step per run ----
        // Rows are prefetched kPrefetch deep: with one sample its pileup sits in L2, but a cohort
        // launch streams every sample's rows from HBM once, and two rows ahead (~0.6 us of work)
        // does not cover that latency.  The loads are unconditional: a prefetch past the tile's
        // last row reads the next tile's rows (never consumed), and the array ends in
        // 2 * kPrefetch padding rows (context.cpp).
        // The read loop is bound by the LDS pipe, the epilogue by the FP64 VALU, and the waves of a CU
        // are spread over both phases at any time: waves in the read loop get issue priority, so that
        // their ds_reads go out as soon as they can and the LDS pipe stays busy, while the epilogue
        // waves fill the VALU slots in between (-0.8 % per launch measured; the other way round: +0.8 %)
        // A search round (the four-point shape on the work queue) has at most one item per wave, and the round ends with the
        // wave that holds the deepest tiles (17 rows at C3 against a median of 9: the first items of the low-numbered
        // workgroups): the deeper a wave's item -- the lower its index -- the higher its priority on its SIMD (the three
        // deepest items of a workgroup are on three different SIMDs: the first deal above).  OptimizeLLK 5.98 -> 5.90 ms on the same box.
        if (MODE == 3 && ONEGRP && dyn) {
            if (idx < 4u) __builtin_amdgcn_s_setprio(3);
            else if (idx < 8u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        } else
            __builtin_amdgcn_s_setprio(1);
        // the run words by 32-bit byte offsets from the array's (scalar) base: the list of one sample stays below 4 GiB
        // (Context::create refuses more rows), and a row's address costs no 64-bit vector arithmetic
        const uint32_t cbase = LCACHE ? 0u : rec.x * kRowBytes + (uint32_t)m * kLaneBytes;
        const uint32_t crow = rec.x + (uint32_t)m * kLaneBytes;      // (LCACHE: this lane's word of the tile's first row, LDS)
        // a scalar when TPW == 1.  PD: {ref steps | all steps << 16}: the tile's markers have their ref steps at [0, s1) and
        // their alt steps at [s1, s2), two steps to a row (walk_pd)
        // (one tile per wave: both are the same in every lane, and the compiler is told so -- the branches on them are scalar)
        const int steps_ref_v = PD ? (have_tile ? (int)(rec.y & 0xffffu) : 0) : 0;
        const int rows_v = have_tile ? (PD ? (int)(((rec.y >> 16) + (W16 ? 3u : 1u)) >> (W16 ? 2 : 1)) : (int)rec.y) : 0;
        const int steps_ref = (PD && TPW == 1) ? __builtin_amdgcn_readfirstlane(steps_ref_v) : steps_ref_v;
        const int rows = (PD && TPW == 1) ? __builtin_amdgcn_readfirstlane(rows_v) : rows_v;
        // (ONE sample's pileup sits in L2, and a deep prefetch costs more than it hides there: the loads run past the
        // tile's last row -- up to kPf useless row loads per item of 6..16 rows -- and every block of kPf rows begins
        // by waiting for all of them.  Measured on one box, depth 8 / 4 / 2: 48-point launch 74.9 / 74.4 / 76.6 us;
        // OptimizeLLK at C3 (one 4-point item per wave and round) 7.8 / 7.55 / 7.35 ms, at 10 000 markers 2.39 /
        // 2.31 / 2.43 ms.)
        // (STREAM: rows past the tile's last are the NEXT tiles' -- another workgroup's, at another time: fetched here
        // they came from HBM twice, 573 MB instead of 356 MB per one-point step of 32 C3 samples (FETCH_SIZE, round 3).
        // The row index is clamped to the tile's last row instead: the same cache line again, no new bytes.)
        const int last_row = rows > 0 ? rows - 1 : 0;
        auto load_row = [&](int j) -> RowWord {              // row j of this lane's run words
            if constexpr (PD && (kAblate & kAblNoRowLoads) != 0) {      // (ablation build: no trip to memory for the steps -- made-up rows)
                const uint32_t r0 = ((uint32_t)m * 7u + (uint32_t)j * 2u) % (uint32_t)nrow, r1 = ((uint32_t)m * 7u + (uint32_t)j * 2u + 1u) % (uint32_t)nrow;
                return (r0 * (uint32_t)row_bytes) | ((r1 * (uint32_t)row_bytes) << 16);
            }
            if constexpr (LCACHE && PD) return *reinterpret_cast<lds_cuint*>(crow + (uint32_t)j * kRowBytes);
            else if constexpr (LCACHE) return *reinterpret_cast<lds_cuint2v*>(crow + (uint32_t)j * kRowBytes);
            else {
                g_cchar* a_ = reinterpret_cast<g_cchar*>(g_codes) + (cbase + (uint32_t)(STREAM ? (j < last_row ? j : last_row) : j) * kRowBytes);
                if constexpr (PD) return *reinterpret_cast<g_cuint*>(a_);
                else return *reinterpret_cast<g_cuint2*>(a_);
            }
        };
        if (!PIPE) {
#pragma unroll
            for (int j = 0; j < kPf; ++j) w[j] = load_row(j);
        }
        auto walk_block = [&](const int s0, const bool refill, auto first_tag) {     // rows s0 .. s0 + kPf - 1 of the tile
            constexpr bool kFirstBlock = decltype(first_tag)::value;
#pragma unroll
            for (int u = 0; u < kPf; ++u) {
                if (s0 + u >= rows) break;
                const RowWord w_cur = w[u];
                if (refill) w[u] = load_row(s0 + u + kPf);
                if constexpr (PD) {
                    if constexpr (W16) {
                        if (kFirstBlock && u == 0) walk_pd8(w_cur, acc, my_tab, std::true_type(), (s0 + u) * 4, steps_ref);
                        else walk_pd8(w_cur, acc, my_tab, std::false_type(), (s0 + u) * 4, steps_ref);
                    } else {
                        if (kFirstBlock && u == 0) walk_pd1(w_cur, acc, my_tab, std::true_type());
                        else walk_pd1(w_cur, acc, my_tab, std::false_type());
                    }
                } else {
                    if (kFirstBlock && u == 0) walk_word(w_cur, acc, my_tab, my_tab_w16, std::true_type(), cst);
                    else walk_word(w_cur, acc, my_tab, my_tab_w16, std::false_type(), cst);
                }
            }
        };
        VB2_IP_USE(w[0].x);
        VB2_IP_T(ip_t2);
        if constexpr (PD && TPW > 1 && STREAM) {
            // (cohort steps of the narrow shapes: one kind of step -- walk_pd1 --, so the run words' loops serve, with their rows
            // requested an item ahead; a lane's first step IS its products, a lane without rows has none.  A single sample's
            // launches and search rounds -- lists in L2 / LDS -- take the two loops below: OptimizeLLK 5.41 against 5.58 ms)
            if (__any(rows == 0)) {
                if (rows == 0) {
#pragma unroll
                    for (int i = 0; i < BTL * 6; ++i) acc[i] = 1.0;
                }
            }
            if constexpr (PIPE) {
                const bool more = __any(rows > kPf);
                walk_block(0, more, std::true_type());
                if (more)
                    for (int s0 = kPf; s0 < rows; s0 += kPf) walk_block(s0, true, std::false_type());
                issue_rows(rec_n2, have_next);
                cst_nx = other_const(mt_next, have_next);
                rec_nx = rec_n2;
            } else {
                walk_block(0, true, std::true_type());
                for (int s0 = kPf; s0 < rows; s0 += kPf) walk_block(s0, true, std::false_type());
            }
        } else if constexpr (PD) {
            // The tile's ref rows, then its alt rows: TWO loops of one body each (as one loop with a test per row the
            // compiler kept the twelve products in different registers on the two paths and moved them all where the paths
            // meet).  The rows in flight are a window of kPf words that moves up one row per iteration.  In the paired shapes
            // the bounds are per half of the wave: the two tiles are neighbours of the sorted order and nearly always agree.
            int r = 0;
            // The rows in flight are a RING of kPf words with fixed slots -- row r in slot r % kPf -- and every step names its
            // slot at compile time.  (Before, the window MOVED up a register per row: the move of the newest word is a use of
            // it, so every row began by waiting for the load issued one row earlier.  Worth 0.6 % of a 48-point launch: the
            // lists' loads hide behind the table reads either way -- with made-up steps and no loads at all, VB2_ABLATE = 128,
            // the launch is no shorter; profiles/r06/ab_ablations.txt.)
            static_assert(kPf == 2, "the ring below has two slots");
            auto row_step = [&](auto slot_tag, auto alt0_tag, auto alt1_tag, auto first_tag) {
                constexpr int kSlot = decltype(slot_tag)::value;
                const uint32_t cur = w[kSlot];
                w[kSlot] = load_row(r + kPf);
                walk_pd(cur, acc, my_tab, alt0_tag, alt1_tag, first_tag);
                ++r;
            };
            const std::integral_constant<int, 0> kS0;
            const std::integral_constant<int, 1> kS1;
            // (the first row's first step IS the products: no initial values, no multiplies).  Rows [0, s1 / 2) hold two ref
            // steps, the rows behind two alt steps; if s1 is odd, row s1 / 2 holds the last ref and the first alt step.
            const int rows_ra = steps_ref >> 1;
            const std::false_type kRef;
            const std::true_type kAltT, kFirstT;
            const std::false_type kNotFirst;
            if (rows > 0) {
                if (steps_ref >= 2) row_step(kS0, kRef, kRef, kFirstT);
                else if (steps_ref == 1) row_step(kS0, kRef, kAltT, kFirstT);
                else row_step(kS0, kAltT, kAltT, kFirstT);
            } else {
#pragma unroll
                for (int i = 0; i < BTL * 6; ++i) acc[i] = 1.0;
            }
            // rows [r, end) of one kind: an odd row first, then pairs, then an even one
            auto rows_of_kind = [&](const int end, auto alt_tag) {
                if (r < end && (r & 1)) row_step(kS1, alt_tag, alt_tag, kNotFirst);
                while (r + 1 < end) {
                    row_step(kS0, alt_tag, alt_tag, kNotFirst);
                    row_step(kS1, alt_tag, alt_tag, kNotFirst);
                }
                if (r < end) row_step(kS0, alt_tag, alt_tag, kNotFirst);
            };
            rows_of_kind(rows_ra, kRef);
            if ((steps_ref & 1) && r == rows_ra && r < rows) {
                if (r & 1) row_step(kS1, kRef, kAltT, kNotFirst);
                else row_step(kS0, kRef, kAltT, kNotFirst);
            }
            rows_of_kind(rows, kAltT);
            if constexpr (PIPE) {
                issue_rows(rec_n2, have_next);
                cst_nx = other_const(mt_next, have_next);
                rec_nx = rec_n2;
            }
        } else if constexpr (PIPE) {
            // the first kPf rows outside any loop (loads in flight are counted, not drained); an item with more
            // rows -- wide quality alphabets -- refills the ring as before
            const bool more = __any(rows > kPf);
            walk_block(0, more, std::false_type());
            if (more)
                for (int s0 = kPf; s0 < rows; s0 += kPf) walk_block(s0, true, std::false_type());
            // this item's rows are walked: the next item's go out now, under the epilogue
            issue_rows(rec_n2, have_next);
            cst_nx = other_const(mt_next, have_next);
            rec_nx = rec_n2;
        } else if constexpr (PEEL) {
            if (rows > 0) {
                walk_block(0, true, std::true_type());
                for (int s0 = kPf; s0 < rows; s0 += kPf) walk_block(s0, true, std::false_type());
            } else {
#pragma unroll
                for (int i = 0; i < BTL * 6; ++i) acc[i] = cst;
            }
        } else {
            for (int s0 = 0; s0 < rows; s0 += kPf) walk_block(s0, true, std::false_type());
        }

        __builtin_amdgcn_s_setprio(0);
#ifdef VB2_STAMP_CTRL
        if (stamps && lane == 0 && wave == 1 && !hook_blk) stamps[3] = wall_clock64();
#else
        if (stamps && lane == 0 && wave == 1) stamps[3] = wall_clock64();      // wave 1: out of its (first) read loop
#endif
        VB2_IP_USE(acc[0]); VB2_IP_USE(acc[BTL * 6 - 1]);
        VB2_IP_T(ip_t3);
        VB2_IP_USE(e0); VB2_IP_USE(e1); VB2_IP_USE(e2); VB2_IP_USE(mur); VB2_IP_USE(udr[0]); VB2_IP_USE(udr[3]);
        VB2_IP_T(ip_t4);
        // ---- per-marker epilogue: this marker's likelihood as (mantissa, exponent) per point ----
        double lk_m[BTL];
        int lk_e[BTL];
        marker_lk(live, pos, acc, e0, e1, e2, udr, mur, my_ptq, lk_m, lk_e, cst);
        VB2_IP_USE(lk_m[0]); VB2_IP_USE(lk_m[BTL - 1]);
        VB2_IP_T(ip_t5);
        if (!dyn) {
#pragma unroll
            for (int t = 0; t < BTL; ++t) {               // running product, renormalised lazily
                wave_prod[t].m *= lk_m[t];
                wave_prod[t].e += lk_e[t];
            }
            if (++nfactor == kLazyRenorm) renorm_wave();
            if (have_sched) {
                ++s_i;
                idx = PIPE ? idx_next : (s_i < s_end ? (uint32_t)sch.item[s_i] : nitem);
                continue;
            }
            // next round, direction reversed: the items are depth-sorted, and a plain deal would
            // hand wave 0 the deepest tile of every round
            ++round;
            idx = round * (uint32_t)nwave + ((round & 1u) ? (uint32_t)(nwave - 1 - wave) : (uint32_t)wave);
            continue;
        }
        // every MICRO-TILE's product goes to its own slot (not the item's): the slots and the
        // order they are multiplied in are then the same for every wave shape, so a point's value
        // does not depend on whether it was evaluated alone, among four, or in a group of eight.
        // Butterfly over the 16 lanes that share slot g: mantissas multiply (16 factors in
        // [0.5, 1): no underflow), exponents add as integers (three dwords per exchange, not four)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const int partner = lane_of(m ^ off, g4);
#pragma unroll
            for (int t = 0; t < BTL; ++t) {
                lk_m[t] *= lane_xchg_f64(lk_m[t], off, partner);
                lk_e[t] += lane_xchg_i32(lk_e[t], off, partner);
            }
        }
        if (m == 0 && have_tile) {
#pragma unroll
            for (int t = 0; t < BTL; ++t) {
                const size_t o = ((((size_t)grp * NVS + vs) * ntile_blk + it) * NP + g * BTL + t) * 2;
                tile_llk[o] = lk_m[t];
                tile_llk[o + 1] = (double)lk_e[t];
            }
        }
        VB2_IP_T(ip_t6);
        // next work item of this workgroup, whichever wave gets there first
        idx = draw_item();
#ifdef VB2_ITEM_PROF
        {
            const unsigned long long ip_t7 = __builtin_readcyclecounter();
            ip_sum[0] += ip_t1 - ip_t0; ip_sum[1] += ip_t2 - ip_t1; ip_sum[2] += ip_t3 - ip_t2; ip_sum[3] += ip_t4 - ip_t3;
            ip_sum[4] += ip_t5 - ip_t4; ip_sum[5] += ip_t6 - ip_t5; ip_sum[6] += ip_t7 - ip_t6; ip_sum[7] += 1;
        }
#endif
    }
#ifdef VB2_ITEM_PROF
    if (VB2_STAMPS_OF(L) && lane == 0 && dyn)
        for (int i = 0; i < 8; ++i) atomicAdd(&VB2_STAMPS_OF(L)[(size_t)blk * 8 + i], ip_sum[i]);
#endif
    if (!dyn) {                                           // slots (wave, group): factor 1 if idle
        for (uint32_t grp = grp_wave; grp < (uint32_t)ngrp; ++grp) flush_wave(grp);
    }

    if (stamps && lane == 0 && !hook_mine) atomicMax(&stamps[4], wall_clock64());   // the wave that is done with its tiles last
    // ---- deterministic block reduction -> one partial per (point, block) ----
    const uint32_t nres = dyn ? ntile_blk : (uint32_t)nwave;   // result slots per group
    // Every point at once, SIXTEEN lanes (one DPP row) per point: lane j of the row multiplies the slots j, j + 16, ... in index
    // order, then the row's sixteen products meet through four DPP moves -- mirror of the row, mirror of its halves, quad
    // permutes over distance 1 and 2: VALU moves, no trip through the LDS crossbar -- and the row's lane 0 takes the point's
    // only logarithm.  (Until round 4's second session a WAVE reduced a point, and a wave's three points of a 48-point launch
    // one after the other: 64-lane butterflies of ds_bpermutes with a renormalisation per step, 4.9 us between "last wave
    // done with its tiles" and "block reduced" in the launch's timeline -- tools/stamps.py.)  The order of the products is a
    // function of the slot count alone, the same for every wave shape and kernel, so a point's value still does not depend
    // on how it was evaluated.  Sixteen factors >= 2^-64 cannot underflow: one renormalisation after the butterfly.
    __syncthreads();
    {
        const uint32_t j16 = (uint32_t)lane & 15u;
        for (int b = tid >> 4; b < NVS * NPT; b += nthread >> 4) {          // (uniform over a row of 16 lanes)
            const int vsb = SPLIT ? b / NPT : 0, bl = b - vsb * NPT;
            const int grp = bl / NP, bb = bl - grp * NP;
            ScaledProd p{1.0, 0.0};
            const uint32_t nres_b = SPLIT ? nt_of((uint32_t)vsb) : nres;
            for (uint32_t i = j16; i < nres_b; i += 16) {
                const size_t o = ((((size_t)grp * NVS + vsb) * nres + i) * NP + bb) * 2;
                p.m *= tile_llk[o];
                p.e += tile_llk[o + 1];
                sp_renorm(p);                              // a slot can be as small as 2^-64
            }
            auto row_move = [&](double x, auto ctrl) -> double {
                constexpr int kCtrl = decltype(ctrl)::value;
                return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), kCtrl, 0xf, 0xf, false),
                                        __builtin_amdgcn_update_dpp(0, __double2loint(x), kCtrl, 0xf, 0xf, false));
            };
            auto row_step = [&](auto ctrl) {
                const double om = row_move(p.m, ctrl), oe = row_move(p.e, ctrl);
                p.m *= om;
                p.e += oe;
            };
            row_step(std::integral_constant<int, 0x140>());     // row_mirror:      lane j <- lane 15 - j
            row_step(std::integral_constant<int, 0x141>());     // row_half_mirror: lane j <- lane 7 - j of its half
            row_step(std::integral_constant<int, 0xB1>());      // quad_perm [1,0,3,2]
            row_step(std::integral_constant<int, 0x4E>());      // quad_perm [2,3,0,1]
            sp_renorm(p);
            // the only logarithm of this (workgroup, point): log(prod lk) = log(m) + e*ln2
            if (j16 == 0) red[b] = log_nonneg(p.m) + p.e * 6.93147180559945286227e-01;
        }
    }
    __syncthreads();
    if (VB2_STAMPS_OF(L) && tid == 0) {
        const unsigned long long t5 = wall_clock64();
        if (stamps) stamps[5] = t5;
        if (blk == 0 && nblk > 32 && !hook.no_collect()) VB2_STAMPS_OF(L)[20 * 8 + 7] += t5 - VB2_STAMPS_OF(L)[7];     // (accumulated: since "has the round")
    }
    }      // (!collect_only)
    // A result that the host waits for (done_flag: llk_out is mapped host memory) is stored THROUGH the caches (a relaxed
    // system-scope store); before the flag goes out the storing lanes wait for their stores' acknowledgements (vmcnt).
    // Round 3 stored plainly and then fenced -- __threadfence_system, an ACQ_REL counter and a RELEASE flag store: three
    // write-backs / invalidations of the whole L2 -- which was 10 of the 20 us an EMPTY cohort step took (round 4 ablation).
    auto put_result = [&](int b, double v) {
        if (done_flag) __hip_atomic_store(&llk_out[b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else llk_out[b] = v;
    };
    if (ticket == nullptr) {                     // two-kernel mode: llk_finalize_kernel follows
        if (tid < NPT) partials[(size_t)tid * nblk + blk] = red[tid];
        return;
    }
    if (tag == 0) {
        // ---- single-launch mode A (large batches): the last workgroup to arrive sums all partials ----
        // Hand-off through 8-byte agent-scope atomics on both sides (write-through stores,
        // L1-bypassing loads), drained before the ticket is drawn: placement independent.
        if (tid < NVS * NPT) {
            // (SPLIT: the sums of this workgroup's points over the virtual blocks vb0 + j * vstep, where a plain launch has them)
            const int vsb = SPLIT ? tid / NPT : 0, bl = tid - vsb * NPT;
            __hip_atomic_store(&partials[(size_t)(p_off + bl) * nblk + (vb0 + (uint32_t)vsb * vstep)], red[tid],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned int* last_flag = queue;                                    // the queue is drained
        if (tid == 0) {
            const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = (t == nblk - 1) ? 1u : 0u;
        }
        __syncthreads();
        if (*last_flag == 0u) return;
        // same summation order as llk_finalize_kernel: lane-strided, then a wave butterfly
        const int nbt = (int)nblk;
        // (a wave's points side by side -- three of a 48-point launch: their loads, each a trip to memory, are in flight
        // together, and so are their butterflies; every point's sum in the order it always had)
        constexpr int kSide = 3;
        for (int b0 = wave; b0 < num_valid; b0 += kSide * nwave) {
            double s[kSide];
#pragma unroll
            for (int u = 0; u < kSide; ++u) s[u] = 0;
            for (int base = 0; base < nbt; base += 8 * 64) {
                double x[kSide][8];
#pragma unroll
                for (int u = 0; u < kSide; ++u) {
                    const int b = b0 + u * nwave;
                    const double* p = partials + (size_t)(b < num_valid ? b : b0) * nbt;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = base + q * 64 + lane;
                        x[u][q] = i < nbt ? __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < kSide; ++u)
#pragma unroll
                    for (int q = 0; q < 8; ++q) s[u] += x[u][q];
            }
#pragma unroll
            for (int u = 0; u < kSide; ++u) {
                const int b = b0 + u * nwave;
                const double t = wave_sum(s[u]);
                if (lane == 0 && b < num_valid) put_result(b, t);
            }
        }
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    // ---- single-launch mode B (small batches, resident search): workgroup 0 collects all partials ----
    // Every workgroup publishes its NPT sums plus a check word (XOR of position-dependent word
    // hashes ^ mix(launch tag)) with write-through stores and is done: no drain, no arrival counter.
    // Workgroup 0 polls the NPT+1 words of each workgroup with L1-bypassing loads until the
    // check word fits -- stale words of an earlier launch or a half-arrived set fail it -- so
    // the hand-off costs one store propagation plus one read instead of three dependent
    // round trips through the fabric (store drain, ticket atomic, reload).
    unsigned long long* pw = reinterpret_cast<unsigned long long*>(partials);
    if (wave == 0 && !collect_only) {
        const unsigned long long bits = lane < NPT ? (unsigned long long)__double_as_longlong(red[lane]) : 0ull;
        // (the sums first: they do not wait for the check word; then the check word from lane 0, which has the XOR of the
        // NPT word hashes after log2(NPT) exchange steps -- a search round's four words: two quad permutes -- instead of a
        // 64-lane butterfly of six: the check word is what workgroup 0 waits for)
        if (lane < NPT)
            __hip_atomic_store(&pw[(size_t)lane * nblk + blk], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long x = lane < NPT ? word_hash(bits, (unsigned)lane) : 0ull;
        int top = 1;
        while (top < NPT) top <<= 1;
        for (int off = top >> 1; off >= 1; off >>= 1) x ^= __shfl_xor(x, off, 64);
        if (lane == 0)
            __hip_atomic_store(&pw[(size_t)NPT * nblk + blk], x ^ resident_mix(tag), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if (collect_only ? false : (blk != 0 || hook.no_collect())) return;
    if (!hook.sum_stage()) __syncthreads();            // the table is dead: its space stages the partials
    const int nb = (int)nblk;
    double* stage = hook.sum_stage() ? hook.sum_stage() : lds;      // [NPT][nb]
    {
        const unsigned long long want = resident_mix(tag);
        const unsigned long long t_wait = wall_clock64();
        for (int b = tid; b < nb; b += nthread) {      // one thread per workgroup's set
            // (Not waiting for this workgroup's own set -- it is in its LDS already -- was measured twice and dropped: the set
            // that arrives last is not this workgroup's own.  HISTORY.md, rounds 3 and 4.)
            for (unsigned it = 0;; ++it) {
                // check word and the first words are requested together, NP + 1 loads in flight
                unsigned long long x = __hip_atomic_load(&pw[(size_t)NPT * nb + b], __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                for (int w0 = 0; w0 < NPT; w0 += NP) {       // NPT is a multiple of NP (4 or 8)
                    unsigned long long v[NP];
#pragma unroll
                    for (int u = 0; u < NP; ++u)
                        v[u] = __hip_atomic_load(&pw[(size_t)(w0 + u) * nb + b], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        x ^= word_hash(v[u], (unsigned)(w0 + u));
                        stage[(size_t)(w0 + u) * nb + b] = __longlong_as_double((long long)v[u]);
                    }
                }
                if (x == want) break;
                // a quarter of a second (100 MHz ticks): a workgroup is missing -- part of the grid is not on the CUs
                // (a shared GPU).  The caller sees NaN and redoes the step with the arrival-ticket hand-off, which
                // waits for nobody.  (Round 2 waited two seconds: the hiccup of a co-residency failure is this wait.)
                if ((it & 63) == 63 && wall_clock64() - t_wait > kHandoffGiveUpTicks) {
                    for (int w = 0; w < NPT; ++w) stage[(size_t)w * nb + b] = __builtin_nan("");
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    // same summation order as llk_finalize_kernel: lane-strided, then a wave butterfly
    for (int b = wave; b < num_valid; b += nwave) {
        const double* p = stage + (size_t)b * nb;
        double s = 0;
        for (int base = 0; base < nb; base += 8 * 64) {
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = base + q * 64 + lane;
                x[q] = i < nb ? p[i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += x[q];
        }
        s = wave_sum(s);
        if (lane == 0) put_result(b, s);                 // NaN if a workgroup never reported
    }
    }
    if (VB2_STAMPS_OF(L) && tid == 0) {
        const unsigned long long t6 = wall_clock64();
        if (stamps) stamps[6] = t6;
        if ((collect_only || (blk == 0 && !hook.no_collect())) && nblk > 32) VB2_STAMPS_OF(L)[21 * 8 + 7] += t6 - VB2_STAMPS_OF(L)[7];
    }
    if (done_flag && !(kAblate & kAblNoSignal)) {
        // host hand-off without a stream synchronisation: results (in mapped host memory)
        // first, then the sequence number the host is spinning on.  In a multi-sample launch
        // the sample that completes last (batch_done counter) is the one that signals.
        // (every sample's results are acknowledged before its count goes up; the counter's own order then puts the
        // last sample's flag behind all of them.  Relaxed operations: nothing else of this launch is read by the host.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            bool signal = true;
            if (batch_done) {
                const unsigned int t = __hip_atomic_fetch_add(batch_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                signal = (t == batch_active - 1);
                if (signal) __hip_atomic_store(batch_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (a system-scope RELEASE store, in this one workgroup: VERDICT r5 #6 -- no measurable cost against the relaxed store,
            // profiles/r06/ab_flag_release.txt; the host's NaN prefill of the result words stays as the second line)
            if (signal)
                __hip_atomic_store(done_flag, done_seq, VB2_FLAG_RELEASE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int MODE, int QUEUE, int KSEL = 0, bool PD = false>
__global__ void __launch_bounds__(Geom<MODE>::kMaxWaves * 64, Geom<MODE>::kWavesPerSimd)
llk_eval_kernel(const DeviceLayout L, const InlinePoints ip, const double* __restrict__ points,
                int num_valid, double* __restrict__ partials, double* __restrict__ llk_out,
                unsigned int* __restrict__ ticket, unsigned long long* __restrict__ done_flag,
                unsigned long long done_seq, int ngrp, unsigned long long tag, const Schedule sch)
{
    eval_body<MODE, false, NoHook, false, QUEUE, KSEL, false, 8, PD>(L, ip.v, ip.count, points, num_valid, partials, llk_out, ticket,
                                                                     done_flag, done_seq, blockIdx.x, gridDim.x, nullptr, 0u, ngrp, tag, sch);
}

// see eval_body, SPLIT
template <int KSEL, int S>
__global__ void __launch_bounds__(Geom<2>::kMaxWaves * 64, Geom<2>::kWavesPerSimd)
llk_eval_split_kernel(const DeviceLayout L, const double* __restrict__ points, int num_valid, double* __restrict__ partials,
                      double* __restrict__ llk_out, unsigned int* __restrict__ ticket, unsigned long long* __restrict__ done_flag,
                      unsigned long long done_seq, int ngrp)
{
    eval_body<2, false, NoHook, false, 1, KSEL, false, 8, true, S>(L, nullptr, 0, points, num_valid, partials, llk_out, ticket, done_flag,
                                                                     done_seq, blockIdx.x, gridDim.x, nullptr, 0u, ngrp, 0ull,
                                                                     Schedule{nullptr, nullptr});
}

// A call of more points than the LDS holds tables for -- wide quality alphabets: 118 codes x 8 points are 48.5 KB per point
// group, two groups per workgroup -- as ONE launch of several passes (round 4; before: one launch per 16 points, each with
// its own start-up, table build, tail and hand-off, ~10 us of a 41.6 us launch).  A pass is eval_body on its own points,
// its own stretch of the partial sums and its own arrival ticket; a workgroup that has delivered a pass's sums starts on the
// next pass at once -- only the workgroup that arrives last at a pass adds that pass up -- so the hand-offs of all passes but
// the last hide behind the other workgroups' work.  Same grid, same groups, same order as the separate launches: the same
// bits.  The results of the earlier passes are stored through the caches like the last one's (eval_body does that when it
// is given a flag to raise: theirs is a scratch word on the device), and acknowledged before the storing workgroup draws
// its next ticket, so the flag the host waits for is behind every pass's results.
template <int KSEL, bool PD = false>
__global__ void __launch_bounds__(Geom<2>::kMaxWaves * 64, Geom<2>::kWavesPerSimd)
llk_eval_passes_kernel(const DeviceLayout L, const double* __restrict__ points, int num_valid, int points_per_pass,
                       double* __restrict__ partials, double* __restrict__ llk_out, unsigned int* __restrict__ tickets,
                       unsigned long long* __restrict__ done_flag, unsigned long long done_seq,
                       unsigned long long* __restrict__ scratch_flag)
{
    const int stride = 2 * L.num_pc + 1;
    int pass = 0;
    for (int first = 0; first < num_valid; first += points_per_pass, ++pass) {
        const int left = num_valid - first;
        const int nv = left < points_per_pass ? left : points_per_pass;
        const bool last = left <= points_per_pass;
        if (pass > 0) __syncthreads();                      // the pass before is done with the workgroup's LDS
        eval_body<2, false, NoHook, false, 1, KSEL, false, (PD ? 8 : 6), PD>(
            L, nullptr, 0, points + (size_t)first * stride, nv, partials + (size_t)first * gridDim.x, llk_out + first,
            tickets + pass, (last || !done_flag) ? done_flag : scratch_flag, done_seq, blockIdx.x, gridDim.x, nullptr, 0u,
            (nv + 7) / 8, 0ull, Schedule{nullptr, nullptr});
    }
}

// Two translation units from this file (csrc/Makefile, CMakeLists.txt): the single-sample kernels and the resident search
// kernel are compiled under LLVM's iterative-ILP scheduler (+1.1 % on the 48-point launch, -1.2 % on OptimizeLLK), which
// costs llk_eval_passes_kernel 7 % (118 codes: 566 -> 525 k evals/s) and the cohort steps of two and more points 1 % (a
// cohort search 3 %: 700 -> 678 samples/s) -- so those are instantiated in llk_passes.hip's unit (VB2_TU_PASSES: the
// device code, llk_eval_passes_kernel's instantiations, llk_eval_multi_kernel with its launcher; default scheduler) and
// the main unit (VB2_TU_MAIN) only declares them.  Neither macro: everything in one unit (the profiling builds,
// tools/build_variant.sh).
#define VB2_PASSES_KERNEL_ARGS                                                                                           \
    const DeviceLayout, const double*, int, int, double*, double*, unsigned int*, unsigned long long*, unsigned long long, \
        unsigned long long*
#if defined(VB2_TU_MAIN)
extern template __global__ void llk_eval_passes_kernel<4>(VB2_PASSES_KERNEL_ARGS);
extern template __global__ void llk_eval_passes_kernel<2>(VB2_PASSES_KERNEL_ARGS);
extern template __global__ void llk_eval_passes_kernel<0>(VB2_PASSES_KERNEL_ARGS);
extern template __global__ void llk_eval_passes_kernel<4, true>(VB2_PASSES_KERNEL_ARGS);
extern template __global__ void llk_eval_passes_kernel<2, true>(VB2_PASSES_KERNEL_ARGS);
extern template __global__ void llk_eval_passes_kernel<0, true>(VB2_PASSES_KERNEL_ARGS);
#elif defined(VB2_TU_PASSES)
template __global__ void llk_eval_passes_kernel<4>(VB2_PASSES_KERNEL_ARGS);
template __global__ void llk_eval_passes_kernel<2>(VB2_PASSES_KERNEL_ARGS);
template __global__ void llk_eval_passes_kernel<0>(VB2_PASSES_KERNEL_ARGS);
template __global__ void llk_eval_passes_kernel<4, true>(VB2_PASSES_KERNEL_ARGS);
template __global__ void llk_eval_passes_kernel<2, true>(VB2_PASSES_KERNEL_ARGS);
template __global__ void llk_eval_passes_kernel<0, true>(VB2_PASSES_KERNEL_ARGS);
#endif

// Multi-sample launch (BASELINE configs[4]: a cohort in lock-step): workgroup w serves sample
// w / bps as that sample's workgroup w % bps.  Every sample has its own layout, parameter rows,
// partials, ticket and output slot; samples with num_valid == 0 sit this step out.
// STATIC: every sample of the launch is known (on the host) to run the static deal: the item loop is compiled for it
// alone, and pipelined across items (eval_body: PIPE).
template <int MODE, bool W16, int KSEL = 0, int STATIC = 0, bool PD = false>     // KSEL 2 / 4: every sample has that --NumPC and no known-AF column
__global__ void __launch_bounds__(Geom<MODE>::kMaxWaves * 64, Geom<MODE>::kWavesPerSimd)
llk_eval_multi_kernel(const DeviceLayout* __restrict__ layouts, const Schedule* __restrict__ scheds,
                      const double* __restrict__ points,
                      const int* __restrict__ num_valid, double* __restrict__ partials,
                      double* __restrict__ llk_out, unsigned int* __restrict__ tickets, int bps,
                      unsigned long long* __restrict__ done_flag, unsigned long long done_seq,
                      unsigned int* __restrict__ batch_done, unsigned int batch_active, int use_ticket,
                      const MultiInline mi)
{
    constexpr int NP = ModeNp<MODE>::value;
    const int s = blockIdx.x / bps;
    // (mi: the step's point counts and parameter rows as kernel arguments when they fit -- a step of 32 samples x 1 point
    // or 16 x 2 --: otherwise every workgroup reads them from mapped host memory, two dependent trips over PCIe, ~2.4 us
    // of the ~20 an empty step took in round 3)
    const int nv = (kAblate & kAblNoMap) ? NP : mi.count > 0 ? (int)mi.nv[s] : num_valid[s];
    if (nv <= 0) return;                                   // uniform for the workgroup
    const DeviceLayout L = layouts[s];
    const int stride = 2 * L.num_pc + 1;
    eval_body<MODE, W16, NoHook, true, (STATIC ? 0 : -1), KSEL, false, 8, PD>(L, mi.v + (size_t)s * NP * stride, mi.count, points + (size_t)s * NP * stride, nv,
                          partials + (size_t)s * (NP + 1) * bps, llk_out + (size_t)s * NP, tickets + s,
                          done_flag, done_seq, (uint32_t)(blockIdx.x % bps), (uint32_t)bps,
                          batch_done, batch_active, 1, use_ticket ? 0ull : done_seq,
                          scheds ? scheds[s] : Schedule{nullptr, nullptr});
}

#ifndef VB2_TU_PASSES      // ---- main unit only, down to raise_lds_limit ----

#include "resident_kernel.inc"

// Sums the per-block partials in a fixed order -> bitwise reproducible: wave w takes
// points w, w+4, ...; lane l adds blocks l, l+64, ... (8 independent loads in flight),
// then a butterfly over the wave.
__global__ void __launch_bounds__(256)
llk_finalize_kernel(const double* __restrict__ partials, int nb, int num_point,
                    double* __restrict__ llk_out, unsigned long long* __restrict__ done_flag,
                    unsigned long long done_seq)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < num_point; b += 4) {
        const double* p = partials + (size_t)b * nb;
        double s = 0;
        for (int base = 0; base < nb; base += 8 * 64) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 64 + lane;
                x[u] = i < nb ? p[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += x[u];
        }
        s = wave_sum(s);
        if (lane == 0) {
            if (done_flag) __hip_atomic_store(&llk_out[b], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else llk_out[b] = s;
        }
    }
    if (done_flag) {                    // (stores through the caches, acknowledged, then the flag: see eval_body)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done_flag, done_seq, VB2_FLAG_RELEASE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
bool eval_takes_the_queue(const DeviceLayout& L, int nblk, int nwave, int ngrp)
{
    return eval_is_dynamic(L, (uint32_t)nblk, nwave, ngrp);
}

static size_t split_shmem(const DeviceLayout& L, int grid, int block_waves, int ngrp_total, int ways);

LaunchGeom launch_geom(const DeviceLayout& L, int btl, int ngrp)
{
    const int max_waves = Geom<2>::kMaxWaves;
    const int grid_target = Geom<2>::kBlocksPerCU * L.num_cu;
    (void)btl;
    int bw = (L.num_mt + grid_target - 1) / grid_target;
    bw = bw < 4 ? 4 : (bw > max_waves ? max_waves : bw);
    int grid = (L.num_mt + bw - 1) / bw;
    grid = grid < 1 ? 1 : (grid > grid_target ? grid_target : grid);
    if (L.pd && grid > (L.num_mt >> 1)) grid = L.num_mt >> 1 > 0 ? L.num_mt >> 1 : 1;      // (a workgroup owns pairs of tiles)
    // A dictionary whose tables of a full launch (kMaxGroups groups) do not fit PAIRS of workgroups is split three ways
    // (eval_body, SPLIT): the grid -- the same for every launch kind of the context, so that a point's sum is the same bits in
    // all of them -- is then a multiple of 6 (256 CUs: 252 workgroups)
    if (L.pd && grid >= 12 && split_shmem(L, grid & ~1, max_waves, kMaxGroups, 2) > (size_t)kLdsLimitBytes)
        grid -= grid % 6;
    // Several point groups: a workgroup's work items are (tile, group) pairs, so a small sample -- a marker shard
    // of an 8-GPU run: 12 500 markers = 3-4 tiles per workgroup -- still has work for 16 waves, and 1 024 threads
    // to build its six tables with (4-wave workgroups took 47 us for a 48-point launch on 12 500 markers, against
    // 74 us on 100 000).  The grid, hence the tiles a workgroup owns, stays what it was.
    if (ngrp > 1) {
        const int items = (int)owned_most(L.pd ? 1 : 0, (uint32_t)L.num_mt, (uint32_t)grid) * ngrp;
        if (items > bw) bw = items > max_waves ? max_waves : items;
    }
    return LaunchGeom{grid, bw};
}

#endif  // !VB2_TU_PASSES

// More than 64 KiB of dynamic LDS is an opt-in per kernel function AND per device (the function
// object is per device in the runtime), so the flag is kept per (function slot, device).
// (keyed by the function's address and the device -- ADVICE r3: the hand-numbered slots of round 3 were one
// specialisation away from a collision --: an open-addressed table of (function, device) keys, lock-free, a few dozen
// entries ever)
static hipError_t raise_lds_limit(const void* fn)
{
    constexpr unsigned kCap = 1024;
    static std::atomic<uintptr_t> keys[kCap];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uintptr_t key = (reinterpret_cast<uintptr_t>(fn) << 8) ^ (uintptr_t)(dev + 1);     // (never 0)
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) % kCap;
    for (unsigned probe = 0; probe < kCap; ++probe, h = (h + 1) % kCap) {
        const uintptr_t seen = keys[h].load(std::memory_order_acquire);
        if (seen == key) return hipSuccess;
        if (seen == 0) break;
    }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e != hipSuccess) return e;
    for (unsigned probe = 0; probe < kCap; ++probe, h = (h + 1) % kCap) {
        uintptr_t expect = 0;
        if (keys[h].compare_exchange_strong(expect, key, std::memory_order_acq_rel) || expect == key) break;
    }
    return hipSuccess;
}

#ifndef VB2_TU_PASSES      // ---- main unit only, down to the cohort kernels' launcher ----
// the single-sample kernel of a wave shape: one per way of dealing (queue / static); --NumPC 2 and 4 without a known-AF column
// have kernels of their own (the static deal of the search shapes only in the general form: a search takes the queue)
template <int MODE, bool PD>
static const void* eval_kernel_fn(int ksel, bool dyn)
{
    constexpr int kStaticQ = MODE == 2 ? 0 : 1;     // (never selected for MODE != 2: ksel is 0 there)
    return ksel == 4 ? (dyn ? reinterpret_cast<const void*>(&llk_eval_kernel<MODE, 1, 4, PD>)
                            : reinterpret_cast<const void*>(&llk_eval_kernel<MODE, kStaticQ, 4, PD>))
           : ksel == 2 ? (dyn ? reinterpret_cast<const void*>(&llk_eval_kernel<MODE, 1, 2, PD>)
                              : reinterpret_cast<const void*>(&llk_eval_kernel<MODE, kStaticQ, 2, PD>))
           : dyn  ? reinterpret_cast<const void*>(&llk_eval_kernel<MODE, 1, 0, PD>)
                  : reinterpret_cast<const void*>(&llk_eval_kernel<MODE, 0, 0, PD>);
}

template <int MODE>
static hipError_t launch_btl(const DeviceLayout& L, const double* d_points, const double* h_points,
                             int num_valid, int ngrp,
                             double* d_partials, double* d_out, unsigned int* d_ticket,
                             unsigned long long* done_flag, unsigned long long done_seq,
                             unsigned long long tag, hipStream_t stream, ScheduleProvider* sp)
{
    constexpr int NP = ModeNp<MODE>::value;     // points per group
    const LaunchGeom gm = launch_geom(L, MODE == 2 ? 2 : 1, ngrp);
    const Schedule sch = sp ? sp->get(MODE, ngrp, gm.grid, gm.block_waves) : Schedule{nullptr, nullptr};
    const size_t shmem = eval_shmem_np(L, NP, gm.grid, gm.block_waves, ngrp);
    // (the group cap of launch_llk_eval is worked out on the geometry of a kMaxGroups launch; this launch's own
    // geometry -- fewer groups, maybe fewer waves -- needs no more LDS than that today, but nothing else says so)
    if (shmem > (size_t)kLdsLimitBytes) return hipErrorInvalidConfiguration;
    // one kernel per way of dealing (queue / static); --NumPC 2 and 4 without a known-AF column have kernels of their own
    // (the static deal of the search shapes only in the general form: a search takes the queue)
    const bool dyn = eval_is_dynamic(L, (uint32_t)gm.grid, gm.block_waves, ngrp);
    const int ksel = (L.known_af == nullptr && (MODE == 2 || dyn)) ? (L.num_pc == 4 ? 4 : L.num_pc == 2 ? 2 : 0) : 0;
    const void* fn = L.pd ? eval_kernel_fn<MODE, true>(ksel, dyn) : eval_kernel_fn<MODE, false>(ksel, dyn);
    {
        hipError_t e = raise_lds_limit(fn);
        if (e != hipSuccess) return e;
    }
    InlinePoints ip;
    ip.count = 0;
    const int ndbl = num_valid * (2 * L.num_pc + 1);
    if (h_points && ndbl <= kInlinePointDoubles) {      // small batch with host-visible values
        ip.count = ndbl;
        for (int i = 0; i < ndbl; ++i) ip.v[i] = h_points[i];
    }
    {
        DeviceLayout Lc = L;
        const double* a_points = d_points;
        int a_nv = num_valid, a_ngrp = ngrp;
        unsigned long long a_seq = done_seq, a_tag = tag;
        Schedule a_sch = sch;
        void* args[] = {&Lc, &ip, &a_points, &a_nv, &d_partials, &d_out, &d_ticket, &done_flag, &a_seq, &a_ngrp, &a_tag, &a_sch};
        return hipLaunchKernel(fn, dim3(gm.grid), dim3(gm.block_waves * 64), args, shmem, stream);
    }
}

// see llk_eval_passes_kernel.  Needs the work queue (a slot per item: the launches it replaces take it too at these sizes).
static hipError_t launch_passes(const DeviceLayout& L, const double* d_points, int num_valid, int groups_per_launch,
                                double* d_partials, double* d_out, unsigned int* d_tickets,
                                unsigned long long* done_flag, unsigned long long done_seq, hipStream_t stream, bool* taken)
{
    *taken = false;
    if (L.known_af != nullptr) return hipSuccess;
    // point groups per pass: what fits beside the compact exp table (4 KiB instead of 16: see exp_nonpos) -- 118 codes: three
    // groups instead of two, i.e. 75 work items for a workgroup's 16 waves instead of 50 (5 rounds at 94 % instead of 4 at
    // 78 %) and two passes per 48 points instead of three
    constexpr int kCompactExpTabDoubles = 64 * 8;
    int g = kMaxGroups;
    LaunchGeom gm = launch_geom(L, 2, g);
    while (g > 1 && eval_shmem_np(L, 8, gm.grid, gm.block_waves, g, kCompactExpTabDoubles) > (size_t)kLdsLimitBytes) {
        --g;
        gm = launch_geom(L, 2, g);
    }
    const int ngroup = (num_valid + 7) / 8;
    const int npass = (ngroup + g - 1) / g;
    // only where the passes are fewer than the launches they replace (118 codes: 2 for 3, 125 -> 120 us per 48 points; 72
    // codes: 2 for 2 -- measured equal, 92.3 / 93.4 us, and the launches keep the conflict-free exp table)
    if (npass >= (ngroup + groups_per_launch - 1) / groups_per_launch) return hipSuccess;
    const int gpp = (ngroup + npass - 1) / npass;            // balanced: 6 groups at 4 per pass -> 3 + 3, not 4 + 2
    if (npass > kTicketScratchWord) return hipSuccess;
    gm = launch_geom(L, 2, gpp);
    if (!eval_is_dynamic(L, (uint32_t)gm.grid, gm.block_waves, gpp)) return hipSuccess;
    const size_t shmem = eval_shmem_np(L, 8, gm.grid, gm.block_waves, gpp, kCompactExpTabDoubles);
    if (shmem > (size_t)kLdsLimitBytes) return hipSuccess;
    const void* fn = L.pd ? (L.num_pc == 4 ? reinterpret_cast<const void*>(&llk_eval_passes_kernel<4, true>)
                             : L.num_pc == 2 ? reinterpret_cast<const void*>(&llk_eval_passes_kernel<2, true>)
                                             : reinterpret_cast<const void*>(&llk_eval_passes_kernel<0, true>))
                     : L.num_pc == 4 ? reinterpret_cast<const void*>(&llk_eval_passes_kernel<4>)
                     : L.num_pc == 2 ? reinterpret_cast<const void*>(&llk_eval_passes_kernel<2>)
                                     : reinterpret_cast<const void*>(&llk_eval_passes_kernel<0>);
    hipError_t e = raise_lds_limit(fn);
    if (e != hipSuccess) return e;
    DeviceLayout Lc = L;
    const double* a_points = d_points;
    int a_nv = num_valid, a_ppp = 8 * gpp;
    unsigned long long a_seq = done_seq;
    unsigned long long* a_scratch = reinterpret_cast<unsigned long long*>(d_tickets + kTicketScratchWord);
    void* args[] = {&Lc, &a_points, &a_nv, &a_ppp, &d_partials, &d_out, &d_tickets, &done_flag, &a_seq, &a_scratch};
    *taken = true;
    return hipLaunchKernel(fn, dim3(gm.grid), dim3(gm.block_waves * 64), args, shmem, stream);
}

// LDS of a split launch's workgroup (eval_body, SPLIT = ways): the tables of its share of the groups, the result slots of `ways` virtual blocks
static size_t split_shmem(const DeviceLayout& L, int grid, int block_waves, int ngrp_total, int ways)
{
    // (a workgroup of a plain launch of this grid with 1 / ways of the groups, plus the other virtual blocks' result slots -- as
    // many as the first's -- and their sums)
    const size_t per = (size_t)((ngrp_total + ways - 1) / ways);
    const size_t slots1 = (size_t)owned_most(1, (uint32_t)L.num_mt, (uint32_t)(grid >= 1 ? grid : 1)) * per;
    return eval_shmem_np(L, 8, grid >= 1 ? grid : 1, block_waves, (int)per) + sizeof(double) * ((size_t)(ways - 1) * 2 * slots1 * 8 + (size_t)(ways - 1) * 8 * per);
}
// ways of a split launch of ngrp groups: the fewest workgroups per set whose tables fit (0: none does)
static int split_ways(const DeviceLayout& L, const LaunchGeom& gm, int ngrp)
{
    for (int ways = 2; ways <= 3; ++ways) {
        if (gm.grid < ways || gm.grid % ways || (ways == 3 && tunables().split == 2)) continue;
        const int per = (ngrp + ways - 1) / ways;
        if (per * (ways - 1) >= ngrp) continue;            // (a workgroup of the set would have no group)
        if (split_shmem(L, gm.grid, gm.block_waves, ngrp, ways) <= (size_t)kLdsLimitBytes) return ways;
    }
    return 0;
}

// see eval_body, SPLIT.  *taken = false: the geometry does not allow it (the caller falls back to passes / several launches)
static hipError_t launch_split(const DeviceLayout& L, const double* d_points, int num_valid, double* d_partials, double* d_out,
                               unsigned int* d_ticket, unsigned long long* done_flag, unsigned long long done_seq,
                               hipStream_t stream, bool* taken)
{
    *taken = false;
    if (!L.pd) return hipSuccess;
    const int ngrp = (num_valid + 7) / 8;
    if (ngrp < 2) return hipSuccess;
    const LaunchGeom gm = launch_geom(L, 2, ngrp);      // (a set of workgroups has the items of one workgroup of a plain launch, `ways` times)
    const int ways = split_ways(L, gm, ngrp);
    if (ways == 0) return hipSuccess;
    // every item through the queue, a slot apiece
    const uint32_t tiles_set = (uint32_t)ways * owned_most(1, (uint32_t)L.num_mt, (uint32_t)gm.grid);
    if (tiles_set * (uint32_t)((ngrp + ways - 1) / ways) > (uint32_t)(L.dyn_limit * gm.block_waves)) return hipSuccess;
    const size_t shmem = split_shmem(L, gm.grid, gm.block_waves, ngrp, ways);
    const int ksel = L.known_af == nullptr ? (L.num_pc == 4 ? 4 : L.num_pc == 2 ? 2 : 0) : 0;
    const void* fn = ways == 2 ? (ksel == 4 ? reinterpret_cast<const void*>(&llk_eval_split_kernel<4, 2>)
                                  : ksel == 2 ? reinterpret_cast<const void*>(&llk_eval_split_kernel<2, 2>)
                                              : reinterpret_cast<const void*>(&llk_eval_split_kernel<0, 2>))
                               : (ksel == 4 ? reinterpret_cast<const void*>(&llk_eval_split_kernel<4, 3>)
                                  : ksel == 2 ? reinterpret_cast<const void*>(&llk_eval_split_kernel<2, 3>)
                                              : reinterpret_cast<const void*>(&llk_eval_split_kernel<0, 3>));
    hipError_t e = raise_lds_limit(fn);
    if (e != hipSuccess) return e;
    DeviceLayout Lc = L;
    const double* a_points = d_points;
    int a_nv = num_valid, a_ngrp = ngrp;
    unsigned long long a_seq = done_seq;
    void* args[] = {&Lc, &a_points, &a_nv, &d_partials, &d_out, &d_ticket, &done_flag, &a_seq, &a_ngrp};
    *taken = true;
    return hipLaunchKernel(fn, dim3(gm.grid), dim3(gm.block_waves * 64), args, shmem, stream);
}

hipError_t launch_llk_eval(const DeviceLayout& L, int num_point, const double* d_points,
                           const double* h_points, double* d_partials, double* d_out,
                           unsigned int* d_ticket,
                           unsigned long long* done_flag, unsigned long long done_seq,
                           unsigned long long* tag_counter, hipStream_t stream, int reduce_override,
                           ScheduleProvider* sched)
{
    const Tunables& tn = tunables();
    unsigned int* tk = tn.single_launch ? d_ticket : nullptr;
    const int reduce_mode = reduce_override ? reduce_override : tn.reduce;
    const int stride = 2 * L.num_pc + 1;
    int done = 0;
    while (done < num_point) {
        const int left = num_point - done;
        const double* p = d_points + (size_t)done * stride;
        const double* hp = h_points ? h_points + (size_t)done * stride : nullptr;
        // up to max_groups x 8 points per launch (more points amortise the fixed costs)
        const LaunchGeom gm2 = launch_geom(L, 2, kMaxGroups);
        // (8-point groups need the wide table rows; a context whose dictionary is too big for
        // 16-bit offsets into wide rows has narrow ones and evaluates 4 points per launch)
        const int cap = L.row_bytes == kRowBytesWide ? 8 * max_groups(L, 2, gm2.grid, gm2.block_waves) : 4;
        // probability-domain contexts: pairs of workgroups split the point groups (Tunables::split 0: passes)
        if (tn.split && L.pd && tk && reduce_mode != 2 && cap >= 8 && left > cap) {
            const int take = left < kMaxPointsPerLaunch ? left : kMaxPointsPerLaunch;
            bool taken = false;
            unsigned long long* dfp = (done + take >= num_point) ? done_flag : nullptr;
            hipError_t es = launch_split(L, p, take, d_partials, d_out + done, tk, dfp, done_seq, stream, &taken);
            if (es != hipSuccess) return es;
            if (taken) {
                done += take;
                continue;
            }
        }
        // more points than one launch's tables hold: the passes of ONE launch (Tunables::passes 0: a launch per `cap` points)
        if (tn.passes && tk && reduce_mode != 2 && L.row_bytes == kRowBytesWide && cap >= 8 && left > cap) {
            const int take = left < kMaxPointsPerLaunch ? left : kMaxPointsPerLaunch;
            bool taken = false;
            unsigned long long* dfp = (done + take >= num_point) ? done_flag : nullptr;
            hipError_t ep = launch_passes(L, p, take, cap / 8, d_partials, d_out + done, tk, dfp, done_seq, stream, &taken);
            if (ep != hipSuccess) return ep;
            if (taken) {
                done += take;
                continue;
            }
        }
        const int step = left < cap ? left : cap;
        const int ngrp = step > 4 ? (step + 7) / 8 : 1;
        unsigned long long* df = (done + step >= num_point) ? done_flag : nullptr;   // last launch signals
        unsigned long long* df_eval = tk ? df : nullptr;
        hipError_t e;
        // hand-off protocol: tagged sets for <= 16 points (one fabric round trip less: latency matters -- 8 points 18.6 ->
        // 16.4 us, 16 points 27.7 -> 26.6 us), arrival ticket for bigger batches (32 points 45.5 against 47.6 us, 48 points
        // 64.3 against 68.5 us: there workgroup 0's polling passes cost more than the ticket's third trip)
        const bool tagged = reduce_mode == 2 || (reduce_mode == 0 && step <= tn.tagged_max);
        const unsigned long long tag = tagged ? ++*tag_counter : 0ull;   // unique per launch on this buffer
        if (step > 4)
            e = launch_btl<2>(L, p, hp, step, ngrp, d_partials, d_out + done, tk, df_eval, done_seq, tag, stream, sched);
        else if (step == 1)
            e = launch_btl<4>(L, p, hp, step, 1, d_partials, d_out + done, tk, df_eval, done_seq, tag, stream, sched);
        else
            e = launch_btl<3>(L, p, hp, step, 1, d_partials, d_out + done, tk, df_eval, done_seq, tag, stream, sched);
        if (e != hipSuccess) return e;
        if (!tk) {
            hipLaunchKernelGGL(llk_finalize_kernel, dim3(1), dim3(256), 0, stream, d_partials,
                               launch_geom(L, step > 4 ? 2 : 1).grid, step, d_out + done, df, done_seq);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        done += step;
    }
    return hipSuccess;
}

size_t eval_shmem_np(const DeviceLayout& L, int np, int nblk, int block_waves, int ngrp, int exp_tab_doubles)
{
    if (exp_tab_doubles <= 0) exp_tab_doubles = kExpTabDoubles;
    const size_t NP = (size_t)np, G = (size_t)ngrp;
    const size_t items = (size_t)owned_most(L.pd ? 1 : 0, (uint32_t)L.num_mt, (uint32_t)nblk) * G;      // (one slot per micro-tile and group)
    const size_t fixed_tabs = L.pd ? 0 : (size_t)exp_tab_doubles + (size_t)kLogTabDoubles;      // (no exp / log table in the probability domain)
    const size_t slots = items <= (size_t)L.dyn_limit * block_waves ? items : (size_t)block_waves * G;
    const size_t bytes = sizeof(double) * (G * (L.num_code + 1) * (size_t)(L.row_bytes / 8) + G * NP + 2 +
                                           G * NP * (2 * L.num_pc + 1) + 1 + G * NP * 2 * L.num_pc +
                                           1 + 2 * (size_t)L.num_prim +
                                           fixed_tabs + 2 * slots * NP);
    // workgroup 0 stages every workgroup's partial sums ([points][workgroups]) over the dead table
    const size_t stage = sizeof(double) * G * NP * (size_t)nblk;
    return bytes > stage ? bytes : stage;
}

int pd_row_budget(int num_marker, int num_pc, int num_cu)
{
    DeviceLayout T;
    std::memset(&T, 0, sizeof(T));
    T.pd = 1;
    T.row_bytes = kRowBytesWide;
    T.num_pc = num_pc;
    T.num_mt = ((num_marker + kMtMarkers - 1) / kMtMarkers + 1) & ~1;
    T.num_cu = num_cu;
    T.dyn_limit = tunables().dyn_tiles;
    // (the tables of a 48-point launch: all six groups' in one workgroup, or -- Tunables::split = ways -- 6 / ways in each of a set)
    for (int rows = kMaxWideCodes; rows > 1; --rows) {
        T.num_code = T.num_prim = rows;
        const LaunchGeom gm = launch_geom(T, 2, kMaxGroups);
        const int ways = split_ways(T, gm, kMaxGroups);
        const size_t need = (tunables().split != 0 && ways >= 2) ? split_shmem(T, gm.grid, gm.block_waves, kMaxGroups, ways)
                                  : eval_shmem_np(T, 8, gm.grid, gm.block_waves, kMaxGroups);
        if (need <= (size_t)kLdsLimitBytes) return rows;
    }
    return 1;
}

size_t eval_shmem_bytes(const DeviceLayout& L, int btl, int nblk, int block_waves, int ngrp)
{
    return eval_shmem_np(L, 4 * btl, nblk, block_waves, ngrp);
}

// Largest number of point groups one launch may carry: LDS (160 KiB per CU) and 4 at most.
int max_groups(const DeviceLayout& L, int btl, int nblk, int block_waves)
{
    int g = kMaxGroups;
    while (g > 1 && eval_shmem_bytes(L, btl, nblk, block_waves, g) > kLdsLimitBytes) --g;
    return g;
}

#endif  // !VB2_TU_PASSES

#ifndef VB2_TU_MAIN         // ---- the cohort kernels are instantiated (here, by their launcher) in the second unit ----
// The cohort kernels of one wave shape
template <int MODE>
static hipError_t launch_multi_mode(const MultiLaunch& ml, hipStream_t stream)
{
    const dim3 grid(ml.num_sample * ml.bps), block(ml.block_waves * 64);
    // The 16-bit run lists (half the bytes of the run words from HBM) in every wave shape.  With round 4's decode -- one
    // byte permute + one 24-bit multiply per run, as many instructions as the 32-bit word's and + add -- and the item
    // loop compiled for the static deal alone (PIPE), they also pay where a step is VALU-bound: 32 C3 samples x 4 points
    // 202 -> 189 us, x 8 points 365 -> 346 us on one box (round 3, with a five-instruction decode and the deal decided
    // in the kernel: 231 -> 307 us, hence "1 and 2 points only" until round 5); 1 point 130.6 -> 117.5, 2 points 139.2 ->
    // 127.9 us when they came in.
    const int use_ticket = ml.force_ticket ? 1 : 0;
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kernel));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, grid, block, ml.shmem, stream, ml.d_layouts, ml.d_scheds, ml.d_points,
                           ml.d_num_valid, ml.d_partials, ml.d_out, ml.d_tickets, ml.bps, ml.done_flag, ml.done_seq,
                           ml.d_batch_done, ml.batch_active, use_ticket, ml.inl);
        return hipGetLastError();
    };
    // (every shape also compiled for --NumPC 2 / 4 without a known-AF column: one-point steps of 32 samples 99 -> 94 us)
    if (ml.pd) {                      // (every sample a probability-domain context)
        if constexpr (MODE == 4 || MODE == 5) {    // (the one- and two-point shapes stream the 8-bit step lists when every sample has them: eval_body, walk_pd8)
            if (ml.w16) {
                if (ml.all_static) {
                    if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, true, 4, 1, true>);
                    if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, true, 2, 1, true>);
                    return go(&llk_eval_multi_kernel<MODE, true, 0, 1, true>);
                }
                if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, true, 4, 0, true>);
                if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, true, 2, 0, true>);
                return go(&llk_eval_multi_kernel<MODE, true, 0, 0, true>);
            }
        }
        if (ml.all_static) {
            if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, false, 4, 1, true>);
            if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, false, 2, 1, true>);
            return go(&llk_eval_multi_kernel<MODE, false, 0, 1, true>);
        }
        if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, false, 4, 0, true>);
        if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, false, 2, 0, true>);
        return go(&llk_eval_multi_kernel<MODE, false, 0, 0, true>);
    }
    if (ml.w16) {
        if (ml.all_static) {          // (the pipelined item loop: compiled for the static deal only)
            if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, true, 4, 1>);
            if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, true, 2, 1>);
            return go(&llk_eval_multi_kernel<MODE, true, 0, 1>);
        }
        if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, true, 4>);
        if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, true, 2>);
        return go(&llk_eval_multi_kernel<MODE, true>);
    }
    if (ml.ksel == 4) return go(&llk_eval_multi_kernel<MODE, false, 4>);
    if (ml.ksel == 2) return go(&llk_eval_multi_kernel<MODE, false, 2>);
    return go(&llk_eval_multi_kernel<MODE, false>);
}

hipError_t launch_llk_eval_multi(const MultiLaunch& ml, hipStream_t stream)
{
    // wave shape by points per sample: 8 -> MODE 2, 4 -> 3, 2 -> 5, 1 -> 4
    switch (ml.np) {
    case 8: return launch_multi_mode<2>(ml, stream);
    case 1: return launch_multi_mode<4>(ml, stream);
    case 2: return launch_multi_mode<5>(ml, stream);
    default: return launch_multi_mode<3>(ml, stream);
    }
}
#endif  // !VB2_TU_MAIN

#ifndef VB2_TU_PASSES      // ---- main unit only, to the end ----
// (the kernels of the flatten -- classify_kernel, pack_layout_kernel, pack_sched_kernel, pack_codes16_kernel -- are in
// flatten_kernels.hip)

bool build_schedule(const uint32_t* rows, int num_mt, int nblk, int nwave, int tpu, int ngrp,
                    std::vector<uint32_t>* off, std::vector<uint16_t>* item, int own_shift)
{
    off->clear();
    item->clear();
    if (num_mt <= 0 || nblk <= 0 || nwave <= 0) return false;
    const uint32_t max_tiles = owned_most(own_shift, (uint32_t)num_mt, (uint32_t)nblk);
    if ((size_t)((max_tiles + tpu - 1) / tpu) * ngrp > 65535) return false;
    off->reserve((size_t)nblk * nwave + 1);
    std::vector<uint64_t> load(nwave);
    std::vector<std::vector<uint16_t>> mine(nwave);
    // cost model (VALU instructions per lane): 28 per row of two runs x two points, ~370 for the
    // per-marker epilogue and the item's fixed work; only the ratio matters
    constexpr uint64_t kRowCost = 28, kFixCost = 370;
    for (int b = 0; b < nblk; ++b) {
        const uint32_t ntile = owned_count(own_shift, (uint32_t)num_mt, (uint32_t)b, (uint32_t)nblk);
        const uint32_t nunit = (ntile + tpu - 1) / tpu;
        std::fill(load.begin(), load.end(), 0);
        for (auto& v : mine) v.clear();
        // the workgroup's tiles b, b + nblk, ... are in descending row order, so walking the units
        // in index order IS longest-first
        for (uint32_t u = 0; u < nunit; ++u) {
            uint32_t r = 0;
            for (int h = 0; h < tpu; ++h) {
                const uint32_t it = (uint32_t)tpu * u + h;
                if (it < ntile) r = std::max(r, rows[owned_tile(own_shift, (uint32_t)b, (uint32_t)nblk, it)]);
            }
            const uint64_t cost = kRowCost * r + kFixCost;
            for (int g = 0; g < ngrp; ++g) {
                int best = 0;
                for (int w = 1; w < nwave; ++w)
                    if (load[w] < load[best]) best = w;
                load[best] += cost;
                mine[best].push_back((uint16_t)((uint32_t)g * nunit + u));
            }
        }
        for (int w = 0; w < nwave; ++w) {
            std::sort(mine[w].begin(), mine[w].end());       // by (group, unit): few group changes per wave
            off->push_back((uint32_t)item->size());
            item->insert(item->end(), mine[w].begin(), mine[w].end());
        }
    }
    off->push_back((uint32_t)item->size());
    return true;
}

// A profiler's tool library in the process (rocprofv3 preloads librocprofiler-sdk-tool.so; rocprof v1/v2 their own)?
bool profiler_attached()
{
    static const bool attached = [] {
        bool found = false;
        dl_iterate_phdr(
            [](struct dl_phdr_info* info, size_t, void* data) -> int {
                const char* n = info->dlpi_name;
                if (n && (std::strstr(n, "rocprofiler-sdk-tool") || std::strstr(n, "librocprofiler64") ||
                          std::strstr(n, "libroctracer") || std::strstr(n, "rocprofv3"))) {
                    *static_cast<bool*>(data) = true;
                    return 1;
                }
                return 0;
            },
            &found);
        return found;
    }();
    return attached;
}

size_t resident_state_doubles(int nmax, int num_pc)
{
    return 8 + DeviceSimplex::lds_doubles(nmax, num_pc) + (size_t)resident_words(num_pc) + 2 +
           (size_t)resident_stage_doubles(nmax, num_pc);
}

uint32_t resident_cache_rows(const uint32_t* rows, int num_mt, int nblk, int own_shift)
{
    uint32_t most = 0;
    for (int b = 0; b < nblk; ++b) {
        const uint32_t ntile = owned_count(own_shift, (uint32_t)num_mt, (uint32_t)b, (uint32_t)nblk);
        uint32_t off = 0, prev = 0;
        for (uint32_t it = 0; it < ntile; ++it) {
            const uint32_t t = owned_tile(own_shift, (uint32_t)b, (uint32_t)nblk, it);
            if (t >= (uint32_t)num_mt) break;
            const uint32_t start = cache_start(own_shift != 0, off, prev, it);
            off = start + rows[t];
            prev = start;
        }
        most = std::max(most, off);
    }
    return most + 4;        // slack: the read loop requests up to kPrefetchL2 rows past a tile's last
}

hipError_t launch_llk_resident(const DeviceLayout& L, ResidentArgs* ra_io, double* d_partials,
                               unsigned int* d_ticket, hipStream_t stream, bool* cooperative)
{
    const LaunchGeom gm = launch_geom(L, 1);
    // dynamic LDS: the evaluation body's, then workgroup 0's search state (16-byte aligned)
    size_t shmem = (eval_shmem_bytes(L, 1, gm.grid, gm.block_waves, 1) + 15) / 16 * 16;
    ResidentArgs& ra = *ra_io;           // state_off / state_nmax and the LDS areas behind the state are filled in here
    ra.state_off = (int32_t)(shmem / sizeof(double));
    shmem += sizeof(double) * resident_state_doubles(ra.state_nmax, L.num_pc);
    if (shmem > (size_t)kLdsLimitBytes && ra.state_nmax > 0) {      // no room: search on the host
        ra.state_nmax = 0;
        shmem = ra.state_off * sizeof(double) + sizeof(double) * resident_state_doubles(0, L.num_pc);
    }
    if (shmem > (size_t)kLdsLimitBytes || gm.grid > L.num_cu) return hipErrorInvalidConfiguration;
    // workgroup 0's own staging area for the partial sums, if there is room (else over the dead tables, as in round 3)
    ra.sum_stage_off = 0;
    {
        const size_t at = (shmem + 15) / 16 * 16, need = sizeof(double) * 4 * (size_t)gm.grid;
        if (at + need <= (size_t)kLdsLimitBytes) {
            ra.sum_stage_off = (int32_t)(at / sizeof(double));
            shmem = at + need;
        }
    }
    const bool dyn = eval_is_dynamic(L, (uint32_t)gm.grid, gm.block_waves, 1);
    const int ksel = (dyn && L.known_af == nullptr) ? (L.num_pc == 4 ? 4 : L.num_pc == 2 ? 2 : 0) : 0;
    // the run-list cache: the paired shape on the work queue (what a search on one device runs), at most 8 tiles per wave
    bool lcache = false;
    ra.cache_off = 0;
    if (ra.cache_rows > 0 && dyn && ra.cache_tiles <= 8 * gm.block_waves) {
        ra.cache_tiles = (ra.cache_tiles + 1) & ~1;
        const size_t at = (shmem + 15) / 16 * 16;
        const size_t need = (size_t)ra.cache_tiles * 8 + (size_t)ra.cache_rows * kMtMarkers * (L.pd ? 4 : 8);
        if (at + need <= (size_t)kLdsLimitBytes) {
            ra.cache_off = (int32_t)(at / sizeof(double));
            shmem = at + need;
            lcache = true;
        }
    }
    if (!lcache) ra.cache_rows = 0;
    auto pick = [&](auto pd_tag) -> const void* {
        constexpr bool kPd = decltype(pd_tag)::value;
        return lcache ? (ksel == 4 ? reinterpret_cast<const void*>(&llk_resident_kernel<1, 4, true, kPd>)
                         : ksel == 2 ? reinterpret_cast<const void*>(&llk_resident_kernel<1, 2, true, kPd>)
                                     : reinterpret_cast<const void*>(&llk_resident_kernel<1, 0, true, kPd>))
               : ksel == 4 ? reinterpret_cast<const void*>(&llk_resident_kernel<1, 4, false, kPd>)
               : ksel == 2 ? reinterpret_cast<const void*>(&llk_resident_kernel<1, 2, false, kPd>)
               : dyn       ? reinterpret_cast<const void*>(&llk_resident_kernel<1, 0, false, kPd>)
                           : reinterpret_cast<const void*>(&llk_resident_kernel<0, 0, false, kPd>);
    };
    const void* fn = L.pd ? pick(std::true_type()) : pick(std::false_type());
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e != hipSuccess) return e;
    // Every workgroup must be on a CU at the same time (they all wait for the control wave and for each other's sums).
    // The grid is at most one 1024-thread workgroup per CU; a cooperative launch makes the runtime GUARANTEE that it is
    // resident as a whole (hipLaunchCooperativeKernel), where a plain one merely finds it so on a device that runs
    // nothing else.  Cooperative is the default.  Plain launches remain for: a process under a profiler (ROCm 7.2's
    // rocprofv3 crashes in its exit handler after a cooperative launch: the caller checks profiler_attached()), a
    // runtime that refuses the cooperative launch, and Tunables::coop = 0 -- the bounded waits on both sides then turn
    // a partly resident grid into a retry with plain per-step launches, not a hang.
    // one workgroup more for the control wave where the grid leaves a CU free (resident_kernel.inc: extra_ctl)
    ra.extra_ctl = (tunables().ctl_block != 0 && ra.state_nmax > 0 && dyn && gm.grid + 1 <= L.num_cu && gm.grid > 4) ? 1 : 0;
    const unsigned launch_grid = (unsigned)gm.grid + (ra.extra_ctl ? 1u : 0u);
    DeviceLayout Lc = L;
    ResidentArgs rc = ra;
    void* args[] = {&Lc, &rc, &d_partials, &d_ticket};
    if (cooperative && *cooperative) {
        e = hipLaunchCooperativeKernel(fn, dim3(launch_grid), dim3(gm.block_waves * 64), args, (unsigned int)shmem, stream);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        *cooperative = false;
    }
    return hipLaunchKernel(fn, dim3(launch_grid), dim3(gm.block_waves * 64), args, shmem, stream);
}

// After a collective on the same stream: the reduced values go to mapped host memory, then the sequence
// number the host spins on -- the marker-shard step needs no device-to-host copy call and no stream
// synchronisation (shard.cpp).  One wave; every lane fences its own stores before the flag is written.
__global__ void __launch_bounds__(64)
publish_kernel(const double* __restrict__ src, double* __restrict__ dst_mapped, int n,
               unsigned long long* __restrict__ done_flag, unsigned long long done_seq)
{
    // (the collective's output is read past the caches -- another kernel, maybe another agent, wrote it --, the copies
    // go through them to host memory, are acknowledged, and then the flag goes out: no cache-wide fence)
    for (int j = threadIdx.x; j < n; j += 64)
        __hip_atomic_store(&dst_mapped[j], __hip_atomic_load(&src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(done_flag, done_seq, VB2_FLAG_RELEASE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_publish(const double* d_src, double* d_dst_mapped, int n, unsigned long long* done_flag,
                          unsigned long long done_seq, hipStream_t stream)
{
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, stream, d_src, d_dst_mapped, n, done_flag, done_seq);
    return hipGetLastError();
}

// Zero-marker case: LLK of an empty sum.
__global__ void fill_zero_kernel(double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.0;
}

hipError_t launch_fill_zero(double* d_out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_zero_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_out, n);
    return hipGetLastError();
}
#endif  // !VB2_TU_PASSES

}  // namespace vb2

// flatten_kernels.hip -- the one-time flatten of a sample on the device (gfx950).
//
// BASELINE north star: "SimplePileupViewer runs once on host to flatten pileup bases/quals into pinned SoA arrays that are
// hipMemcpyAsync'd to HBM, then ... kernels".  What the viewer holds -- bases, qualities, read offsets, alt alleles -- goes up
// as it is; everything that does not depend on (alpha, PC) is made here, byte for byte what the host flatten of context.cpp
// produces (tunable host_flatten; the digest tests compare the two):
//
//   classify_kernel     one thread per marker.  classifyBase + quality clamp (ContaminationEstimator.h:180-184, 296-298) through
//                       byte tables, a marker's reads counted per (class, quality) code in an LDS column of the thread, the codes
//                       that occur in three 64-bit sets; then, in dictionary order, the run words (code | count << 8, counts above 31
//                       split), the alpha-free diagonal sums D[g] = sum count * log c[code][g] (one multiply, one add: the host's
//                       order and rounding), c_other (class "other", in read order), exp(c_other + D[g]) with libm's own exp
//                       (libm_exp_any: oracle/check_exp_restatement.c is its proof), the code histogram, the marker skips of
//                       h:239-249 (absent, depth 0, +-3 sd).
//   pack_layout_kernel  one thread per position of the sorted, padded marker list: run words [tile][row][marker] (plain dictionary
//                       order), panel rows, per-marker constants -- pure data movement.
//   pack_sched_kernel   wide quality alphabets: one 16-lane row per micro-tile places the tile's runs (tile_sched.h), state in LDS.
//   pack_codes16_kernel the 16-bit run lists of the cohort steps, from the 32-bit ones.
#include "llk_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>

#include "tile_sched.h"

namespace vb2 {

namespace {

__device__ const unsigned long long kFlattenExpTab[256] = {
#include "libm_exp_table.inc"
};

// exp() as glibc >= 2.28 computes it on an FMA-capable x86-64, for every argument: resident_kernel.inc's libm_exp (the same
// statements) plus libm's special cases -- tiny and huge arguments, NaN and the infinities, and the results around the
// subnormal range, which libm's `specialcase` rounds once (that helper is compiled without contraction: its two
// multiply-adds are a multiply and an add).  oracle/check_exp_restatement.c: device_exp_any is this routine on the host,
// bit-identical to libm on 6e7 arguments in (-800, 730) and around every edge.
__device__ __forceinline__ double libm_exp_any(double x, const unsigned long long* tab /* LDS copy */)
{
    const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8,
                 NegLn2loN = -0x1.cf79abc9e3b3ap-47, Shift = 0x1.8p52;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
                 C5 = 0x1.1111167a4d017p-7;
    const unsigned long long xb = (unsigned long long)__double_as_longlong(x);
    const unsigned abstop = (unsigned)(xb >> 52) & 0x7ffu;
    bool special = false;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {
        if (abstop < 0x3c9u) return 1.0 + x;                       // |x| < 2^-54
        if (abstop >= 0x409u) {                                    // |x| >= 1024, inf, NaN
            if (xb == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (xb >> 63) ? 0.0 : __longlong_as_double(0x7ff0000000000000ll);
        }
        special = true;
    }
    double kd = fma(InvLn2N, x, Shift);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const double tail = __longlong_as_double((long long)tab[idx]);
    unsigned long long sbits = tab[idx + 1] + (ki << 45);
    const double r2 = r * r;
    const double tmp = fma(r2 * r2, fma(r, C5, C4), fma(r2, fma(r, C3, C2), tail + r));
    if (!special) {
        const double scale = __longlong_as_double((long long)sbits);
        return fma(scale, tmp, scale);
    }
    if ((ki & 0x80000000ull) == 0) {                               // k > 0: the exponent of scale may have overflowed
        sbits -= 1009ull << 52;
        const double scale = __longlong_as_double((long long)sbits);
        return 0x1p1009 * fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                                        // k < 0: take care in the subnormal range
    const double scale = __longlong_as_double((long long)sbits);
    double y = scale + scale * tmp;                                // (-ffp-contract=off: a multiply and an add, like libm's)
    if (y < 1.0) {
        double lo = scale - y + scale * tmp;
        const double hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
}

constexpr int kSchedTilesPerBlock = 4;
struct SchedLdsOps {
    static __device__ __forceinline__ void set(uint64_t& w, uint64_t bits) { atomicOr(reinterpret_cast<unsigned long long*>(&w), (unsigned long long)bits); }
    static __device__ __forceinline__ void clear(uint64_t& w, uint64_t bits) { atomicAnd(reinterpret_cast<unsigned long long*>(&w), ~(unsigned long long)bits); }
};
constexpr int kClassifyThreads = 64;
constexpr int kCodeSlots = 192;            // >= kMaxCode, three 64-bit sets

}  // namespace

// One thread per marker, one wave per workgroup.  LDS: a column of kCodeSlots counters per thread ([code][thread]: the 64
// threads of a wave hit 64 different banks whatever their codes), the byte tables, the logarithm rows, the exp table, the
// workgroup's histogram.  PACKED (no marker of 65 536 reads or more: the host knows from the offsets): two 16-bit counters
// per word -- 24 KB of counters instead of 48, four workgroups per CU instead of two.
template <bool PACKED>
__global__ void __launch_bounds__(kClassifyThreads)
classify_kernel(const ClassifyArgs a)
{
    constexpr int kRows = PACKED ? kCodeSlots / 2 : kCodeSlots;
    __shared__ unsigned cnt[kRows][kClassifyThreads];
    __shared__ unsigned long long s_exp[256];
    __shared__ double s_other[256];
    __shared__ double s_lc3[kMaxCode * 3];
    __shared__ unsigned s_hist[kCodeSlots + 2];
    __shared__ PdDict s_dict;                       // (its row numbers are not assigned yet: the kernel counts steps)
    __shared__ double s_lhet[kNumQual];
    __shared__ unsigned long long s_bound;          // + reads counted (two halves would overflow: kept as 64-bit below)
    __shared__ unsigned long long s_reads, s_others;
    __shared__ unsigned char s_qidx[256];
    const int tid = threadIdx.x;
    for (int e = tid; e < kRows * kClassifyThreads; e += kClassifyThreads) (&cnt[0][0])[e] = 0u;
    for (int e = tid; e < 256; e += kClassifyThreads) {
        s_exp[e] = kFlattenExpTab[e];
        s_other[e] = a.other_lc[e];
        s_qidx[e] = a.qidx[e];
    }
    for (int e = tid; e < kMaxCode * 3; e += kClassifyThreads) s_lc3[e] = a.lc3[e];
    for (int e = tid; e < kCodeSlots + 2; e += kClassifyThreads) s_hist[e] = 0u;
    if (tid == 0) { s_reads = 0ull; s_others = 0ull; s_bound = 0ull; }
    const bool pd = a.pd != 0;
    if (pd) {
        for (int e = tid; e < kNumQual; e += kClassifyThreads) s_lhet[e] = a.lhet[e];
        for (int e = tid; e < (int)(sizeof(PdDict) / 2); e += kClassifyThreads)
            reinterpret_cast<uint16_t*>(&s_dict)[e] = reinterpret_cast<const uint16_t*>(&a.dict)[e];
    }
    __syncthreads();

    const int i = blockIdx.x * kClassifyThreads + tid;
    if (i < a.M) {
        const uint32_t beg = a.off[i], depth = a.off[i + 1] - beg;
        bool counts = depth != 0;
        if (counts && a.sanity && ((double)depth < a.lo || (double)depth > a.hi)) counts = false;
        int32_t eff = -1;
        uint32_t steps_ref = 0, steps_alt = 0;
        double p_other = 0.0;
        if (counts) {
            unsigned alt_up = a.alt[i];
            if (alt_up >= 'a' && alt_up <= 'z') alt_up -= 32;
            const unsigned char* bs = a.bases + beg;
            const unsigned char* qs = a.quals + beg;
            unsigned long long bm0 = 0, bm1 = 0, bm2 = 0;
            double c_other = 0.0;
            unsigned n_other = 0;
            // (eight reads' bytes are requested before the first of them is looked at: one trip to memory per eight reads, not
            // per read; the reads are still taken in order -- c_other's sum is the host's)
            constexpr uint32_t kAhead = 8;
            for (uint32_t j0 = 0; j0 < depth; j0 += kAhead) {
                unsigned bb[kAhead], qq[kAhead];
#pragma unroll
                for (uint32_t u = 0; u < kAhead; ++u) {
                    const uint32_t j = j0 + u < depth ? j0 + u : depth - 1;
                    bb[u] = bs[j];
                    qq[u] = qs[j];
                }
#pragma unroll
                for (uint32_t u = 0; u < kAhead; ++u) {
                    if (j0 + u >= depth) break;
                    const unsigned b = bb[u], qv = qq[u];
                    unsigned up = b;
                    if (up >= 'a' && up <= 'z') up -= 32;
                    const unsigned cls = (b == '.' || b == ',') ? 0u : (up == alt_up ? 1u : 2u);
                    if (cls == 2u) {
                        c_other += s_other[qv];
                        ++n_other;
                        continue;
                    }
                    const unsigned idx = (unsigned)s_qidx[qv] + cls;
                    // (the thread's own column: ds_add_u32, nothing to wait for)
                    if (PACKED) atomicAdd(&cnt[idx >> 1][tid], 1u << ((idx & 1u) * 16u));
                    else atomicAdd(&cnt[idx][tid], 1u);
                    const unsigned long long bit = 1ull << (idx & 63u);
                    if (idx < 64u) bm0 |= bit;
                    else if (idx < 128u) bm1 |= bit;
                    else bm2 |= bit;
                }
            }
            uint16_t* out = a.runs + beg;
            eff = 0;
            double dg0 = 0.0, dg1 = 0.0, dg2 = 0.0, bound = 0.0;
            PdWin win_ref{0u, 0u, 0ull}, win_alt{0u, 0u, 0ull};
            auto count_ref = [&](uint32_t) { ++steps_ref; };
            auto count_alt = [&](uint32_t) { ++steps_alt; };
#pragma unroll 1
            for (int w = 0; w < 3; ++w) {
                unsigned long long bits = w == 0 ? bm0 : w == 1 ? bm1 : bm2;
                while (bits) {
                    const unsigned idx = (unsigned)w * 64u + (unsigned)__builtin_ctzll(bits);
                    bits &= bits - 1ull;
                    unsigned left = PACKED ? (cnt[idx >> 1][tid] >> ((idx & 1u) * 16u)) & 0xffffu : cnt[idx][tid];
                    atomicAdd(&s_hist[idx], left);
                    const double n = (double)left;
                    const double* lc = &s_lc3[idx * 3u];
                    dg0 += n * lc[0]; dg1 += n * lc[1]; dg2 += n * lc[2];
                    if (pd) bound += n * s_lhet[idx >> 1];
                    while (left > 0u) {
                        const unsigned c1 = left > (unsigned)kMaxRunCount ? (unsigned)kMaxRunCount : left;
                        out[eff++] = (uint16_t)(idx | (c1 << 8));
                        left -= c1;
                        if (pd) {
                            if (idx & 1u) pd_run(s_dict, win_alt, idx >> 1, c1, count_alt);
                            else pd_run(s_dict, win_ref, idx >> 1, c1, count_ref);
                        }
                    }
                }
            }
            if (pd) {
                pd_flush(s_dict, win_ref, count_ref);
                pd_flush(s_dict, win_alt, count_alt);
            }
            double* cd = a.cd + (size_t)i * 4;
            cd[0] = c_other;
            cd[1] = libm_exp_any(dg0 + c_other, s_exp);
            cd[2] = libm_exp_any(dg1 + c_other, s_exp);
            cd[3] = libm_exp_any(dg2 + c_other, s_exp);
            if (pd) {
                p_other = libm_exp_any(c_other, s_exp);
                bound += c_other * -0x1.71547652b82fep+0;             // (c_other <= 0: binary orders of magnitude it takes)
                // (steps beyond 16 bits: the marker cannot be laid out -- an impossible bound keeps the context out of the layout)
                if (steps_ref > 0xffffu || steps_alt > 0xffffu) bound = 1e300;
                if (!(bound >= 0.0)) bound = 1e300;                    // (NaN / -inf: quality 0)
                atomicMax(&s_bound, (unsigned long long)__double_as_longlong(bound));
            }
            atomicAdd(&s_reads, (unsigned long long)depth);
            atomicAdd(&s_others, (unsigned long long)n_other);
        }
        a.eff[i] = eff;
        if (pd) {
            a.eff_pd[i] = (steps_ref & 0xffffu) | (steps_alt << 16);
            a.pother[i] = p_other;
        }
    }
    __syncthreads();
    for (int e = tid; e < kMaxCode; e += kClassifyThreads)
        if (s_hist[e]) atomicAdd(&a.hist[e], (unsigned long long)s_hist[e]);
    if (tid == 0) {
        if (s_reads) atomicAdd(&a.hist[kMaxCode], s_reads);
        if (s_others) atomicAdd(&a.hist[kMaxCode + 1], s_others);
        if (pd && s_bound) atomicMax(&a.hist[kMaxCode + 2], s_bound);
    }
}

hipError_t launch_classify(const ClassifyArgs& a, hipStream_t stream)
{
    if (a.M <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.M + kClassifyThreads - 1) / kClassifyThreads)), block(kClassifyThreads);
    if (a.max_depth < 65536u) hipLaunchKernelGGL(classify_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(classify_kernel<false>, grid, block, 0, stream, a);
    return hipGetLastError();
}

// One thread per (micro-tile, row of four runs, marker): the 32-bit run words of `codes` re-coded as
// dictionary index | count code << 8 (count code: llk_kernels.h, codes16).  Rows past a tile's own (and the slack rows at the end) hold padding
// words: the zero table row with count 0.
__global__ void __launch_bounds__(256)
pack_codes16_kernel(const DeviceLayout L, uint2* __restrict__ codes16, const uint2* __restrict__ mt_rec16,
                    uint32_t rows16_total)
{
    const uint32_t pad = (uint32_t)L.num_code;                       // count 0
    const uint32_t pad2 = pad | (pad << 16);
    const int mt = blockIdx.x;
    if (mt >= L.num_mt) {                                            // the last block writes the slack rows
        for (uint32_t e = threadIdx.x; e < (uint32_t)kCodeSlackRows * kMtMarkers; e += blockDim.x)
            codes16[(size_t)rows16_total * kMtMarkers + e] = make_uint2(pad2, pad2);
        return;
    }
    const uint2 r32 = L.mt_rec[mt], r16 = mt_rec16[mt];
    for (uint32_t e = threadIdx.x; e < r16.y * kMtMarkers; e += blockDim.x) {
        const uint32_t row = e / kMtMarkers, m = e % kMtMarkers;
        uint32_t w[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t row32 = 2 * row + h;
            uint2 v = make_uint2(0u, 0u);
            const bool have = row32 < r32.y;
            if (have) v = L.codes[((size_t)r32.x + row32) * kMtMarkers + m];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t rw = j ? v.y : v.x;
                const uint32_t idx = (rw & 0xffffu) / (uint32_t)L.row_bytes;
                // count code: the run word's top half IS the top half of double(n); relative to that of 1.0 it fits a byte
                // (n <= 31 -> <= 0x4f); a padding run (count 0, the zero table row) gets code 0
                const uint32_t top = rw >> 16;
                const uint32_t code = top >= 0x3ff0u ? (top - 0x3ff0u) & 0xffu : 0u;
                w[2 * h + j] = have ? (idx | (code << 8)) : pad;
            }
        }
        codes16[((size_t)r16.x + row) * kMtMarkers + m] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
    }
}

hipError_t launch_pack_codes16(const DeviceLayout& L, uint2* codes16, const uint2* mt_rec16, uint32_t rows16_total,
                               hipStream_t stream)
{
    hipLaunchKernelGGL(pack_codes16_kernel, dim3(L.num_mt + 1), dim3(256), 0, stream, L, codes16, mt_rec16, rows16_total);
    return hipGetLastError();
}

// The cohort steps' 8-bit lists of a probability-domain context (DeviceLayout::codes16 there; eval_body: walk_pd8): one thread per
// (micro-tile, row of four steps, marker) -- the 16-bit offsets of `codes` re-coded as row indices, a byte apiece (an alt step's
// offset loses its kPdAltOffset: the step's position says it is one).  Steps past a tile's own, and the slack rows at the end:
// the padding row.
__global__ void __launch_bounds__(256)
pack_pd_codes8_kernel(const DeviceLayout L, uint32_t* __restrict__ codes8, const uint2* __restrict__ mt_rec8, uint32_t rows8_total)
{
    const uint32_t pad = (uint32_t)L.num_code;
    const uint32_t pad4 = pad | (pad << 8) | (pad << 16) | (pad << 24);
    const int mt = blockIdx.x;
    if (mt >= L.num_mt) {
        for (uint32_t e = threadIdx.x; e < (uint32_t)kCodeSlackRows * kMtMarkers; e += blockDim.x)
            codes8[(size_t)rows8_total * kMtMarkers + e] = pad4;
        return;
    }
    const uint2 r16 = L.mt_rec[mt], r8 = mt_rec8[mt];
    const uint32_t s1 = r16.y & 0xffffu, s2 = r16.y >> 16, rows8 = (s2 + 3u) >> 2;
    const uint16_t* h16 = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint32_t*>(L.codes) + (size_t)r16.x * kMtMarkers);
    for (uint32_t e = threadIdx.x; e < rows8 * kMtMarkers; e += blockDim.x) {
        const uint32_t row = e / kMtMarkers, m = e % kMtMarkers;
        uint32_t w = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t g = 4u * row + j;
            uint32_t idx = pad;
            if (g < s2) {
                uint32_t off = h16[((size_t)(g >> 1) * kMtMarkers + m) * 2 + (g & 1u)];
                if (g >= s1) off -= (uint32_t)kPdAltOffset;
                idx = off / (uint32_t)L.row_bytes;
            }
            w |= idx << (8u * j);
        }
        codes8[((size_t)r8.x + row) * kMtMarkers + m] = w;
    }
}

hipError_t launch_pack_pd_codes8(const DeviceLayout& L, uint32_t* codes8, const uint2* mt_rec8, uint32_t rows8_total, hipStream_t stream)
{
    hipLaunchKernelGGL(pack_pd_codes8_kernel, dim3(L.num_mt + 1), dim3(256), 0, stream, L, codes8, mt_rec8, rows8_total);
    return hipGetLastError();
}

// One thread per position of the padded, sorted marker list (see PackArgs): 16 consecutive threads = one micro-tile, so a
// row of run words leaves as one 128-byte store per tile.
__global__ void __launch_bounds__(256)
pack_layout_kernel(const PackArgs a)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < a.m_pad) {
        const int t = (int)(m / kMtMarkers), lane = (int)(m % kMtMarkers);
        const uint2 rec = a.mt_rec[t];
        const bool have = m < a.m_active;
        const uint32_t eff = have ? a.eff[m] : 0u;
        const uint16_t* src = a.runs + (have ? a.src_off[m] : 0u);
        uint2* out = a.codes + (size_t)rec.x * kMtMarkers + lane;
        for (uint32_t r = 0; r < (a.sched ? 0u : rec.y); ++r) {         // (sched: pack_sched_kernel writes the run words)
            uint32_t w0 = a.pad4, w1 = a.pad4;
            if (2 * r < eff) { const uint32_t rw = src[2 * r]; w0 = a.row_of_idx[rw & 0xffu] | a.hi_of_count[rw >> 8]; }
            if (2 * r + 1 < eff) { const uint32_t rw = src[2 * r + 1]; w1 = a.row_of_idx[rw & 0xffu] | a.hi_of_count[rw >> 8]; }
            out[(size_t)r * kMtMarkers] = make_uint2(w0, w1);
        }
        const int64_t i = have ? (int64_t)a.pidx[m] : 0;
        if (a.kaf_s) a.kaf_s[m] = have ? a.kaf[i] : 0.0;
        else {
            for (int kk = 0; kk < a.k; ++kk) a.ud_s[(size_t)kk * a.m_pad + m] = have ? a.ud[(size_t)i * a.k + kk] : 0.0;
            a.mu_s[m] = have ? a.mu[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) a.cdiag[(size_t)q * a.m_pad + m] = have ? a.cd[(size_t)i * 4 + q] : 0.0;
    }
    // the slack rows behind the last tile (the read loops request past a tile's rows): padding words
    const int64_t nslack = (int64_t)a.slack_rows * kMtMarkers;
    for (int64_t e = m; e < nslack; e += (int64_t)gridDim.x * blockDim.x)
        a.codes[(size_t)a.total_rows * kMtMarkers + e] = make_uint2(a.pad4, a.pad4);
}

// Pass B of a probability-domain context (PackPdArgs): one thread per position of the sorted, padded marker list.
__global__ void __launch_bounds__(256)
pack_pd_kernel(const PackPdArgs a)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < a.m_pad) {
        const int t = (int)(m / kMtMarkers), lane = (int)(m % kMtMarkers);
        const uint2 rec = a.mt_rec[t];
        const uint32_t s1 = rec.y & 0xffffu, s2 = rec.y >> 16;       // ref steps at [0, s1), alt steps at [s1, s2), two to a row
        const bool have = m < a.m_active;
        const uint32_t nrun = have ? a.nrun[m] : 0u;
        const uint16_t* src = a.runs + (have ? a.src_off[m] : 0u);
        uint32_t* out = a.codes + (size_t)rec.x * kMtMarkers + lane;
        uint32_t step = 0, cur = 0;
        auto put = [&](uint32_t off) {
            if (step & 1u) out[(size_t)(step >> 1) * kMtMarkers] = cur | (off << 16);
            else cur = off;
            ++step;
        };
#pragma unroll 1
        for (uint32_t cls = 0; cls < (a.sched ? 0u : 2u); ++cls) {      // (sched: pack_pd_sched_kernel writes the steps)
            PdWin win{0u, 0u, 0ull};
            auto put_row = [&](uint32_t row) { put(row * (uint32_t)a.row_bytes + cls * (uint32_t)kPdAltOffset); };
            for (uint32_t j = 0; j < nrun; ++j) {
                const uint32_t rw = src[j], idx = rw & 0xffu;
                if ((idx & 1u) != cls) continue;
                pd_run(a.dict, win, idx >> 1, rw >> 8, put_row);
            }
            pd_flush(a.dict, win, put_row);
            const uint32_t end = cls == 0 ? s1 : 2u * ((s2 + 1u) >> 1);
            while (step < end) put(a.pad_off + cls * (uint32_t)kPdAltOffset);
        }
        const int64_t i = have ? (int64_t)a.pidx[m] : 0;
        if (a.kaf_s) a.kaf_s[m] = have ? a.kaf[i] : 0.0;
        else {
            for (int kk = 0; kk < a.k; ++kk) a.ud_s[(size_t)kk * a.m_pad + m] = have ? a.ud[(size_t)i * a.k + kk] : 0.0;
            a.mu_s[m] = have ? a.mu[i] : 0.0;
        }
        a.cdiag[m] = have ? a.pother[i] : 0.0;
#pragma unroll
        for (int q = 1; q < 4; ++q) a.cdiag[(size_t)q * a.m_pad + m] = have ? a.cd[(size_t)i * 4 + q] : 0.0;
    }
    const uint32_t pad2 = a.pad_off | (a.pad_off << 16);
    const int64_t nslack = (int64_t)a.slack_rows * kMtMarkers;
    for (int64_t e = m; e < nslack; e += (int64_t)gridDim.x * blockDim.x)
        a.codes[(size_t)a.total_rows * kMtMarkers + e] = pad2;
}

// The steps of a probability-domain tile, placed (PackPdArgs::sched; tile_sched.h): one 16-lane row per micro-tile, four tiles
// per workgroup, phase after phase -- the tile's ref steps into its ref rows, then its alt steps into its alt rows.  A lane's
// steps of the phase (its runs of that class, a run of count c as ceil(c / K) table rows) are staged in LDS as row indices;
// the three passes of the scheduler are those of pack_sched_kernel; the host pack composes them serially: the same bytes.
namespace {
struct SchedIdentity {
    __host__ __device__ int operator[](uint32_t i) const { return (int)i; }
};
}  // namespace

__global__ void __launch_bounds__(4 * kMtMarkers)
pack_pd_sched_kernel(const PackPdArgs a)
{
    constexpr int kTiles = 4;
    __shared__ TileSched s_state[kTiles];
    __shared__ uint8_t s_runs[kTiles][kMtMarkers][kSchedMaxSteps];      // (row indices: <= 162 rows -- bytes, so that seven workgroups fit a CU's LDS and every tile of a C3 sample is in flight at once)
    __shared__ uint8_t s_at[kTiles][kSchedMaxSteps][kMtMarkers];        // (row index + 1; 0 = padding)
    __shared__ uint32_t s_eff[kTiles][kMtMarkers];
    __shared__ uint8_t s_home[kTiles][kSchedMaxPos];
    const int row = threadIdx.x / kMtMarkers, lane = threadIdx.x % kMtMarkers;
    const int t = (int)blockIdx.x * kTiles + row;
    const bool live = t < a.num_mt;
    uint2 rec = make_uint2(0u, 0u);
    uint32_t nrun = 0;
    const uint16_t* src = a.runs;
    if (live) {
        rec = a.mt_rec[t];
        const int64_t m = (int64_t)t * kMtMarkers + lane;
        if (m < a.m_active) {
            nrun = a.nrun[m];
            src = a.runs + a.src_off[m];
        }
    }
    const SchedIdentity ident;
    uint32_t first_step = 0;
    // (a step is a 16-bit half of a word: step g of the tile sits in word g / 2 of the lane's column, half g % 2)
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(a.codes + (size_t)rec.x * kMtMarkers + lane);
    auto put_step = [&](uint32_t g, uint32_t off) { out16[(size_t)(g >> 1) * (kMtMarkers * 2) + (g & 1u)] = (uint16_t)off; };
    const uint32_t s1 = rec.y & 0xffffu, s2 = rec.y >> 16;
#pragma unroll 1
    for (uint32_t cls = 0; cls < 2; ++cls) {
        const int steps = (int)(cls == 0 ? s1 : s2 - s1);
        // the lane's steps of this phase, as row indices
        uint32_t n = 0;
        {
            PdWin win{0u, 0u, 0ull};
            auto stage = [&](uint32_t r) {
                if (n < (uint32_t)kSchedMaxSteps) s_runs[row][lane][n] = (uint8_t)r;
                ++n;
            };
            for (uint32_t j = 0; j < nrun; ++j) {
                const uint32_t rw = src[j], idx = rw & 0xffu;
                if ((idx & 1u) != cls) continue;
                pd_run(a.dict, win, idx >> 1, rw >> 8, stage);
            }
            pd_flush(a.dict, win, stage);
        }
        s_eff[row][lane] = n;
        __syncthreads();
        const bool plain = sched_is_plain(s_eff[row], steps, a.num_code);      // (the same for the 16 lanes of a row)
        if (plain) {
            if (live) {              // the steps in plain order (pack_pd_kernel's loop)
                uint32_t step = 0;
                auto put_plain = [&](uint32_t off) { put_step(first_step + step, off); ++step; };
                PdWin win{0u, 0u, 0ull};
                auto put_row = [&](uint32_t r) { put_plain(r * (uint32_t)a.row_bytes + cls * (uint32_t)kPdAltOffset); };
                for (uint32_t j = 0; j < nrun; ++j) {
                    const uint32_t rw = src[j], idx = rw & 0xffu;
                    if ((idx & 1u) != cls) continue;
                    pd_run(a.dict, win, idx >> 1, rw >> 8, put_row);
                }
                pd_flush(a.dict, win, put_row);
                while (step < (uint32_t)steps) put_plain(a.pad_off + cls * (uint32_t)kPdAltOffset);
            }
        } else {
            TileSched& S = s_state[row];
            const uint64_t all = sched_all_steps(steps);
            for (int d = lane; d <= a.num_code; d += kMtMarkers) {        // (num_code: the padding row's position)
                S.holds[d] = 0;
                s_home[row][d] = (uint8_t)sched_home_step(d < a.num_code ? d : a.num_code - 1, steps, a.num_code);
            }
            S.open[lane] = all;
        }
        __syncthreads();
        auto get = [&](int l, int j) -> uint32_t { return s_runs[row][l][j]; };
        auto put = [&](int l, int c, uint32_t rw) { s_at[row][c][l] = (uint8_t)(rw + 1u); };
        auto pad = [&](int l, int c) { s_at[row][c][l] = 0; };
        auto home = [&](int d) -> int { return s_home[row][d]; };
        const bool side_by_side = sched_home_commutes(steps > 0 ? steps : 1, a.num_code > 0 ? a.num_code : 1);
        if (!plain && side_by_side) sched_home<SchedLdsOps>(s_state[row], lane, n, ident, home, get, put);
        __syncthreads();
        for (int l = 0; l < kMtMarkers; ++l) {
            if (!plain && !side_by_side && lane == l) sched_home<SchedSerialOps>(s_state[row], lane, n, ident, home, get, put);
            __syncthreads();
        }
        for (int l = 0; l < kMtMarkers; ++l) {
            if (!plain && lane == l) sched_rest(s_state[row], lane, n, steps, a.num_code, sched_all_steps(steps), ident, home, get, put);
            __syncthreads();
        }
        if (!plain) sched_pad(s_state[row], lane, steps, pad);
        __syncthreads();
        if (!plain && live) {
            auto off_of = [&](uint32_t v) { return (v ? (v - 1u) * (uint32_t)a.row_bytes : a.pad_off) + cls * (uint32_t)kPdAltOffset; };
            for (int c = 0; c < steps; ++c) put_step(first_step + (uint32_t)c, off_of(s_at[row][c][lane]));
        }
        __syncthreads();
        first_step += (uint32_t)steps;
    }
    if (live && (s2 & 1u)) put_step(s2, a.pad_off + (uint32_t)kPdAltOffset);          // (an odd number of steps: the last word's second half)
}

hipError_t launch_pack_pd(const PackPdArgs& a, hipStream_t stream)
{
    const int64_t n = std::max<int64_t>(a.m_pad, (int64_t)a.slack_rows * kMtMarkers);
    hipLaunchKernelGGL(pack_pd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !a.sched || a.num_mt <= 0) return e;
    hipLaunchKernelGGL(pack_pd_sched_kernel, dim3((unsigned)((a.num_mt + 3) / 4)), dim3(4 * kMtMarkers), 0, stream, a);
    return hipGetLastError();
}

// Wide quality alphabets: one 16-lane row per micro-tile (a wave = four tiles) places the tile's run words -- the phases of
// tile_sched.h, which the host pack composes serially: the same bytes.  The tile's runs are staged in LDS, the schedule
// is built there as 16-bit run words [step][lane] (0 = padding), and leaves as 128-byte rows.

__global__ void __launch_bounds__(kSchedTilesPerBlock * kMtMarkers)
pack_sched_kernel(const PackArgs a)
{
    __shared__ TileSched s_state[kSchedTilesPerBlock];
    __shared__ uint16_t s_runs[kSchedTilesPerBlock][kMtMarkers][kSchedMaxSteps];
    __shared__ uint16_t s_at[kSchedTilesPerBlock][kSchedMaxSteps][kMtMarkers];
    __shared__ uint32_t s_eff[kSchedTilesPerBlock][kMtMarkers];
    __shared__ uint8_t s_home[kSchedTilesPerBlock][kSchedMaxPos];
    __shared__ uint8_t s_dict[kMaxCode];
    const int row = threadIdx.x / kMtMarkers, lane = threadIdx.x % kMtMarkers;
    const int t = (int)blockIdx.x * kSchedTilesPerBlock + row;
    const bool live = t < a.num_mt;
    for (int e = threadIdx.x; e < kMaxCode; e += blockDim.x) s_dict[e] = a.dict_of[e];
    uint2 rec = make_uint2(0u, 0u);
    uint32_t eff = 0;
    const uint16_t* src = a.runs;
    if (live) {
        rec = a.mt_rec[t];
        const int64_t m = (int64_t)t * kMtMarkers + lane;
        if (m < a.m_active) {
            eff = a.eff[m];
            src = a.runs + a.src_off[m];
        }
    }
    s_eff[row][lane] = eff;
    __syncthreads();
    const int steps = (int)(2u * rec.y);
    uint32_t* const out = reinterpret_cast<uint32_t*>(a.codes + (size_t)rec.x * kMtMarkers);
    auto word = [&](uint32_t rw) { return rw ? (a.row_of_idx[rw & 0xffu] | a.hi_of_count[rw >> 8]) : a.pad4; };
    const bool plain = sched_is_plain(s_eff[row], steps, a.num_code);      // (the same for the 16 lanes of a row)
    if (plain) {
        // plain dictionary order, every lane its own marker
        if (live)
            for (int j = 0; j < steps; ++j)
                out[((size_t)(j >> 1) * kMtMarkers + lane) * 2 + (j & 1)] = (uint32_t)j < eff ? word(src[j]) : a.pad4;
    } else {
        TileSched& S = s_state[row];
        for (uint32_t j = 0; j < eff; ++j) s_runs[row][lane][j] = src[j];
        const uint64_t all = sched_all_steps(steps);
        for (int d = lane; d <= a.num_code; d += kMtMarkers) {        // (num_code: the padding row's position)
            S.holds[d] = 0;
            s_home[row][d] = (uint8_t)sched_home_step(d < a.num_code ? d : a.num_code - 1, steps, a.num_code);
        }
        S.open[lane] = all;
    }
    __syncthreads();
    auto get = [&](int l, int j) -> uint32_t { return s_runs[row][l][j]; };
    auto put = [&](int l, int c, uint32_t rw) { s_at[row][c][l] = (uint16_t)rw; };
    auto pad = [&](int l, int c) { s_at[row][c][l] = 0; };
    auto home = [&](int d) -> int { return s_home[row][d]; };
    const bool side_by_side = sched_home_commutes(steps > 0 ? steps : 1, a.num_code > 0 ? a.num_code : 1);
    if (!plain && side_by_side) sched_home<SchedLdsOps>(s_state[row], lane, eff, s_dict, home, get, put);
    __syncthreads();
    // lane after lane: a wave's LDS operations execute in program order, so lane l + 1 finds what lane l left
    for (int l = 0; l < kMtMarkers; ++l) {
        if (!plain && !side_by_side && lane == l) sched_home<SchedSerialOps>(s_state[row], lane, eff, s_dict, home, get, put);
        __syncthreads();
    }
    for (int l = 0; l < kMtMarkers; ++l) {
        if (!plain && lane == l) sched_rest(s_state[row], lane, eff, steps, a.num_code, sched_all_steps(steps), s_dict, home, get, put);
        __syncthreads();
    }
    if (!plain) sched_pad(s_state[row], lane, steps, pad);
    __syncthreads();
    if (!plain && live)
        for (int c = 0; c < steps; c += 2)
            reinterpret_cast<uint2*>(out)[(size_t)(c >> 1) * kMtMarkers + lane] =
                make_uint2(word(s_at[row][c][lane]), word(s_at[row][c + 1][lane]));
}

hipError_t launch_pack_layout(const PackArgs& a, hipStream_t stream)
{
    const int64_t n = std::max<int64_t>(a.m_pad, (int64_t)a.slack_rows * kMtMarkers);
    hipLaunchKernelGGL(pack_layout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !a.sched || a.num_mt <= 0) return e;
    hipLaunchKernelGGL(pack_sched_kernel, dim3((unsigned)((a.num_mt + kSchedTilesPerBlock - 1) / kSchedTilesPerBlock)),
                       dim3(kSchedTilesPerBlock * kMtMarkers), 0, stream, a);
    return hipGetLastError();
}

}  // namespace vb2

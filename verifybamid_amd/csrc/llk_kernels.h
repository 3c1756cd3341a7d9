// llk_kernels.h -- device data layout + kernel launchers (internal).
#ifndef VB2_LLK_KERNELS_H_
#define VB2_LLK_KERNELS_H_

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include <vector>

namespace vb2 {

constexpr int kNumQual = 94;             // Phred 0..93 (ContaminationEstimator.h:65-74)
constexpr int kMaxCode = 2 * kNumQual;   // classes ref/alt x qualities
constexpr int kPadCode = 255;            // never a real dictionary index
constexpr int kMtMarkers = 16;           // markers per micro-tile (one 16-lane group)
constexpr int kMaxBlockWaves = 16;       // 1024-thread blocks at most
constexpr int kMaxGroups = 6;             // groups of 8 points per launch
constexpr int kMaxPointsPerLaunch = 8 * kMaxGroups;
constexpr int kLdsLimitBytes = 160 * 1024;
constexpr int kRowBytesWide = 400;         // 8 points x 6 doubles + 2 doubles of padding (25 16-byte slots: odd)
constexpr int kRowBytesNarrow = 208;       // 4 points x 6 doubles + 2 (13 slots)
constexpr int kMaxWideCodes = 65535 / kRowBytesWide - 1;   // run words hold 16-bit byte offsets (pad row included)
constexpr int kMaxRunCount = 31;           // the 16-bit prefix of a double holds integers up to 31 exactly
// Probability-domain contexts (DeviceLayout::pd, round 6): a table row is P^n of one quality, class ref (class alt reads it
// mirrored), n = 1 .. its quality's K <= kMaxPow; a marker's list is one 16-bit row offset per STEP (a run of count c takes
// ceil(c / K) steps), ref steps first, then alt steps
constexpr int kMaxPow = 8;
// An ALT step's offset points kPdAltOffset bytes into its row: launch shapes of at most four points per table row keep the
// mirror image of the row's values there, so that their read loops need not know a step's class (the 8-point shape, which has
// no room for it, takes the offset off again and names its products the other way round)
constexpr int kPdAltOffset = 192;
constexpr int kPdSampleMarkers = 768;      // markers whose run counts choose the K's (before the reads are walked)
// WINDOW rows (PdDict): the most frequent qualities come in aligned windows of w neighbouring ranks; a window has a row for
// every product P_1^e1 ... P_w^ew, 0 <= e <= E (not all 0), so that a marker's reads of the window's qualities take ONE step
// as long as it holds at most E of each
constexpr int kPdMaxWin = 4;
constexpr int kPdPairSample = 128;         // markers of the sample on which the candidate dictionaries are priced
constexpr double kPdMaxBound = 900.0;      // binary orders of magnitude a marker's likelihood may lie below 1 at most (context.cpp:
                                           // its (het, het) term's, a lower bound of the likelihood that no alpha or PC changes)
// d_ticket of launch_llk_eval: kTicketWords zero-initialised unsigned ints -- [0, kTicketScratchWord) the arrival tickets of a
// launch's passes (llk_eval_passes_kernel; a plain launch uses [0]), two words from kTicketScratchWord on a scratch flag
constexpr int kTicketScratchWord = 8, kTicketWords = 16;
constexpr int kInlinePointDoubles = 96;    // parameter rows that travel as kernel arguments (768 B)

// Everything the kernels read, in HBM.  "Sorted order" = active markers sorted by
// (non-"other") depth, descending; position = micro_tile*16 + m.
struct DeviceLayout {
    const uint2* codes;           // [sum_t mt_rows[t]][16] x {run, run}; run = row byte offset | hi16(double count) << 16
    const uint2* mt_rec;          // [num_mt] {first row, rows = ceil(most runs in the tile / 2)}
    const double* ud;             // [num_pc][m_pad]  (SoA)
    const double* mu;             // [m_pad]
    const double* ediag;          // [4][m_pad]: c_other, exp(c_other+D[g]) for g = 0,1,2
    const double* known_af;       // [m_pad] or nullptr
    const double* dict_perr;      // [num_code] +10^(-q/10) class ref, -10^(-q/10) class alt
    const double2* prim;          // [num_prim] codes whose table rows are computed: {signed pErr, bits: code |
                                  // twin << 16}, twin = the alt code of the same quality (its row is the mirror
                                  // image of this one) or 0xffff
    // The cohort-step copy of the run lists (nullptr unless built: Context::ensure_codes16): the same runs in
    // the same order as 16-bit words -- dictionary index | count code << 8, count code = (top 16 bits of
    // double(count)) - 0x3ff0, i.e. exponent offset << 4 | top four mantissa bits -- four per uint2, stored
    // [micro-tile][step/4][marker].  Half the bytes of `codes`; the kernel pays one multiply and one
    // conversion per run to expand them, which a cohort step (every sample's lists streamed from HBM once per
    // step, 1-2 points each) can afford and a single-sample launch (lists in L2, VALU-bound) cannot.
    // (round 4: one byte permute rebuilds the double 2 * count and one 24-bit multiply the row offset: two
    // instructions per run instead of five; the table of a 16-bit step holds T / 2)
    const uint2* codes16;
    const uint2* mt_rec16;        // [num_mt] {first row, rows = ceil(most runs in the tile / 4)} of codes16
    int32_t num_code;
    int32_t row_bytes;            // LDS table row stride: kRowBytesWide, or kRowBytesNarrow for > kMaxWideCodes codes
    int32_t num_prim;
    int32_t num_mt;
    int32_t num_pc;
    int32_t num_cu;               // compute units of the device
    int32_t dyn_limit;            // work items per wave up to which the waves of a workgroup pull items from an
                                  // LDS queue (per-item result slots); above it the static deal.  Tunables::dyn_tiles
                                  // (10: the queue adapts to the waves' actual speeds), 0 = always static
    int32_t stagger;              // the second half of a workgroup's waves starts its tiles this many x 64
                                  // cycles late (Tunables::stagger)
    int32_t pd;                   // 1: probability-domain layout (see kMaxPow): codes = [rows][16] x uint32 {step, step} of 16-bit
                                  // row offsets, mt_rec = {first row, ref rows | alt rows << 16}, prim = [num_code] {pErr, n},
                                  // ediag[0] = exp(c_other); a workgroup owns PAIRS of neighbouring micro-tiles (owned_tile)
    int32_t num_pair;             // probability domain: the LAST num_pair records of prim make product rows: {bits: row a | row b << 16,
                                  // bits: the row} = the product of two rows built from the records before them (PdDict: windows)
    unsigned long long* stamps;   // profiling aid: [grid][8] wall-clock stamps, or nullptr
    int64_t num_active;
    int64_t m_pad;                // num_mt * 16
    int32_t num_pair2;            // ... and of those the last num_pair2 multiply rows that records of this kind made (a second barrier)
    int32_t reserved1;
};

// Static work schedule of one launch shape: wave w of workgroup b evaluates the work items
// item[off[b * waves + w] .. off[b * waves + w + 1]) (indices into the workgroup's (group, unit)
// item list, sorted by group).  Built on the host from the tiles' row counts (longest-processing-
// time-first over the waves of a workgroup: the balance of a dynamic queue, but a FIXED order, so a
// lane can keep one running product in registers).  Null pointers = snake deal computed in-kernel.
struct Schedule {
    const uint32_t* off;
    const uint16_t* item;
};
// Host side: off/item for `nblk` workgroups of `nwave` waves; tiles_per_unit = micro-tiles a wave
// takes per item (1, 2 or 4), ngrp = point groups of the launch.  rows[t] = rows of micro-tile t.
// Returns false (and leaves the vectors empty) when a workgroup has more than 65535 items.
bool build_schedule(const uint32_t* rows, int num_mt, int nblk, int nwave, int tiles_per_unit, int ngrp,
                    std::vector<uint32_t>* off, std::vector<uint16_t>* item, int own_shift = 0 /* DeviceLayout::pd: owned_tile */);
// Supplies the schedule of a launch shape (mode = wave shape 2..4 of llk_kernels.hip).
struct ScheduleProvider {
    virtual Schedule get(int mode, int ngrp, int grid, int block_waves) = 0;
    virtual ~ScheduleProvider() {}
};

// Which micro-tiles a workgroup owns: blk, blk + nblk, ... -- or, probability-domain contexts, the PAIRS blk, blk + nblk, ... of
// neighbouring tiles (num_mt is even there): the two tiles a wave of the paired shapes walks side by side are then neighbours
// in the sorted order, with the same number of ref and alt rows, so that both halves of the wave are in the same phase.
__host__ __device__ inline uint32_t owned_tile(int sh, uint32_t blk, uint32_t nblk, uint32_t it)
{
    return (((it >> sh) * nblk + blk) << sh) | (it & ((1u << sh) - 1u));
}
__host__ __device__ inline uint32_t owned_count(int sh, uint32_t num_mt, uint32_t blk, uint32_t nblk)
{
    return (((num_mt >> sh) + nblk - 1u - blk) / nblk) << sh;
}
__host__ __device__ inline uint32_t owned_most(int sh, uint32_t num_mt, uint32_t nblk)     // (workgroup 0's count)
{
    return (((num_mt >> sh) + nblk - 1u) / nblk) << sh;
}

// The dictionary of a probability-domain context as the flatten sees it: which table rows exist and what a marker's runs cost.
// Ranks below qp: windows of w ranks; window t's rows start at win_base[t], the row of the exponents (e_0 .. e_{w-1}) at
// + sum e_i (E + 1)^i - 1.  A marker's reads of a window's qualities (per class) take max_i ceil(c_i / E) steps: every step
// takes min(c_i, E) of each.  The other ranks: rows P^1 .. P^K of their own from single_row[rank] on, ceil(c / K) steps.
// Both the step counts (pass A) and the steps (pass B) come from pd_run / pd_flush below: one statement of the rule.
struct PdDict {
    uint16_t win_base[kNumQual];                 // [window] first row
    uint16_t single_row[kNumQual];               // [rank >= qp] row of P^1 (P^2 .. P^K follow)
    uint8_t kpow[kNumQual];                      // [rank >= qp] K
    uint8_t w, e, qp, per_win;                   // window width, highest exponent, ranks in windows (the last window may be partial), rows per window
};
// Row of window t's exponent index idx = sum e_i (E + 1)^i (1 .. per_win).  The rows most markers read -- small exponents -- have
// small indices; ODD windows count downwards, so that two neighbouring windows' frequent rows, which often meet in one step
// of a tile, start in different bank groups (tile_sched.h).
__host__ __device__ __forceinline__ uint32_t pd_win_row(const PdDict& D, uint32_t t, uint32_t idx)
{
    return (uint32_t)D.win_base[t] + ((t & 1u) ? (uint32_t)D.per_win - idx : idx - 1u);
}
struct PdWin {                                   // the open window of a marker's class: its index and the reads of its w qualities
    uint32_t t;
    uint32_t open;
    unsigned long long c;                        // 16 bits per quality
};
template <class Emit>
__host__ __device__ __forceinline__ void pd_flush(const PdDict& D, PdWin& s, Emit&& emit)
{
    if (!s.open) return;
    const uint32_t e = D.e, radix = e + 1u;
    unsigned long long c = s.c;
    while (c) {
        uint32_t idx = 0, mul = 1;
        unsigned long long left = 0;
#pragma unroll
        for (int i = 0; i < kPdMaxWin; ++i) {
            const uint32_t ci = (uint32_t)(c >> (16 * i)) & 0xffffu;
            const uint32_t take = ci < e ? ci : e;
            idx += take * mul;
            mul *= radix;
            left |= (unsigned long long)(ci - take) << (16 * i);
        }
        emit(pd_win_row(D, s.t, idx));
        c = left;
    }
    s.c = 0;
    s.open = 0;
}
// one run of a class: quality rank, n reads (runs come in rank order; a class's walk ends with pd_flush)
template <class Emit>
__host__ __device__ __forceinline__ void pd_run(const PdDict& D, PdWin& s, uint32_t rank, uint32_t n, Emit&& emit)
{
    if (rank < (uint32_t)D.qp) {
        const uint32_t w = D.w, t = rank / w, i = rank - t * w;
        if (s.open && s.t != t) pd_flush(D, s, emit);
        s.open = 1;
        s.t = t;
        s.c += (unsigned long long)n << (16u * i);
        return;
    }
    pd_flush(D, s, emit);
    const uint32_t kq = D.kpow[rank];
    const uint32_t full = (n - 1u) / kq, rem = n - full * kq;
    for (uint32_t f = 0; f < full; ++f) emit((uint32_t)D.single_row[rank] + kq - 1u);
    emit((uint32_t)D.single_row[rank] + rem - 1u);
}

struct LaunchGeom { int grid, block_waves; };
// Workgroups / waves per workgroup used for a launch of 4*btl points.
LaunchGeom launch_geom(const DeviceLayout& L, int btl, int ngrp = 1);
constexpr int kMaxGridPerCU = 2;
constexpr int kCodeSlackRows = 16;        // padding rows after the last tile: the read loop prefetches unconditionally
inline int max_grid(const DeviceLayout& L) { return kMaxGridPerCU * L.num_cu; }

// Enqueue evaluation of num_point candidate rows (pc1 | pc2 | alpha) on stream.
// d_partials: >= (kMaxPointsPerLaunch + 1) * kMaxGridPerCU * L.num_cu doubles of scratch.
// tag_counter: incremented per kernel launch; tags must never repeat on one partials buffer.
// d_ticket: kTicketWords zero-initialised unsigned ints (arrival counters of the single-launch mode: see kTicketScratchWord).
// done_flag: optional word in mapped host memory that receives done_seq after the results of
// the LAST launch are written (lets the host wait without hipStreamSynchronize).
// h_points: the same rows readable by the host (or nullptr): small batches then travel as
// kernel arguments instead of being read from (possibly mapped host) memory.
hipError_t launch_llk_eval(const DeviceLayout& L, int num_point, const double* d_points,
                           const double* h_points, double* d_partials, double* d_out,
                           unsigned int* d_ticket,
                           unsigned long long* done_flag, unsigned long long done_seq,
                           unsigned long long* tag_counter, hipStream_t stream,
                           int reduce_override = 0,      // 1 ticket / 2 tagged for this call only
                           ScheduleProvider* sched = nullptr);

// A cohort step's point counts and parameter rows as kernel arguments (count = doubles valid in v; 0 = the kernel reads
// d_num_valid / d_points): rows of sample s at v[s * np * (2k+1) ..)
constexpr int kMultiInlineDoubles = 432, kMultiInlineSamples = 64;
struct MultiInline {
    int count;
    unsigned char nv[kMultiInlineSamples];
    double v[kMultiInlineDoubles];
};
// One launch over several samples (contexts on the same device): see llk_eval_multi_kernel.
struct MultiLaunch {
    const DeviceLayout* d_layouts;   // [num_sample] in HBM
    const Schedule* d_scheds;        // [num_sample] in HBM (this launch's wave shape), or nullptr
    const double* d_points;          // [num_sample][NP][2k+1], NP = np points per sample
    const int* d_num_valid;          // [num_sample] 0 = sample sits this step out
    double* d_partials;              // [num_sample][NP + 1][bps]; done_seq doubles as the launch tag
    double* d_out;                   // [num_sample][NP]
    unsigned int* d_tickets;         // [num_sample], zero-initialised
    unsigned int* d_batch_done;      // one zero-initialised counter
    unsigned long long* done_flag;   // mapped host word or nullptr
    unsigned long long done_seq;
    unsigned int batch_active;       // samples with num_valid > 0
    int num_sample, bps, block_waves;
    int np;                          // points per sample of this step: 1, 2, 4 or 8 (picks the wave shape)
    bool force_ticket;               // arrival-ticket hand-off instead of tagged sets (the retry after a NaN)
    bool w16;                        // every sample has codes16: stream the 16-bit run lists
    bool pd;                         // every sample is a probability-domain context (DeviceLayout::pd)
    bool all_static;                 // every sample runs the static deal (eval_takes_the_queue is false for all): the
                                     // 16-bit one- and two-point steps then take the kernels with the pipelined item loop
    int ksel;                        // 2 or 4: every sample has --NumPC of that and no known-AF column (the kernels compiled
                                     // for it); 0: the general kernels
    size_t shmem;
    MultiInline inl;                 // the step's counts and rows again, for the kernel-argument segment (count 0: not used)
};
// a launch of this geometry pulls its work items through the LDS queue (else: the static deal)
bool eval_takes_the_queue(const DeviceLayout& L, int nblk, int nwave, int ngrp);
size_t eval_shmem_bytes(const DeviceLayout& L, int btl, int nblk, int block_waves, int ngrp = 1);   // groups of 4*btl points
size_t eval_shmem_np(const DeviceLayout& L, int np, int nblk, int block_waves, int ngrp, int exp_tab_doubles = 0 /* the 16-KiB table */);
int max_groups(const DeviceLayout& L, int btl, int nblk, int block_waves);
// table rows of a probability-domain context of ~M markers that leave a 48-point launch (six point groups) its LDS
int pd_row_budget(int num_marker, int num_pc, int num_cu);
hipError_t launch_llk_eval_multi(const MultiLaunch& ml, hipStream_t stream);
hipError_t launch_fill_zero(double* d_out, int n, hipStream_t stream);
// codes / mt_rec -> codes16 / mt_rec16 on the device (rec16 already holds the tiles' {first row, rows};
// rows16_total + kCodeSlackRows rows are written, the slack as padding words)
hipError_t launch_pack_codes16(const DeviceLayout& L, uint2* codes16, const uint2* mt_rec16, uint32_t rows16_total,
                               hipStream_t stream);
// ... of a probability-domain context: its 16-bit step lists as 8-bit row indices, four steps per 32-bit word (rec8: {first row,
// ref steps | all steps << 16} like mt_rec; the rows of a tile = ceil(all steps / 4))
hipError_t launch_pack_pd_codes8(const DeviceLayout& L, uint32_t* codes8, const uint2* mt_rec8, uint32_t rows8_total, hipStream_t stream);
// The flatten on the device (flatten_kernels.hip).  Pass A, classify_kernel: what the pileup viewer holds goes up as it is --
// bases, qualities, read offsets, alt alleles -- and one thread per marker classifies, counts, run-length codes and sums
// the alpha-free terms; the host keeps the dictionary order (known from a sample of the qualities before any read is
// walked), the sort of the markers by run count and the tile records.
struct ClassifyArgs {
    const unsigned char* bases;    // [reads]
    const unsigned char* quals;    // [reads]
    const uint32_t* off;           // [M + 1] read offsets, relative to the first read
    const unsigned char* alt;      // [M] alt allele
    const unsigned char* qidx;     // [256] quality character -> 2 * rank of its (clamped) quality in the dictionary order
    const double* other_lc;        // [256] quality character -> log c of class "other"
    const double* lc3;             // [kMaxCode][3] code index -> log c[g], g = 0, 1, 2
    uint16_t* runs;                // out [reads]: a marker's runs (code index | count << 8) at its reads' position
    int32_t* eff;                  // out [M]: runs of the marker, -1 = the marker does not count (h:239-249)
    double* cd;                    // out [M][4]: c_other, exp(c_other + D[g])
    unsigned long long* hist;      // out [kMaxCode + 2], zero on entry: reads per code index, reads of counted markers, reads of class "other"
    int32_t M;
    uint32_t max_depth;            // reads of the deepest marker (below 65 536: 16-bit counters)
    int32_t sanity;                // 1: the +-3 sd depth filter is on
    double lo, hi;                 // its bounds
    // probability-domain bookkeeping (pd == 0: none): per marker its steps, ref | alt << 16, under the dictionary chosen
    // from the sample (dict: the K's and the pair bands; its row numbers are not known yet and not needed to count);
    // exp(c_other); and, in hist[kMaxCode + 2], the largest kPdMaxBound-style bound of a counted marker (bits of a
    // non-negative double)
    int32_t pd;
    PdDict dict;
    const double* lhet;            // [kNumQual] quality rank -> -log2 c[1] of the quality: what a read costs the (het, het) term
    uint32_t* eff_pd;              // out [M]
    double* pother;                // out [M]
};
constexpr int kHistWords = kMaxCode + 3;
hipError_t launch_classify(const ClassifyArgs& a, hipStream_t stream);
// Pass B, pack_layout_kernel (+ pack_sched_kernel for wide alphabets): the kernel-order arrays -- run words
// [tile][row][marker], panel rows, per-marker constants -- from the panel-order arrays above.  Pure data movement: the same
// bytes as the host's pass B (tested).
struct PackArgs {
    const uint16_t* runs;          // a marker's runs (dictionary index | count << 8), at src_off[m]
    const uint32_t* src_off;       // [m_active] sorted marker m -> offset of its runs
    const uint32_t* eff;           // [m_active] its number of runs
    const int32_t* pidx;           // [m_active] its panel row
    const double* cd;              // [M][4] panel order: c_other, exp(c_other + D[g])
    const double* ud;              // [M][k] panel order (nullptr with known_af)
    const double* mu;              // [M]
    const double* kaf;             // [M] or nullptr
    const uint2* mt_rec;           // [num_mt] {first row, rows}
    uint2* codes;                  // out: [rows + slack][16]
    double* ud_s;                  // out: [k][m_pad]
    double* mu_s;                  // out: [m_pad]
    double* kaf_s;                 // out: [m_pad] or nullptr
    double* cdiag;                 // out: [4][m_pad]
    int64_t m_active, m_pad;
    int32_t k, num_mt;
    uint32_t total_rows, slack_rows, pad4;
    uint32_t row_of_idx[kMaxCode];
    uint32_t hi_of_count[kMaxRunCount + 1];
    int32_t sched;                 // 1: the run words of a tile are placed by schedule_tile (tile_sched.h: wide alphabets)
    int32_t num_code;
    uint8_t dict_of[kMaxCode];     // dictionary index -> dictionary position (the scheduler's bank groups)
};
hipError_t launch_pack_layout(const PackArgs& a, hipStream_t stream);
// Pass B of a probability-domain context: a marker's runs (dictionary index | count << 8, classes interleaved) become its
// steps -- ref runs first, then alt runs, a run of count c split into ceil(c / K) steps of its quality's rows -- as 16-bit row
// offsets, two per word, [tile][row][marker]; the tile's ref phase padded to rows_ref rows, its alt phase to rows_alt
// (padding = the row of ones).  Panel rows and constants as in pack_layout_kernel, ediag[0] = exp(c_other).
struct PackPdArgs {
    const uint16_t* runs;
    const uint32_t* src_off;       // [m_active]
    const uint32_t* nrun;          // [m_active] runs of the marker
    const int32_t* pidx;           // [m_active]
    const double* cd;              // [M][4]
    const double* pother;          // [M]
    const double* ud;
    const double* mu;
    const double* kaf;
    const uint2* mt_rec;           // [num_mt] {first row, rows_ref | rows_alt << 16}
    uint32_t* codes;               // out: [rows + slack][16]
    double* ud_s;
    double* mu_s;
    double* kaf_s;
    double* cdiag;
    int64_t m_active, m_pad;
    int32_t k, num_mt;
    uint32_t total_rows, slack_rows, pad_off;
    PdDict dict;                   // the table's rows (a step = row index * row_bytes, + kPdAltOffset for class alt)
    int32_t sched;                 // 1: the steps of a tile's phase are placed by schedule_tile (tile_sched.h), as the runs of a
                                   // wide alphabet's tile are: the rows the 16 markers read in a step then start in 16 different
                                   // bank groups (plain order: 1.25 LDS passes per step at 42 codes; placed: 1.01)
    int32_t num_code, row_bytes;
};
hipError_t launch_pack_pd(const PackPdArgs& a, hipStream_t stream);
// n doubles from device memory to mapped host memory, then done_seq to the mapped flag (stream-ordered
// hand-off to a spinning host: see publish_kernel)
hipError_t launch_publish(const double* d_src, double* d_dst_mapped, int n, unsigned long long* done_flag,
                          unsigned long long done_seq, hipStream_t stream);

// Resident search kernel (llk_resident_kernel): launched once per search, fed through a mailbox.
// Word layout of h_cmd: [0] seq, [1] rows valid (0 = exit), [2..2+4*(2k+1)) rows
// (pc1 | pc2 | alpha), [2+4*(2k+1)] check word: XOR of word_hash(word w, w) over [1, last) ^ resident_mix(seq).
// Relay (device memory, control wave -> every workgroup): [0] = round << 16 | slot << 8 | code, code = rows valid
// 1..4, 0 = exit, kRelayEmptyRound = nothing to evaluate; slot = 0: the rows are relay[2..), slot = 1..8: they
// are rows set slot-1 of the SPECULATION buffer of this round (below); ~0 = the control wave gave up.
// Speculation buffers (two, used alternately by round parity, behind the relay words): [0] the round the
// buffer is for, [1] spare, then 8 sets of 4 rows: the next iteration's {R, E, C_A, C_R} for each of the 8
// ways the iteration in flight can end (resident_kernel.inc: DeviceSimplex::speculate).
constexpr unsigned kRelayEmptyRound = 0xfe;
constexpr int kSpecSets = 8;
inline __host__ __device__ int resident_spec_words(int num_pc) { return 2 + kSpecSets * 4 * (2 * num_pc + 1); }
struct ResidentArgs {
    const unsigned long long* h_cmd;     // mailbox in mapped host memory (device view)
    unsigned long long* relay;           // device memory, zero-initialised: resident_words(k) relay words, then the
                                         // two speculation buffers (resident_relay_words(k) in all)
    unsigned long long* spec;            // = relay + resident_words(k)
    double* h_out;                       // [4] results, mapped host memory (device view)
    unsigned long long* h_done;          // completion sequence number, mapped host memory
    unsigned int* h_state;               // 1 running, 2 exited on command, 3 gave up (idle timeout)
    unsigned long long first_seq;        // sequence number of the first command
    unsigned long long timeout_ticks;    // idle limit in 100 MHz wall-clock ticks
    Schedule sched_multi, sched_single;  // static schedules of the 2..4-point and the 1-point wave shape
    // on-device simplex (MINIMIZE commands; resident_kernel.inc)
    double* h_result;                    // result block in mapped host memory: [8 + nmax + 2k] doubles
    unsigned long long epoch;            // distinguishes this launch's hand-off tags from any earlier one's
    int32_t state_off;                   // doubles from the start of dynamic LDS to workgroup 0's search state
    int32_t state_nmax;                  // largest simplex dimension the state was sized for (0 = no MINIMIZE)
    int32_t relay_reps;                  // copies of the relay word in use (1..kRelayReps)
    int32_t extra_ctl;                   // out: the grid has one workgroup more than the tile workgroups, for the control wave (launch_llk_resident)
    // LDS areas behind the search state (offsets in doubles from the start of dynamic LDS; filled in by
    // launch_llk_resident): workgroup 0's staging of the partial sums ([4][grid]; 0 = over the dead tables, round 3),
    // and the workgroup's own run lists (LCACHE, resident_kernel.inc: cache_start).
    int32_t sum_stage_off;
    int32_t cache_off;
    // in: what the run-list cache needs -- the most micro-tiles a workgroup owns (rounded up to even) and the most rows
    // (placement pads and 4 slack rows included); 0 rows = no cache (Tunables::lds_cache 0, or the host saw it cannot fit)
    int32_t cache_tiles;
    int32_t cache_rows;
};
// Rows of LDS a workgroup's run-list cache takes (the placement rule of resident_kernel.inc on the host):
// rows[] = rows per micro-tile, workgroup b of nblk owns tiles b, b + nblk, ...
uint32_t resident_cache_rows(const uint32_t* rows, int num_mt, int nblk, int own_shift = 0);
constexpr unsigned long long kHandoffGiveUpTicks = 25000000ull;   // 0.25 s of the 100 MHz wall clock (tagged hand-off)
constexpr unsigned long long kResidentMinimize = 0xffffffffull;   // mailbox word [1]: a Minimize() request
constexpr int kDeviceSimplexMaxDim = 63;                           // one lane per coordinate, one per vertex (n + 1 <= 64)
__host__ __device__ inline unsigned long long resident_mix(unsigned long long seq)
{
    return seq * 0x9E3779B97F4A7C15ull;      // spreads consecutive sequence numbers over 64 bits
}
// Position-dependent 64-bit hash of one word (xor-shift, multiply, xor-shift).  The check words of the
// mailbox and of the partial-sum hand-off are the XOR of these over the payload, ^ mix(tag):
// a plain XOR of the payload would be blind to equal words (pc1 == pc2 rows, the four equal
// sums of a one-point launch), i.e. to exactly the stale/new mixtures it has to catch.
__host__ __device__ inline unsigned long long word_hash(unsigned long long v, unsigned int pos)
{
    unsigned long long z = v + 0x9E3779B97F4A7C15ull * (unsigned long long)(pos + 1);
    z = (z ^ (z >> 32)) * 0xBF58476D1CE4E5B9ull;
    return z ^ (z >> 29);
}
// [0] seq, [1] kind, payload (4 rows of 2k+1, or a Minimize() request of 8 + n + 4k <= 6k + 9 words), check word
inline __host__ __device__ int resident_words(int num_pc)
{
    const int rows = 4 * (2 * num_pc + 1), req = 6 * num_pc + 9;
    return 2 + (rows > req ? rows : req) + 1;
}
// The relay word every workgroup waits for exists in kRelayReps copies, kRelayRepStride words apart (one address polled
// by all 256 workgroups is one memory channel's queue; workgroup b polls copy b % ResidentArgs::relay_reps)
constexpr int kRelayReps = 64, kRelayRepStride = 512;
inline __host__ __device__ int resident_rep_base(int num_pc)
{
    return (resident_words(num_pc) + 2 * resident_spec_words(num_pc) + kRelayRepStride - 1) / kRelayRepStride * kRelayRepStride;
}
inline __host__ __device__ int resident_relay_words(int num_pc) { return resident_rep_base(num_pc) + kRelayReps * kRelayRepStride; }
// Words of dynamic LDS the resident kernel needs beyond the evaluation body's (search state + command image +
// every workgroup's staging of the round's rows).
size_t resident_state_doubles(int nmax, int num_pc);
// *cooperative: in: ask for hipLaunchCooperativeKernel (every workgroup guaranteed on the CUs together); out: whether
// the kernel went up that way (a refused cooperative launch falls back to a plain one: the bounded waits on both sides
// then turn a partly resident grid into a retry, not a hang)
hipError_t launch_llk_resident(const DeviceLayout& L, ResidentArgs* ra, double* d_partials,
                               unsigned int* d_ticket, hipStream_t stream, bool* cooperative);
// a profiler's tool library (rocprofv3: librocprofiler-sdk-tool) is loaded in this process: ROCm 7.2's crashes in its
// exit handler after a cooperative launch, so the library launches plainly under it
bool profiler_attached();
// calib_kernels.hip: FP64 issue rates of this device with the read loop's instruction mix (lane-instructions per second)
hipError_t measure_issue_ceiling(int device, double out[3]);

}  // namespace vb2
#endif

// Order of a micro-tile's runs for wide quality alphabets (round 4).  Shared by the host pack (context.cpp: VB2_HOST_PACK=1
// and the dry flatten of the CPU tests) and the device pack (llk_kernels.hip: pack_sched_kernel): integer work only, so the
// two write the same bytes.
//
// At step p of a tile's read loop the 16 markers that one ds_read_b128 pass serves read the table rows of their p-th runs.
// Two DIFFERENT rows whose dictionary positions agree mod 16 start in the same group of four LDS banks (the row stride is an
// odd number of 16-byte slots) and are served one after the other; the same row read by several lanes is a broadcast.  With
// every marker's runs in plain dictionary order a step's rows are neighbours as long as the markers hold much the same
// codes -- 21 qualities, 42 codes: 0.4-3 % conflict cycles -- but with 59 equally likely qualities (BAQ-like data, 118 codes)
// a marker holds a random quarter of the codes, the p-th runs of 16 markers are spread over dozens of rows, and 40 % of the
// LDS pipe's cycles were conflicts (profiles/r04/valu_b48_wide.json).  A marker's sums do not care in which order its runs
// are added (any fixed order is as good as the reference's read order to the tolerance the tests hold), so the order is
// free:
//   * every code d has a HOME step h(d) = d * steps / codes -- a step's home codes are consecutive dictionary positions, fewer
//     than 16 of them, so they never collide, and markers that share a code read it in the same step (a broadcast);
//   * a marker's second run of a step (and repeats of a code: runs hold at most 31 reads) moves to the nearest step where
//     the marker is idle and the run's bank group is free -- or already holds the same code; else to the nearest idle step.
// Expected passes per step on the synthetic wide alphabet: 1.85 -> 1.24 (72 codes: 1.38 -> 1.15; 42 codes would go 1.00 ->
// 1.10, so contexts of at most kSchedMinCodes codes keep the plain order).
#pragma once
#include <cstdint>

namespace vb2 {

constexpr int kSchedMaxSteps = 64;     // steps (= 2 x rows) and runs per marker the scheduler handles; beyond: plain order
constexpr int kSchedMinCodes = 48;     // dictionaries up to this size keep the plain order

struct TileSched {                     // scratch of one tile (~1.2 KB)
    uint8_t occ[kSchedMaxSteps][16];   // step -> bank group -> dictionary position + 1 of the code read there (0: free)
    uint64_t used[16];                 // lane -> steps taken
    uint64_t placed[16];               // lane -> runs placed
};

// eff[l]: runs of lane l (0 for a lane without a marker); get(l, j) -> run word (dictionary index | count << 8);
// dict_of[index] -> dictionary position; put(l, step, run word) for every run, pad(l, step) for every step left over.
template <class GetRun, class Put, class Pad>
__host__ __device__ inline void schedule_tile(TileSched& S, const uint32_t* eff, const int steps, const int num_code,
                                              const uint8_t* dict_of, GetRun get, Put put, Pad pad)
{
    bool plain = steps > kSchedMaxSteps || num_code <= 0;
    for (int l = 0; l < 16; ++l) plain = plain || eff[l] > (uint32_t)kSchedMaxSteps || eff[l] > (uint32_t)steps;
    if (plain) {
        for (int l = 0; l < 16; ++l) {
            for (int j = 0; j < steps; ++j) {
                if ((uint32_t)j < eff[l]) put(l, j, get(l, j));
                else pad(l, j);
            }
        }
        return;
    }
    for (int c = 0; c < steps; ++c)
        for (int r = 0; r < 16; ++r) S.occ[c][r] = 0;
    for (int l = 0; l < 16; ++l) S.used[l] = S.placed[l] = 0;
    // the runs that find their home step free
    for (int l = 0; l < 16; ++l)
        for (uint32_t j = 0; j < eff[l]; ++j) {
            const uint32_t rw = get(l, (int)j);
            const int d = dict_of[rw & 0xffu];
            int h = d * steps / num_code;
            h = h < steps ? h : steps - 1;
            const int r = d & 15;
            if (!((S.used[l] >> h) & 1ull) && (S.occ[h][r] == 0 || S.occ[h][r] == (uint8_t)(d + 1))) {
                S.used[l] |= 1ull << h;
                S.occ[h][r] = (uint8_t)(d + 1);
                S.placed[l] |= 1ull << j;
                put(l, h, rw);
            }
        }
    // the others, lane by lane (lane 0 holds the most runs, i.e. the fewest idle steps)
    for (int l = 0; l < 16; ++l)
        for (uint32_t j = 0; j < eff[l]; ++j) {
            if ((S.placed[l] >> j) & 1ull) continue;
            const uint32_t rw = get(l, (int)j);
            const int d = dict_of[rw & 0xffu];
            int h = d * steps / num_code;
            h = h < steps ? h : steps - 1;
            const int r = d & 15;
            int best_same = -1, best_free = -1, best_any = -1;
            for (int k = 0; k < steps && best_same < 0; ++k)
                for (int s = 0; s < (k ? 2 : 1); ++s) {
                    const int c = s ? h + k : h - k;
                    if (c < 0 || c >= steps || ((S.used[l] >> c) & 1ull)) continue;
                    const uint8_t o = S.occ[c][r];
                    if (o == (uint8_t)(d + 1)) { best_same = c; break; }
                    if (best_free < 0 && o == 0) best_free = c;
                    if (best_any < 0) best_any = c;
                }
            const int c = best_same >= 0 ? best_same : best_free >= 0 ? best_free : best_any;
            if (S.occ[c][r] == 0) S.occ[c][r] = (uint8_t)(d + 1);
            S.used[l] |= 1ull << c;
            put(l, c, rw);
        }
    for (int l = 0; l < 16; ++l)
        for (int c = 0; c < steps; ++c)
            if (!((S.used[l] >> c) & 1ull)) pad(l, c);
}

}  // namespace vb2

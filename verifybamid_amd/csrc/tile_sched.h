// Order of a micro-tile's runs for wide quality alphabets.  Shared by the host pack (context.cpp: tunable host_pack
// and the dry flatten of the CPU tests) and the device pack (flatten_kernels.hip: pack_sched_kernel): integer work only, so the
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
//   * the steps a marker will idle in (it has fewer runs than the tile has steps) read the padding row, which has a bank group
//     like any code: they are chosen next, where another marker already idles or the padding row's group is free;
//   * a marker's second run of a step (and repeats of a code: runs hold at most 31 reads) moves to the nearest step where
//     the marker is idle and the run's bank group is free -- or already holds the same code;
//   * if no idle step qualifies, ONE exchange is tried: a run of the marker that sits where this code fits moves to an idle step
//     where ITS code fits, and this run takes its place; only if that fails too does the run go to the nearest idle step and
//     cost a pass.
// Expected passes per step on the synthetic wide alphabet (tools/ubench/sched_sim.cpp): plain order 1.88, scheduled 1.004
// (72 codes: 1.37 -> 1.001; 188 codes at depth 60: 1.97 -> 1.004; without the exchange and with the padding row left to
// chance it was 1.29 / 1.19 / 1.15).  42 codes: 1.004 either way, so contexts of at most kSchedMinCodes codes keep the plain
// order.  Tiles of two to four steps cannot be helped (118 codes over 4 steps: 2.3 -> 1.9).
//
// The state is kept as bit sets over the steps -- per code the steps where it holds its bank group, per bank group the
// steps where the group is free, per lane the steps taken -- so that "the nearest step where ..." is a handful of bit
// operations instead of a walk, and the work comes in three phases per lane (sched_home, sched_rest, sched_pad) that the
// device runs with one 16-lane row per tile and the state in LDS (flatten_kernels.hip: pack_sched_kernel; its first
// version ran the whole tile in one thread over byte arrays in private scratch: 8.5 ms per context):
//   * sched_home of different lanes commute whenever a step's home codes fall into distinct bank groups (sched_home_commutes:
//     at most 16 codes per step) -- the only shared words are then set to the same values by whoever comes first -- so the
//     16 lanes run it side by side; otherwise, and always for sched_rest, lane after lane in index order;
//   * sched_pad touches nothing shared.
// schedule_tile below is the serial composition the host uses; both write the same bytes (tested through the digests).
#pragma once
#include <cstdint>

namespace vb2 {

constexpr int kSchedMaxSteps = 64;     // steps (= 2 x rows) and runs per marker the scheduler handles; beyond: plain order
constexpr int kSchedMinCodes = 48;     // dictionaries up to this size keep the plain order
constexpr int kSchedMaxPos = 192;      // dictionary positions (<= kMaxCode = 188)

struct TileSched {                     // scratch of one tile (2.9 KB)
    uint64_t holds[kSchedMaxPos];      // dictionary position -> steps where that code holds its bank group
    uint64_t open[16];                 // bank group -> steps where the group is still free
    uint64_t used[16];                 // lane -> steps taken
    uint64_t placed[16];               // lane -> runs placed by sched_home
    uint8_t run_at[16][kSchedMaxSteps]; // lane, step -> the lane's run there (valid where `used` says so)
};

struct SchedSerialOps {                // how a shared word changes: plainly on the host, with LDS atomics on the device
    static __host__ __device__ inline void set(uint64_t& w, uint64_t bits) { w |= bits; }
    static __host__ __device__ inline void clear(uint64_t& w, uint64_t bits) { w &= ~bits; }
};

// the set bit of x nearest to position h (x != 0); of two equally near the lower one
__host__ __device__ inline int sched_nearest(uint64_t x, int h)
{
    const uint64_t below = x & ((2ull << h) - 1ull);          // positions <= h (h = 63: 2 << 63 wraps to 0, minus 1 = all)
    const uint64_t above = h >= 63 ? 0ull : x >> (h + 1);     // positions > h, shifted down
    const int lo = below ? 63 - __builtin_clzll(below) : -1;
    const int hi = above ? h + 1 + __builtin_ctzll(above) : -1;
    if (lo < 0) return hi;
    if (hi < 0) return lo;
    return (h - lo) <= (hi - h) ? lo : hi;
}

__host__ __device__ inline uint64_t sched_all_steps(int steps) { return steps >= 64 ? ~0ull : (1ull << steps) - 1ull; }
__host__ __device__ inline int sched_home_step(int d, int steps, int num_code)
{
    const int h = d * steps / num_code;
    return h < steps ? h : steps - 1;
}
// plain dictionary order (no scheduling) for tiles the bit sets cannot describe
template <class Eff>
__host__ __device__ inline bool sched_is_plain(const Eff& eff, int steps, int num_code)
{
    bool plain = steps > kSchedMaxSteps || num_code <= 0 || num_code >= kSchedMaxPos;
    for (int l = 0; l < 16; ++l) plain = plain || eff[l] > (uint32_t)kSchedMaxSteps || eff[l] > (uint32_t)steps;
    return plain;
}
// a step's home codes are consecutive dictionary positions: at most 16 of them = distinct bank groups
__host__ __device__ inline bool sched_home_commutes(int steps, int num_code) { return (num_code + steps - 1) / steps <= 16; }

// lane l's runs that find their home step free.  home(d) = sched_home_step(d, steps, num_code), possibly from a table
template <class Ops, class State, class Dict, class Home, class GetRun, class Put>
__host__ __device__ inline void sched_home(State& S, int l, uint32_t eff_l, const Dict& dict_of, Home home, GetRun get, Put put)
{
    uint64_t used = 0, placed = 0;
    for (uint32_t j = 0; j < eff_l; ++j) {
        const uint32_t rw = get(l, (int)j);
        const int d = dict_of[rw & 0xffu];
        const int h = home(d);
        const uint64_t bit = 1ull << h;
        const int r = d & 15;
        if (!(used & bit) && ((S.open[r] | S.holds[d]) & bit)) {
            used |= bit;
            Ops::set(S.holds[d], bit);                  // (in this order: another lane with the same code never sees neither)
            Ops::clear(S.open[r], bit);
            placed |= 1ull << j;
            S.run_at[l][h] = (uint8_t)j;
            put(l, h, rw);
        }
    }
    S.used[l] = used;
    S.placed[l] = placed;
}

// lane l's other runs: the nearest idle step where the code already holds its bank group, else the nearest idle step where
// the group is free, else the nearest idle step.  Lane after lane (lane 0 holds the most runs, i.e. the fewest idle steps).
template <class State, class Dict, class Home, class GetRun, class Put>
__host__ __device__ inline void sched_rest(State& S, int l, uint32_t eff_l, int steps, int num_code, uint64_t all,
                                           const Dict& dict_of, Home home, GetRun get, Put put)
{
    uint64_t used = S.used[l];
    const uint64_t placed = S.placed[l];
    // The steps the lane will idle in read the padding row -- dictionary position num_code, a bank group like any other.  They are
    // chosen first, as runs of that "code" (towards the end of the tile, where the other lanes' idle steps are): a step where
    // another lane already idles, else one where the padding row's group is free.
    uint64_t padded = 0;
    {
        const int rp = num_code & 15;
        for (int p = (int)eff_l; p < steps; ++p) {
            const uint64_t idle = all & ~used & ~padded;
            const uint64_t same = idle & S.holds[num_code], open = idle & S.open[rp];
            const int c = sched_nearest(same ? same : open ? open : idle, steps - 1);
            const uint64_t bit = 1ull << c;
            if (S.open[rp] & bit) {
                S.open[rp] &= ~bit;
                S.holds[num_code] |= bit;
            }
            padded |= bit;
        }
    }
    for (uint32_t j = 0; j < eff_l; ++j) {
        if ((placed >> j) & 1ull) continue;
        const uint32_t rw = get(l, (int)j);
        const int d = dict_of[rw & 0xffu];
        const int h = home(d);
        const int r = d & 15;
        const uint64_t idle = all & ~used & ~padded;
        const uint64_t same = idle & S.holds[d], open = idle & S.open[r];
        int c = -1;
        if (!same && !open) {
            // Every idle step of the lane has this bank group taken by another code.  One exchange: a run of this lane that
            // sits at a step s2 where THIS code fits (it holds its group there, or the group is free) moves to an idle
            // step s where ITS code fits; this run takes s2.  The first such pair in step order.
            const uint64_t fits_here = used & (S.holds[d] | S.open[r]);
            for (uint64_t cand = fits_here; cand && c < 0; cand &= cand - 1) {
                const int s2 = __builtin_ctzll(cand);
                const int j2 = S.run_at[l][s2];
                const uint32_t rw2 = get(l, j2);
                const int d2 = dict_of[rw2 & 0xffu], r2 = d2 & 15;
                // (the other run gives up its claim only if no other lane shares the cell: it keeps the claim -- the cell
                // stays marked -- so nothing that was conflict-free becomes a conflict)
                const uint64_t to = idle & (S.holds[d2] | S.open[r2]);
                if (!to) continue;
                const int s = sched_nearest(to, home(d2));
                const uint64_t sbit = 1ull << s;
                if (S.open[r2] & sbit) {
                    S.open[r2] &= ~sbit;
                    S.holds[d2] |= sbit;
                }
                S.run_at[l][s] = (uint8_t)j2;
                put(l, s, rw2);
                used |= sbit;
                c = s2;                     // (already in `used`)
            }
        }
        if (c < 0) c = sched_nearest(same ? same : open ? open : idle, h);
        const uint64_t bit = 1ull << c;
        if (S.open[r] & bit) {
            S.open[r] &= ~bit;
            S.holds[d] |= bit;
        }
        used |= bit;
        S.run_at[l][c] = (uint8_t)j;
        put(l, c, rw);
    }
    S.used[l] = used;
}

template <class State, class Pad>
__host__ __device__ inline void sched_pad(State& S, int l, int steps, Pad pad)
{
    const uint64_t used = S.used[l];
    for (int c = 0; c < steps; ++c)
        if (!((used >> c) & 1ull)) pad(l, c);
}

// eff[l]: runs of lane l (0 for a lane without a marker); get(l, j) -> run word (dictionary index | count << 8);
// dict_of[index] -> dictionary position; put(l, step, run word) for every run, pad(l, step) for every step left over.
template <class State, class Eff, class Dict, class GetRun, class Put, class Pad>
inline void schedule_tile(State& S, const Eff& eff, const int steps, const int num_code, const Dict& dict_of, GetRun get,
                          Put put, Pad pad)
{
    if (sched_is_plain(eff, steps, num_code)) {
        for (int l = 0; l < 16; ++l) {
            for (int j = 0; j < steps; ++j) {
                if ((uint32_t)j < eff[l]) put(l, j, get(l, j));
                else pad(l, j);
            }
        }
        return;
    }
    const uint64_t all = sched_all_steps(steps);
    for (int d = 0; d <= num_code; ++d) S.holds[d] = 0;
    for (int r = 0; r < 16; ++r) S.open[r] = all;
    auto home = [&](int d) { return sched_home_step(d, steps, num_code); };
    for (int l = 0; l < 16; ++l) sched_home<SchedSerialOps>(S, l, eff[l], dict_of, home, get, put);
    for (int l = 0; l < 16; ++l) sched_rest(S, l, eff[l], steps, num_code, all, dict_of, home, get, put);
    for (int l = 0; l < 16; ++l) sched_pad(S, l, steps, pad);
}

}  // namespace vb2

// sched_sim.cpp -- LDS passes per read-loop step of a micro-tile's run order (host only, no GPU).
//   g++ -O2 -std=c++17 -I../../verifybamid_amd/csrc -o /tmp/sched_sim sched_sim.cpp && /tmp/sched_sim [q_lo q_hi [markers [depth [skew]]]]
// Synthetic sample like verifybamid_amd/synth.py (qualities uniform in q_lo..q_hi, depth Poisson), flattened the way
// context.cpp does (codes ordered by quality frequency, ref/alt adjacent; markers sorted by run count; 16-marker tiles),
// then every tile's runs placed by (a) plain dictionary order, (b) schedule_tile (tile_sched.h).  For each step the 16
// lanes read one table row each (idle lanes: the padding row); rows whose positions agree mod 16 but differ are served
// one after the other: passes(step) = max over the 16 bank groups of the number of DISTINCT rows in the group.
#define __host__
#define __device__
#include "tile_sched.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace vb2;

int main(int argc, char** argv)
{
    const int q_lo = argc > 1 ? atoi(argv[1]) : 2, q_hi = argc > 2 ? atoi(argv[2]) : 60;
    const int M = argc > 3 ? atoi(argv[3]) : 100000;
    const double depth = argc > 4 ? atof(argv[4]) : 30.0;
    const int skew = argc > 5 ? atoi(argv[5]) : 0;      // 1: qualities peaked near q_hi with a long tail down to q_lo (sequencer-like)
    std::mt19937_64 rng(7);
    std::poisson_distribution<int> pois(depth);
    std::uniform_int_distribution<int> uq(q_lo, q_hi);
    std::uniform_real_distribution<double> u01(0, 1);
    // per marker: counts per (q, class)
    std::vector<std::vector<uint16_t>> runs(M);      // idx | count << 8, idx = 2 * rank(q) + class
    std::vector<long> qh(94, 0);
    struct Rd { uint8_t q, cls; };
    std::vector<std::vector<Rd>> reads(M);
    for (int i = 0; i < M; ++i) {
        const int d = pois(rng);
        const double af = std::min(0.99995, std::max(0.00005, u01(rng)));
        const int g = (u01(rng) < af) + (u01(rng) < af);
        for (int j = 0; j < d; ++j) {
            int q = uq(rng);
            if (skew) {                                  // 70 % within a few units of the top, the rest spread over the range
                std::exponential_distribution<double> ex(0.35);
                if (u01(rng) < 0.7) q = std::max(q_lo, q_hi - (int)ex(rng));
            }
            const bool alt = u01(rng) < g / 2.0;
            const bool err = u01(rng) < std::pow(10.0, -q / 10.0);
            int cls = alt ? 1 : 0;
            if (err) { const double x = u01(rng); cls = x < 1.0 / 3 ? 1 - cls : 2; }
            if (cls == 2) continue;
            reads[i].push_back({(uint8_t)q, (uint8_t)cls});
            ++qh[q];
        }
    }
    int qof[94], qrank[94];
    for (int q = 0; q < 94; ++q) qof[q] = q;
    std::stable_sort(qof, qof + 94, [&](int a, int b) { return qh[a] > qh[b]; });
    for (int r = 0; r < 94; ++r) qrank[qof[r]] = r;
    std::vector<long> hist(188, 0);
    for (int i = 0; i < M; ++i) {
        int cnt[188] = {0};
        for (auto& r : reads[i]) ++cnt[2 * qrank[r.q] + r.cls];
        for (int idx = 0; idx < 188; ++idx) {
            int left = cnt[idx];
            hist[idx] += left;
            while (left > 0) { const int c = std::min(left, 31); runs[i].push_back((uint16_t)(idx | (c << 8))); left -= c; }
        }
    }
    std::vector<uint8_t> dict_of(188, 255);
    int num_code = 0;
    for (int idx = 0; idx < 188; ++idx) if (hist[idx]) dict_of[idx] = (uint8_t)num_code++;
    std::vector<int> perm(M);
    for (int i = 0; i < M; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return runs[a].size() > runs[b].size(); });
    const int num_mt = (M + 15) / 16;
    double tot_steps = 0, pass_plain = 0, pass_sched = 0, pass_sched_nopad = 0, idle = 0;
    TileSched S;
    for (int t = 0; t < num_mt; ++t) {
        uint32_t eff[16];
        const std::vector<uint16_t>* src[16];
        static const std::vector<uint16_t> none;
        for (int l = 0; l < 16; ++l) {
            const int m = t * 16 + l;
            src[l] = m < M ? &runs[perm[m]] : &none;
            eff[l] = (uint32_t)src[l]->size();
        }
        const int steps = 2 * (((int)eff[0] + 1) / 2);
        if (steps == 0) continue;
        std::vector<int> at((size_t)steps * 16, -1);      // step x lane -> dictionary position (-1: pad)
        auto count = [&](bool with_pad) {
            double p = 0;
            for (int c = 0; c < steps; ++c) {
                int best = 1;
                for (int r = 0; r < 16; ++r) {
                    int rows[17], n = 0;
                    for (int l = 0; l < 16; ++l) {
                        int d = at[(size_t)c * 16 + l];
                        if (d < 0) { if (!with_pad) continue; d = num_code; }
                        if ((d & 15) != r) continue;
                        bool seen = false;
                        for (int k = 0; k < n; ++k) seen = seen || rows[k] == d;
                        if (!seen) rows[n++] = d;
                    }
                    best = std::max(best, n);
                }
                p += best;
            }
            return p;
        };
        for (int l = 0; l < 16; ++l)
            for (int c = 0; c < steps; ++c) at[(size_t)c * 16 + l] = (uint32_t)c < eff[l] ? dict_of[(*src[l])[c] & 0xff] : -1;
        pass_plain += count(true);
        std::fill(at.begin(), at.end(), -1);
        schedule_tile(S, eff, steps, num_code, dict_of.data(),
                      [&](int l, int j) -> uint32_t { return (*src[l])[j]; },
                      [&](int l, int c, uint32_t rw) { at[(size_t)c * 16 + l] = dict_of[rw & 0xff]; },
                      [&](int l, int c) { at[(size_t)c * 16 + l] = -1; });
        pass_sched += count(true);
        pass_sched_nopad += count(false);
        for (int v : at) idle += v < 0;
        tot_steps += steps;
    }
    printf("codes %d, steps/tile %.1f, idle lane-steps %.1f %%\n", num_code, tot_steps / num_mt, 100 * idle / (tot_steps * 16));
    printf("passes per step: plain %.3f, scheduled %.3f (ignoring the padding row: %.3f)\n", pass_plain / tot_steps,
           pass_sched / tot_steps, pass_sched_nopad / tot_steps);
    return 0;
}

// host_pipeline.cpp -- CPU cost of the per-sample host stages of vb2_cohort_run, one thread, no GPU:
// read_pileup, sanity_check + resolve, and the flatten of vb2_ctx_create (dry run: plain memory
// instead of the pinned slab, no HIP call).  Under a container CPU quota these milliseconds, not
// the device, bound the samples/s of a cohort read from text.
//   g++ -O2 -std=c++17 -I verifybamid_amd/csrc tools/ubench/host_pipeline.cpp -L verifybamid_amd -lvb2 \
//       -Wl,-rpath,$PWD/verifybamid_amd -o /tmp/host_pipeline && /tmp/host_pipeline <SVDPrefix> <pileup> <NumPC> [reps]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>

#include "context.h"
#include "hostio.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const std::string pre = argv[1], pile = argv[2];
    const int k = std::atoi(argv[3]), reps = argc > 4 ? std::atoi(argv[4]) : 5;
    vb2_run_args a{};
    const std::string ud = pre + ".UD", mu = pre + ".mu", bed = pre + ".bed";
    a.ud_path = ud.c_str(); a.mean_path = mu.c_str(); a.bed_path = bed.c_str(); a.num_pc = k;
    auto panel = std::make_shared<vb2::Panel>();
    panel->numPC = k;
    double t0 = now();
    if (vb2::load_panel(&a, panel.get())) { std::fprintf(stderr, "panel: %s\n", vb2_last_error()); return 1; }
    std::printf("panel %.1f ms\n", 1e3 * (now() - t0));
    double best[4] = {1e9, 1e9, 1e9, 1e9};
    for (int r = 0; r < reps; ++r) {
        t0 = now();
        std::unique_ptr<vb2_flat> f(new vb2_flat(panel));
        if (vb2::read_pileup(pile, *panel, &f->viewer)) { std::fprintf(stderr, "pileup: %s\n", vb2_last_error()); return 1; }
        const double t1 = now();
        f->sanity_disabled = false;
        vb2::sanity_check(*panel, &f->viewer);
        f->resolve();
        const double t2 = now();
        double ms = 0;
        vb2::flatten_dry_run(&f->input, &ms);
        const double t3 = now();
        f.reset();
        const double t4 = now();
        const double v[4] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3};
        for (int i = 0; i < 4; ++i) best[i] = v[i] < best[i] ? v[i] : best[i];
    }
    std::printf("read_pileup %.1f ms, sanity + resolve %.1f ms, flatten (1 thread) %.1f ms, free %.1f ms\n", 1e3 * best[0],
                1e3 * best[1], 1e3 * best[2], 1e3 * best[3]);
    return 0;
}

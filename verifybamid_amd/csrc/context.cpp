// context.cpp -- vb2_ctx: one-time flattening of the pileup + panel into the
// device SoA layout, and the batched likelihood evaluation entry points.
//
// Reference behaviour restated here (file:line relative to the reference root):
//   marker skipping          ContaminationEstimator.h:236-249
//   classifyBase / q clamp   ContaminationEstimator.h:180-184, 296-298
//   Phred table              ContaminationEstimator.h:65-74
//   COND_LK                  ContaminationEstimator.h:164-177
// Everything that does not depend on (alpha, PC) is hoisted out of the
// per-evaluation path: see DESIGN.md "What is computed once".
#include "context.h"
#include "tile_sched.h"
#include "tunables.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <queue>
#include <functional>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace vb2 {

thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

#define VB2_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#call) + " failed: " + hipGetErrorString(e_));       \
            return VB2_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

std::atomic<int> g_flatten_thread_cap{0};

namespace {
// Freed device / pinned slabs are kept for the next context of the process instead of going
// back to the driver: hipFree and hipHostFree synchronise with the device and cost milliseconds
// each (a cohort run creates and destroys a context per sample: 5 ms per sample went there).
// Bounded: at most kMaxCached slabs and kMaxCachedBytes per kind (a cohort run has up to three
// groups of 64 contexts alive), and a slab is only reused for a request of at least half its
// size.  Tunables::slab_cache = 0 turns the cache off.
struct SlabCache {
    struct Entry { void* p; size_t bytes; int device; };
    static constexpr size_t kMaxCached = 256, kMaxCachedBytes = (size_t)8 << 30;
    std::mutex mu;
    std::vector<Entry> dev, pin, stage;      // device slabs, small pinned slabs, big pinned upload staging
    bool enabled() { return tunables().slab_cache != 0; }
    void* take(std::vector<Entry>& v, size_t bytes, int device, size_t* got)
    {
        if (!enabled()) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        size_t best = v.size();
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].device == device && v[i].bytes >= bytes && v[i].bytes <= 2 * bytes + (1u << 20) &&
                (best == v.size() || v[i].bytes < v[best].bytes))
                best = i;
        if (best == v.size()) return nullptr;
        void* p = v[best].p;
        *got = v[best].bytes;
        v.erase(v.begin() + (long)best);
        return p;
    }
    bool give(std::vector<Entry>& v, void* p, size_t bytes, int device)
    {
        if (!enabled()) return false;
        std::lock_guard<std::mutex> lk(mu);
        if (v.size() >= kMaxCached) return false;
        size_t held = bytes;
        for (const Entry& e : v) held += e.bytes;
        if (held > kMaxCachedBytes) return false;
        v.push_back(Entry{p, bytes, device});
        return true;
    }
    // streams of destroyed contexts (idle: the context synchronised before giving it back), per device
    std::vector<std::pair<hipStream_t, int>> streams;
    hipStream_t take_stream(int device)
    {
        if (!enabled()) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < streams.size(); ++i)
            if (streams[i].second == device) {
                hipStream_t st = streams[i].first;
                streams.erase(streams.begin() + (long)i);
                return st;
            }
        return nullptr;
    }
    bool give_stream(hipStream_t st, int device)
    {
        if (!enabled()) return false;
        std::lock_guard<std::mutex> lk(mu);
        if (streams.size() >= kMaxCached) return false;
        streams.emplace_back(st, device);
        return true;
    }
    ~SlabCache()
    {
        // process exit: the runtime may already be shutting down; leave the memory to it
    }
};
SlabCache& slab_cache()
{
    static SlabCache* c = new SlabCache();      // intentionally never destroyed (see ~SlabCache)
    return *c;
}
}  // namespace

int usable_cpu_count()
{
    static const int cached = [] {
        long n = (long)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<long>(n, std::max(1, CPU_COUNT(&set)));
        long quota = -1, period = -1;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
            std::fclose(f);
        } else {
            if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
                if (std::fscanf(fq, "%ld", &quota) != 1) quota = -1;
                std::fclose(fq);
            }
            if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(fp, "%ld", &period) != 1) period = -1;
                std::fclose(fp);
            }
        }
        if (quota > 0 && period > 0) n = std::min(n, std::max(1L, (quota + period / 2) / period));
        if (tunables().cpus > 0) n = tunables().cpus;
        return (int)n;
    }();
    return cached;
}

int usable_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0)
            ++ok;
    }
    return ok;
}

namespace {

// P(class | genotype, error?)  [err][geno][class], class 0 ref, 1 alt, 2 other
const double kCond[2][3][3] = {
    {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.0}, {0.0, 1.0, 0.0}},
    {{0.0, 1.0 / 3.0, 2.0 / 3.0}, {1.0 / 6.0, 1.0 / 6.0, 2.0 / 3.0}, {1.0 / 3.0, 0.0, 2.0 / 3.0}},
};

inline unsigned char ascii_upper(unsigned char c) { return (c >= 'a' && c <= 'z') ? (unsigned char)(c - 32) : c; }

// classifyBase (h:180-184) is done through byte tables in Context::create: '.' ',' -> ref,
// toupper(base) == toupper(alt) in the "C" locale -> alt, anything else -> other.
inline int clamp_qual(char qc)
{
    int q = (int)(unsigned char)qc - 33;
    if (q < 0) q = 0;
    else if (q > 93) q = 93;
    return q;
}

}  // namespace

Context::~Context()
{
    if (device >= 0) (void)hipSetDevice(device);
    if (resident_active) resident_end();
    if (stream) (void)hipStreamSynchronize(stream);       // nothing of this context is in flight any more
    // every device array / every pinned, device-mapped buffer of the context: back to the cache
    if (d_slab && !slab_cache().give(slab_cache().dev, d_slab, d_slab_bytes, device)) (void)hipFree(d_slab);
    if (h_slab && !slab_cache().give(slab_cache().pin, h_slab, h_slab_bytes, device)) (void)hipHostFree(h_slab);
    if (h_trace_stage) (void)hipHostFree(h_trace_stage);
    if (d_codes16_own) (void)hipFree(d_codes16_own);
    for (CohortSched& e : cohort_sched_)
        if (e.d_mem && !slab_cache().give(slab_cache().dev, e.d_mem, e.bytes, device)) (void)hipFree(e.d_mem);
    if (own_stream && stream && !slab_cache().give_stream(stream, device)) (void)hipStreamDestroy(stream);
}

void* cached_device_slab(size_t bytes, int device, size_t* got) { return slab_cache().take(slab_cache().dev, bytes, device, got); }
bool recycle_device_slab(void* p, size_t bytes, int device) { return slab_cache().give(slab_cache().dev, p, bytes, device); }
void* cached_pinned_slab(size_t bytes, int device, size_t* got) { return slab_cache().take(slab_cache().pin, bytes, device, got); }
bool recycle_pinned_slab(void* p, size_t bytes, int device) { return slab_cache().give(slab_cache().pin, p, bytes, device); }
hipStream_t cached_stream(int device) { return slab_cache().take_stream(device); }
bool recycle_stream(hipStream_t stream, int device) { return slab_cache().give_stream(stream, device); }
void drop_cached_streams(int device)
{
    while (hipStream_t st = slab_cache().take_stream(device)) (void)hipStreamDestroy(st);
}

static thread_local unsigned long long* t_digest_out = nullptr;

int flatten_digest(const vb2_input* in, unsigned long long* digest)
{
    Context* none = nullptr;
    t_digest_out = digest;
    const int rc = Context::create_impl(in, nullptr, &none, true);
    t_digest_out = nullptr;
    return rc;
}

int flatten_dry_run(const vb2_input* in, double* ms)
{
    Context* none = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = Context::create_impl(in, nullptr, &none, true);
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int Context::create(const vb2_input* in, const vb2_options* opt, Context** out)
{
    return create_impl(in, opt, out, false);
}

// dry: the host half only (classification, dictionary, run packing into plain memory) -- no HIP
// call, nothing returned; tools/ubench/host_pipeline.cpp times it where there is no GPU
int Context::create_impl(const vb2_input* in, const vb2_options* opt, Context** out, bool dry)
{
    *out = nullptr;
    if (!in || in->num_marker < 0 || in->num_pc < 1 || in->num_pc > VB2_MAX_PC || !in->read_off ||
        (!in->known_af && (!in->ud || !in->means))) {
        set_error("vb2_ctx_create: invalid input");
        return VB2_ERR_INVALID;
    }
    if (in->num_marker > (1 << 29) - 1024) {      // (the kernels address a marker's constants with 32-bit byte offsets: 8 x position)
        set_error("vb2_ctx_create: more than 2^29 - 1024 markers");
        return VB2_ERR_INVALID;
    }
    if (in->num_marker > 0 && in->read_off[in->num_marker] > in->read_off[0] && (!in->bases || !in->quals || !in->alt_base)) {
        set_error("vb2_ctx_create: reads without bases / quals / alt_base arrays");
        return VB2_ERR_INVALID;
    }
    int ndev = 0, dev = 0;
    hipDeviceProp_t prop;
    std::memset(&prop, 0, sizeof(prop));
    std::unique_ptr<Context> c(new Context());
    if (!dry) {
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
            (void)hipGetLastError();
            set_error("vb2_ctx_create: no HIP device visible (this library has no CPU fallback)");
            return VB2_ERR_NO_DEVICE;
        }
        dev = opt ? opt->device : -1;
        if (dev < 0) VB2_HIP(hipGetDevice(&dev));
        if (dev >= ndev) {
            set_error("vb2_ctx_create: device ordinal out of range");
            return VB2_ERR_INVALID;
        }
        {   // (asked once per device and process: the query takes a fraction of a millisecond, per context, on a reader thread)
            static std::mutex prop_mu;
            static std::vector<std::pair<int, hipDeviceProp_t>> known;
            std::lock_guard<std::mutex> lk(prop_mu);
            bool have = false;
            for (const auto& e : known)
                if (e.first == dev) { prop = e.second; have = true; break; }
            if (!have) {
                VB2_HIP(hipGetDeviceProperties(&prop, dev));
                known.emplace_back(dev, prop);
            }
        }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_error(std::string("vb2_ctx_create: device is ") + prop.gcnArchName +
                      ", kernels are built for gfx950 only");
            return VB2_ERR_NO_DEVICE;
        }
        VB2_HIP(hipSetDevice(dev));
        std::snprintf(c->device_name, sizeof(c->device_name), "%s", prop.name);
        std::snprintf(c->arch, sizeof(c->arch), "%s", prop.gcnArchName);
        // the context's stream: the upload, every launch and the schedules' copies go on it (never the null stream)
        if (opt && opt->stream) {
            c->stream = (hipStream_t)opt->stream;
            c->own_stream = false;
        } else {
            c->stream = slab_cache().take_stream(dev);
            if (!c->stream) VB2_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
    }
    c->device = dev;
    c->num_marker = in->num_marker;
    c->num_pc = in->num_pc;

    const int M = in->num_marker, k = in->num_pc;
    const Tunables& tn = tunables();
    const bool timing = tn.debug_timing != 0;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = tnow();

    // ---- Phred table, exactly the reference's pow() (h:65-74) ----
    double phred[kNumQual];
    for (int i = 0; i < kNumQual; ++i) phred[i] = std::pow(10.0, i / -10.0);

    // log c[class][q][g], c = E[g][class]*pErr + N[g][class]*pOk: the alpha-free
    // value of the table entry for g1 == g2 (and for class "other", any g1,g2).
    std::vector<double> logc(3 * kNumQual * 3);
    for (int bc = 0; bc < 3; ++bc)
        for (int q = 0; q < kNumQual; ++q)
            for (int g = 0; g < 3; ++g)
                logc[(bc * kNumQual + q) * 3 + g] =
                    std::log(kCond[1][g][bc] * phred[q] + kCond[0][g][bc] * (1.0 - phred[q]));

    // ---- pass A (panel order, sequential reads): which markers count (h:239-249); per marker the
    // runs of equal (class, quality), the alpha-free sums, the code histogram ----
    std::vector<int32_t> active;
    active.reserve(M);
    const double lo = in->avg_depth - 3 * in->sd_depth, hi = in->avg_depth + 3 * in->sd_depth;
    int64_t num_read = 0, num_other = 0;
    std::vector<int32_t> eff_depth;     // runs per marker (see below)
    eff_depth.reserve(M);
    for (int i = 0; i < M; ++i)
        if (in->read_off[i + 1] < in->read_off[i]) {
            set_error("vb2_ctx_create: read_off not monotone");
            return VB2_ERR_INVALID;
        }
    // The flattening is embarrassingly parallel over markers: a few host threads for big inputs.
    const int64_t read_base = M > 0 ? in->read_off[0] : 0;
    const int64_t total_reads = M > 0 ? in->read_off[M] - read_base : 0;
    int nthr = std::min(usable_cpu_count(), 16);
    nthr = (int)std::max<int64_t>(1, std::min<int64_t>(nthr, total_reads / 200000));
    if (const int cap = g_flatten_thread_cap.load()) nthr = std::min(nthr, cap);     // cohort runner: many creates at once
    if (tn.flatten_threads > 0) nthr = tn.flatten_threads;
    auto parallel_for = [&](int64_t n, const std::function<void(int, int64_t, int64_t)>& fn) {
        if (nthr == 1 || n < nthr) { fn(0, 0, n); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(fn, t, n * t / nthr, n * (t + 1) / nthr);
        for (auto& x : th) x.join();
    };

    // Dictionary order of the (class, quality) codes: by quality, the more frequent first, the two
    // classes of a quality next to each other (ref, alt).  A marker's runs are stored in that
    // order, so at a given step the 16 markers a ds_read_b128 pass serves sit on NEIGHBOURING table
    // rows whether they are hom-ref, hom-alt or het -- rows whose 16-byte slots differ mod 16, i.e.
    // no bank conflict.  The frequency ranking comes from a strided sample of the reads (any order
    // is correct; this one only has to be known BEFORE the reads are walked, so that pass A can
    // emit every marker's runs already sorted).  `idx` = 2 * rank(quality) + class below.
    int qrank[kNumQual], qof[kNumQual];
    {
        int64_t qh[kNumQual];
        std::fill(qh, qh + kNumQual, 0);
        const int64_t nsample = std::min<int64_t>(total_reads, 1 << 16);
        const int64_t step = nsample > 0 ? total_reads / nsample : 1;
        if (in->quals)
            for (int64_t j = 0; j < nsample; ++j) ++qh[clamp_qual(in->quals[read_base + j * step])];
        for (int q = 0; q < kNumQual; ++q) qof[q] = q;
        std::stable_sort(qof, qof + kNumQual, [&](int x, int y) { return qh[x] > qh[y]; });
        for (int r = 0; r < kNumQual; ++r) qrank[qof[r]] = r;
    }
    struct Luts {
        uint8_t dot[256], up[256], qidx[256];
        double other_lc[256];
    };
    std::unique_ptr<Luts> lut(new Luts);
    for (int ch = 0; ch < 256; ++ch) {
        lut->dot[ch] = (ch == '.' || ch == ',') ? 1 : 0;
        lut->up[ch] = ascii_upper((unsigned char)ch);
        const int q = clamp_qual((char)ch);
        lut->qidx[ch] = (uint8_t)(2 * qrank[q]);
        lut->other_lc[ch] = logc[(2 * kNumQual + q) * 3];
    }
    std::vector<double> lc3((size_t)kMaxCode * 3);                 // [idx][g]
    for (int idx = 0; idx < kMaxCode; ++idx)
        for (int g = 0; g < 3; ++g) lc3[(size_t)idx * 3 + g] = logc[((idx & 1) * kNumQual + qof[idx >> 1]) * 3 + g];

    // ---- probability-domain layout (round 6; llk_kernels.h: kMaxPow): which powers P^n of a quality's table row exist ----
    // A step multiplies a marker's six products by ONE table row, so a run of count c costs ceil(c / K) steps, K = the
    // highest power its quality has a row for.  The K's must be known before the reads are walked (pass A counts steps),
    // so they come from the run counts of a strided sample of markers: every quality starts at K = 1, and the rows the
    // LDS has room for beyond that (pd_row_budget: the tables of a 48-point launch) go, one at a time, to the quality
    // whose next power saves the most steps.  Any K's are correct; these are the cheapest.
    const bool pd_wanted = tn.pd != 0 && M > 0 && in->bases && in->quals;
    PdDict dict;
    std::memset(&dict, 0, sizeof(dict));
    unsigned char* const kpow = dict.kpow;
    double lhet[kNumQual];
    std::fill(kpow, kpow + kNumQual, (unsigned char)1);
    // What a read can cost the marker's likelihood in binary orders of magnitude: the pair (het, het) explains any ref or alt read
    // with probability c[1] = 0.5 (1 - pErr) + pErr / 6 >= 1 / 6, whatever alpha -- so the likelihood, which holds that pair's term
    // exp(c_other + D[1]) * GF[1] * GF2[1], is at least 2^-(sum of these + the "other" reads' + 27).  Pass A sums them per marker
    // (max_bound below): see where `pd` is decided.
    for (int r = 0; r < kNumQual; ++r) {
        const double pe = phred[qof[r]];
        lhet[r] = -std::log2(0.5 * (1.0 - pe) + pe / 6.0);
    }
    const auto t_k0 = tnow();
    if (pd_wanted) {
        static thread_local std::vector<int64_t> H;                   // [rank][count, 63 = more] runs in the sample
        H.assign((size_t)kNumQual * 64, 0);
        // (the runs of the first kPdPairSample sampled markers, class after class in rank order: the candidate dictionaries are
        // priced on them with the flatten's own rule, pd_run)
        static thread_local std::vector<uint16_t> sruns;              // rank | count << 8 (count <= 255), 0xffff = end of a class
        sruns.clear();
        uint32_t cnt2[2 * kNumQual];
        std::fill(cnt2, cnt2 + 2 * kNumQual, 0u);
        const int stride_m = std::max(1, M / kPdSampleMarkers);
        bool seen[kNumQual];
        std::fill(seen, seen + kNumQual, false);
        int nsampled = 0;
        for (int i = 0; i < M; i += stride_m) {
            const int64_t beg = in->read_off[i], depth = in->read_off[i + 1] - beg;
            if (depth <= 0 || depth > 4096) continue;                  // (deep markers: the context will not take this layout anyway)
            const uint8_t alt_up = lut->up[(unsigned char)in->alt_base[i]];
            uint64_t bm[3] = {0, 0, 0};
            for (int64_t j = 0; j < depth; ++j) {
                const unsigned char b = (unsigned char)in->bases[beg + j];
                const unsigned cls = lut->dot[b] ? 0u : (lut->up[b] == alt_up ? 1u : 2u);
                if (cls == 2u) continue;
                const int idx = (int)lut->qidx[(unsigned char)in->quals[beg + j]] + (int)cls;
                ++cnt2[idx];
                bm[idx >> 6] |= 1ull << (idx & 63);
            }
            const bool keep = nsampled < kPdPairSample;
            for (uint32_t cls = 0; cls < 2; ++cls) {
                for (int w = 0; w < 3; ++w)
                    for (uint64_t bits = bm[w]; bits; bits &= bits - 1) {
                        const int idx = w * 64 + __builtin_ctzll(bits);
                        if ((uint32_t)(idx & 1) != cls) continue;
                        if (keep) sruns.push_back((uint16_t)((idx >> 1) | (std::min<uint32_t>(cnt2[idx], 255u) << 8)));
                        ++H[(size_t)(idx >> 1) * 64 + std::min<uint32_t>(cnt2[idx], 63u)];
                        seen[idx >> 1] = true;
                        cnt2[idx] = 0;
                    }
                if (keep) sruns.push_back((uint16_t)0xffffu);
            }
            ++nsampled;
        }
        int nseen = 0, qp = 0;
        for (int r = 0; r < kNumQual; ++r) nseen += seen[r] ? 1 : 0;
        while (qp < kNumQual && seen[qp]) ++qp;                         // (the ranks are by frequency: the qualities met are a prefix)
        const int budget = tn.pd_rows > 0 ? tn.pd_rows : pd_row_budget(M, k, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
        // steps of the sampled runs of a quality under K = 1 .. kMaxPow (once), then the greedy walk over the gains
        std::vector<int64_t> st_at((size_t)kNumQual * (kMaxPow + 1), 0);
        for (int r = 0; r < kNumQual; ++r) {
            if (!seen[r]) continue;
            for (int kq = 1; kq <= kMaxPow; ++kq) {
                int64_t st = 0;
                for (int c = 1; c < 64; ++c) st += H[(size_t)r * 64 + c] * ((c + kq - 1) / kq);
                st_at[(size_t)r * (kMaxPow + 1) + kq] = st;
            }
        }
        // The rows the LDS has room for are shared between WINDOWS over the most frequent qualities (PdDict: w ranks, exponents
        // up to E: (E + 1)^w - 1 rows each) and powers P^1 .. P^K of the qualities behind them.  A handful of window shapes,
        // each over as many of the leading ranks as the rows allow; the powers get what is left, greedily; the sampled markers'
        // steps decide.  Any choice is correct; this one is the cheapest the sample knows.
        static const uint8_t kShapes[][2] = {{1, 0}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {3, 1}, {3, 2}, {4, 1}, {3, 3}, {4, 2}};
        PdDict best = dict;
        int64_t best_steps = -1;
        int best_rows = 0;
        for (const auto& shape : kShapes) {
            const int w = shape[0], e = shape[1];
            if (tn.pd_pairs == 0 && e != 0) break;
            PdDict cand;
            std::memset(&cand, 0, sizeof(cand));
            std::fill(cand.kpow, cand.kpow + kNumQual, (unsigned char)1);
            int per_win = 1;
            for (int i = 0; i < w; ++i) per_win *= e + 1;
            per_win -= 1;
            // windows over the leading ranks: as many as leave every other quality met its one row
            int nwin = 0, cover = 0;
            if (e > 0) {
                nwin = (qp + w - 1) / w;
                while (nwin > 0 && 2 + nwin * per_win + std::max(0, nseen - std::min(qp, nwin * w)) > budget) --nwin;
                cover = std::min(qp, nwin * w);
                if (nwin == 0) continue;
            }
            cand.w = (uint8_t)w;
            cand.e = (uint8_t)e;
            cand.qp = (uint8_t)cover;
            cand.per_win = (uint8_t)(e > 0 ? per_win : 0);
            int rows = 2 + nwin * per_win;                               // (2: room for qualities the sample did not meet)
            for (int r = cover; r < kNumQual; ++r) rows += seen[r] ? 1 : 0;
            while (rows < budget) {
                int bst = -1;
                int64_t best_gain = 0;
                for (int r = cover; r < kNumQual; ++r) {
                    if (!seen[r] || cand.kpow[r] >= kMaxPow) continue;
                    const int64_t gain = st_at[(size_t)r * (kMaxPow + 1) + cand.kpow[r]] - st_at[(size_t)r * (kMaxPow + 1) + cand.kpow[r] + 1];
                    if (gain > best_gain) { best_gain = gain; bst = r; }
                }
                if (bst < 0) break;
                ++cand.kpow[bst];
                ++rows;
            }
            int64_t steps = 0;
            {
                PdWin t{0u, 0u, 0ull};
                auto count = [&](uint32_t) { ++steps; };
                for (const uint16_t rw : sruns) {
                    if (rw == 0xffffu) pd_flush(cand, t, count);
                    else pd_run(cand, t, rw & 0xffu, rw >> 8, count);
                }
            }
            if (best_steps < 0 || steps < best_steps || (steps == best_steps && rows < best_rows)) {
                best = cand;
                best_steps = steps;
                best_rows = rows;
            }
        }
        dict = best;
    }

    const auto t_k1 = tnow();
    // a marker's runs, in dictionary order, at the position of its reads (runs <= reads): low byte idx, high byte count
    // (scratch that a thread creating one context after the other keeps: fresh pages cost more
    // than the passes that fill them)
    struct Scratch {
        std::unique_ptr<uint16_t[]> runs;
        std::unique_ptr<double[]> cd;
        size_t runs_cap = 0, cd_cap = 0;
    };
    static thread_local Scratch scratch;
    // The flatten runs on the device (flatten_kernels.hip).  What the pileup viewer holds -- bases, qualities, read offsets,
    // alt alleles -- goes up as it is, with the byte tables and the panel rows in panel order: one copy into a pinned slab,
    // one hipMemcpyAsync.  classify_kernel (pass A) leaves every marker's run list, constants and run count in device
    // memory; the run counts and the code histogram come back (0.4 MB), the host sorts the markers by run count, cuts the
    // tiles, builds the dictionary, and uploads three words per sorted marker; pack_layout_kernel (pass B) writes the
    // kernel-order arrays.  Tunable host_flatten = 1: pass A on the host (below; the checker of the device's, and the only
    // way for an input of 2^32 reads or more), its run lists and constants written straight into the pinned slab the
    // upload leaves from; host_pack = 1: pass B on the host as well, one upload of the finished arrays.
    const bool host_pack_forced = tn.host_pack != 0;
    const bool device_pack_wanted = !dry && !host_pack_forced && M > 0 && total_reads > 0 && total_reads < ((int64_t)1 << 32);
    const bool device_flatten = device_pack_wanted && tn.host_flatten == 0;
    if (device_flatten) nthr = 1;     // (what is left for the host -- three words per sorted marker -- is not worth a thread's start)
    size_t in_total = 0;
    auto icarve = [&](size_t bytes) {
        const size_t off = (in_total + 255) & ~(size_t)255;
        in_total = off + bytes;
        return off;
    };
    const size_t n_reads_al = (size_t)std::max<int64_t>(total_reads, 1);
    // (a) uploaded before pass A (device flatten only)
    const size_t i_bases = icarve(device_flatten ? n_reads_al : 0);
    const size_t i_quals = icarve(device_flatten ? n_reads_al : 0);
    const size_t i_off = icarve(device_flatten ? ((size_t)M + 1) * sizeof(uint32_t) : 0);
    const size_t i_alt = icarve(device_flatten ? (size_t)M : 0);
    const size_t i_qidx = icarve(device_flatten ? 256 : 0);
    const size_t i_olc = icarve(device_flatten ? 256 * sizeof(double) : 0);
    const size_t i_lc3 = icarve(device_flatten ? (size_t)kMaxCode * 3 * sizeof(double) : 0);
    const size_t i_lhet = icarve(device_flatten && pd_wanted ? (size_t)kNumQual * sizeof(double) : 0);
    const size_t i_ud = icarve(in->known_af ? 0 : (size_t)M * k * sizeof(double));
    const size_t i_mu = icarve(in->known_af ? 0 : (size_t)M * sizeof(double));
    const size_t i_kaf = icarve(in->known_af ? (size_t)M * sizeof(double) : 0);
    const size_t up1_end = in_total;
    // (b) uploaded before pass B: three words per sorted marker (sized for every marker: how many are active is not known yet)
    const size_t i_src = icarve((size_t)M * sizeof(uint32_t));
    const size_t i_eff = icarve((size_t)M * sizeof(uint32_t));
    const size_t i_pidx = icarve((size_t)M * sizeof(int32_t));
    const size_t up2_end = in_total;
    // (c) pass A's results: written by the host flatten (and uploaded), or by classify_kernel (eff_all and hist come back)
    const size_t i_effall = icarve(device_flatten ? (size_t)M * sizeof(int32_t) : 0);
    const size_t i_effpd = icarve(device_flatten && pd_wanted ? (size_t)M * sizeof(uint32_t) : 0);
    const size_t i_hist = icarve(device_flatten ? (size_t)kHistWords * sizeof(unsigned long long) : 0);
    const size_t down_end = in_total;
    const size_t i_runs = icarve(n_reads_al * sizeof(uint16_t));
    const size_t i_cd = icarve((size_t)M * 4 * sizeof(double));
    const size_t i_pother = icarve(pd_wanted ? (size_t)M * sizeof(double) : 0);
    in_total = (in_total + 255) & ~(size_t)255;
    const size_t pinned_need = device_flatten ? ((down_end + 255) & ~(size_t)255) : in_total;   // (the device keeps runs and cd to itself)
    struct InGuard {                          // the pinned slab of the pack kernel's inputs: back to the cache on every way out
        char* p = nullptr; size_t bytes = 0; int dev = 0;
        ~InGuard() { if (p && !slab_cache().give(slab_cache().stage, p, bytes, dev)) (void)hipHostFree(p); }
    } in_stage;
    in_stage.dev = dev;
    if (device_pack_wanted) {
        in_stage.p = static_cast<char*>(slab_cache().take(slab_cache().stage, pinned_need, dev, &in_stage.bytes));
        if (!in_stage.p) {
            VB2_HIP(hipHostMalloc((void**)&in_stage.p, pinned_need, hipHostMallocDefault));
            in_stage.bytes = pinned_need;
        }
    }
    struct DevGuard {                          // the flatten's arrays on the device: back to the cache after the create's sync
        void* p = nullptr; size_t bytes = 0; int dev = 0; hipStream_t st = nullptr;
        ~DevGuard() {
            if (!p) return;
            (void)hipStreamSynchronize(st);    // (an early error return: a kernel may still be reading)
            if (!slab_cache().give(slab_cache().dev, p, bytes, dev)) (void)hipFree(p);
        }
    } d_in;
    d_in.dev = dev;
    d_in.st = c->stream;
    if (device_pack_wanted) {
        d_in.p = slab_cache().take(slab_cache().dev, in_total, dev, &d_in.bytes);
        if (!d_in.p) {
            VB2_HIP(hipMalloc(&d_in.p, in_total));
            d_in.bytes = in_total;
        }
    }
    std::vector<int32_t> eff_host(device_flatten ? 0 : M, -1);   // -1: marker does not count
    int32_t* const eff_all = device_flatten ? reinterpret_cast<int32_t*>(in_stage.p + i_effall) : eff_host.data();
    // probability-domain bookkeeping of pass A: a marker's steps (ref | alt << 16) and exp(c_other)
    std::vector<uint32_t> effpd_host(device_flatten || !pd_wanted ? 0 : M, 0u);
    uint32_t* const eff_pd = !pd_wanted ? nullptr : device_flatten ? reinterpret_cast<uint32_t*>(in_stage.p + i_effpd) : effpd_host.data();
    std::vector<double> pother_host((device_flatten || device_pack_wanted || !pd_wanted) ? 0 : M, 0.0);
    double* const pother = !pd_wanted || device_flatten ? nullptr
                           : device_pack_wanted ? reinterpret_cast<double*>(in_stage.p + i_pother) : pother_host.data();
    double max_bound = 0.0;
    const size_t runs_need = (size_t)std::max<int64_t>(total_reads, 1), cd_need = (size_t)std::max(M, 1) * 4;
    if (!device_pack_wanted) {
        if (scratch.runs_cap < runs_need || scratch.runs_cap > 4 * runs_need + (1u << 20)) {
            scratch.runs.reset(new uint16_t[runs_need]);
            scratch.runs_cap = runs_need;
        }
        if (scratch.cd_cap < cd_need || scratch.cd_cap > 4 * cd_need + (1u << 20)) {
            scratch.cd.reset(new double[cd_need]);
            scratch.cd_cap = cd_need;
        }
    }
    uint16_t* const runs = device_flatten ? nullptr : device_pack_wanted ? reinterpret_cast<uint16_t*>(in_stage.p + i_runs) : scratch.runs.get();
    // c_other, exp(c_other + D[g]) in panel order
    double* const cd_tmp = device_flatten ? nullptr : device_pack_wanted ? reinterpret_cast<double*>(in_stage.p + i_cd) : scratch.cd.get();
    std::vector<int64_t> code_hist(kMaxCode, 0);
    auto t_staged = t_start;
    if (device_flatten) {
        char* const inp = in_stage.p;
        char* const din = static_cast<char*>(d_in.p);
        std::memcpy(inp + i_bases, in->bases + read_base, (size_t)total_reads);
        std::memcpy(inp + i_quals, in->quals + read_base, (size_t)total_reads);
        uint32_t* const off32 = reinterpret_cast<uint32_t*>(inp + i_off);
        uint32_t max_depth = 0;
        for (int i = 0; i <= M; ++i) {
            off32[i] = (uint32_t)(in->read_off[i] - read_base);
            if (i > 0) max_depth = std::max(max_depth, off32[i] - off32[i - 1]);
        }
        std::memcpy(inp + i_alt, in->alt_base, (size_t)M);
        std::memcpy(inp + i_qidx, lut->qidx, 256);
        std::memcpy(inp + i_olc, lut->other_lc, 256 * sizeof(double));
        std::memcpy(inp + i_lc3, lc3.data(), (size_t)kMaxCode * 3 * sizeof(double));
        if (pd_wanted) {
            std::memcpy(inp + i_lhet, lhet, kNumQual * sizeof(double));
        }
        if (in->known_af) std::memcpy(inp + i_kaf, in->known_af, (size_t)M * sizeof(double));
        else {
            std::memcpy(inp + i_ud, in->ud, (size_t)M * k * sizeof(double));
            std::memcpy(inp + i_mu, in->means, (size_t)M * sizeof(double));
        }
        t_staged = tnow();
        VB2_HIP(hipMemcpyAsync(din, inp, up1_end, hipMemcpyHostToDevice, c->stream));
        VB2_HIP(hipMemsetAsync(din + i_hist, 0, (size_t)kHistWords * sizeof(unsigned long long), c->stream));
        ClassifyArgs ca;
        std::memset(&ca, 0, sizeof(ca));
        ca.bases = reinterpret_cast<const unsigned char*>(din + i_bases);
        ca.quals = reinterpret_cast<const unsigned char*>(din + i_quals);
        ca.off = reinterpret_cast<const uint32_t*>(din + i_off);
        ca.alt = reinterpret_cast<const unsigned char*>(din + i_alt);
        ca.qidx = reinterpret_cast<const unsigned char*>(din + i_qidx);
        ca.other_lc = reinterpret_cast<const double*>(din + i_olc);
        ca.lc3 = reinterpret_cast<const double*>(din + i_lc3);
        ca.runs = reinterpret_cast<uint16_t*>(din + i_runs);
        ca.eff = reinterpret_cast<int32_t*>(din + i_effall);
        ca.cd = reinterpret_cast<double*>(din + i_cd);
        ca.hist = reinterpret_cast<unsigned long long*>(din + i_hist);
        ca.M = M;
        ca.max_depth = max_depth;
        ca.sanity = in->sanity_disabled ? 0 : 1;
        ca.lo = lo;
        ca.hi = hi;
        if (pd_wanted) {
            ca.pd = 1;
            ca.dict = dict;
            ca.lhet = reinterpret_cast<const double*>(din + i_lhet);
            ca.eff_pd = reinterpret_cast<uint32_t*>(din + i_effpd);
            ca.pother = reinterpret_cast<double*>(din + i_pother);
        }
        VB2_HIP(launch_classify(ca, c->stream));
        VB2_HIP(hipMemcpyAsync(inp + i_effall, din + i_effall, down_end - i_effall, hipMemcpyDeviceToHost, c->stream));
        VB2_HIP(hipStreamSynchronize(c->stream));
        const unsigned long long* hist = reinterpret_cast<const unsigned long long*>(inp + i_hist);
        for (int c2 = 0; c2 < kMaxCode; ++c2) code_hist[c2] = (int64_t)hist[c2];
        num_read = (int64_t)hist[kMaxCode];
        num_other = (int64_t)hist[kMaxCode + 1];
        std::memcpy(&max_bound, &hist[kMaxCode + 2], sizeof(double));
    } else {
        std::vector<std::vector<int64_t>> hist_t(nthr, std::vector<int64_t>(kMaxCode, 0));
        std::vector<int64_t> reads_t(nthr, 0), other_t(nthr, 0);
        std::vector<double> bound_t(nthr, 0.0);
        // class of a base given the marker's alt allele, one table row per (upper-cased) alt: 0 ref
        // ('.' ','), 1 alt, 2 other
        static const struct ClassTable {
            uint8_t row[256][256];
            ClassTable()
            {
                for (int a = 0; a < 256; ++a)
                    for (int b = 0; b < 256; ++b)
                        row[a][b] = (b == '.' || b == ',') ? 0 : (ascii_upper((unsigned char)b) == a ? 1 : 2);
            }
        } class_table;
        parallel_for(M, [&](int t, int64_t i0, int64_t i1) {
            // (four counter sets, by read index mod 4: consecutive reads mostly carry the same code, and one set
            // would make every increment wait for the previous one's store to the same word -- the loop's
            // critical path is that store-to-load chain, not its instruction count.  Measured and dropped: counting in
            // registers -- "seen once" / "seen twice" bit sets and an overflow array for third occurrences: 11.4 -> 21 ms)
            uint32_t cnt[4][3 * 64];
            std::fill(&cnt[0][0], &cnt[0][0] + 4 * 3 * 64, 0u);
            std::vector<int64_t>& hist = hist_t[t];
            int64_t n_read = 0, n_other = 0;          // thread-local: no shared cache lines in the loop
            double bound_max = 0.0;
            const Luts& T = *lut;
            for (int64_t i = i0; i < i1; ++i) {
                const int64_t beg = in->read_off[i], depth = in->read_off[i + 1] - beg;
                if (depth == 0) continue;
                if (!in->sanity_disabled && ((double)depth < lo || (double)depth > hi)) continue;
                // classifyBase + quality clamp (h:180-184, 296-298) through byte tables
                const uint8_t alt_up = T.up[(unsigned char)in->alt_base[i]];
                const unsigned char* bs = reinterpret_cast<const unsigned char*>(in->bases + beg);
                const unsigned char* qs = reinterpret_cast<const unsigned char*>(in->quals + beg);
                // which codes occur: three 64-bit sets.  The first -- the 32 most frequent qualities -- is kept in a
                // register (an indexed `bm[idx >> 6] |= ...` is a read-modify-write of memory per read: the same
                // store-to-load chain again); the other two are touched by rare qualities only.
                uint64_t bm0 = 0, bm12[2] = {0, 0};
                double c_other = 0.0;                 // class "other": same term for every genotype pair
                const uint8_t* cls_of = class_table.row[alt_up];
                // (Measured and dropped in round 4: the code bytes of 32 reads at a time with AVX2 -- compares for the class,
                // six pshufb tables for the quality rank -- and a code's count as compare + movemask + popcount: same bytes out
                // (digest-checked), classify 11.3 ms against 11.8 for this loop on a C3 sample: with four counter sets the loop
                // already runs at ~4 cycles per read, and the rest of the pass -- per distinct code a histogram update, three
                // dependent multiply-adds, a run word; per marker three exps -- is as long again.)
                for (int64_t j = 0; j < depth; ++j) {
                    const unsigned char qv = qs[j];
                    const unsigned cls = cls_of[bs[j]];
                    if (__builtin_expect(cls == 2, 0)) {
                        c_other += T.other_lc[qv];
                        ++n_other;
                        continue;
                    }
                    const unsigned idx = (unsigned)T.qidx[qv] + cls;
                    ++cnt[j & 3][idx];
                    if (__builtin_expect(idx < 64, 1)) bm0 |= 1ull << idx;
                    else bm12[(idx >> 6) - 1] |= 1ull << (idx & 63);
                }
                const uint64_t bm[3] = {bm0, bm12[0], bm12[1]};
                // steps of this marker in the kernel = runs of equal (class, quality): one
                // (code, count) pair per distinct code, counts above kMaxRunCount split.
                // the g1 == g2 sums, one multiply-add per distinct (class, quality) instead of an add per
                // read (count * log c: the summation order over a marker's reads is free, like the kernel's)
                uint16_t* out = runs + (beg - read_base);
                int32_t eff = 0;
                double dg[3] = {0.0, 0.0, 0.0};
                uint32_t steps_ref = 0, steps_alt = 0;
                double bound = 0.0;
                PdWin win_ref{0u, 0u, 0ull}, win_alt{0u, 0u, 0ull};
                auto count_ref = [&](uint32_t) { ++steps_ref; };
                auto count_alt = [&](uint32_t) { ++steps_alt; };
                for (int w = 0; w < 3; ++w)
                    for (uint64_t bits = bm[w]; bits; bits &= bits - 1) {
                        const unsigned idx = (unsigned)w * 64u + (unsigned)__builtin_ctzll(bits);
                        uint32_t left = cnt[0][idx] + cnt[1][idx] + cnt[2][idx] + cnt[3][idx];
                        cnt[0][idx] = cnt[1][idx] = cnt[2][idx] = cnt[3][idx] = 0;
                        hist[idx] += left;
                        const double n = (double)left;
                        const double* lc = &lc3[(size_t)idx * 3];
                        dg[0] += n * lc[0]; dg[1] += n * lc[1]; dg[2] += n * lc[2];
                        if (pd_wanted) bound += n * lhet[idx >> 1];
                        while (left > 0) {
                            const uint32_t c1 = left > (uint32_t)kMaxRunCount ? (uint32_t)kMaxRunCount : left;
                            out[eff++] = (uint16_t)(idx | (c1 << 8));
                            left -= c1;
                            if (!pd_wanted) continue;
                            if (idx & 1u) pd_run(dict, win_alt, idx >> 1, c1, count_alt);
                            else pd_run(dict, win_ref, idx >> 1, c1, count_ref);
                        }
                    }
                pd_flush(dict, win_ref, count_ref);
                pd_flush(dict, win_alt, count_alt);
                // the g1 == g2 terms of h:307-311 are constants of the marker
                double* cd = cd_tmp + (size_t)i * 4;
                cd[0] = c_other;
                cd[1] = std::exp(dg[0] + c_other);
                cd[2] = std::exp(dg[1] + c_other);
                cd[3] = std::exp(dg[2] + c_other);
                if (pd_wanted) {
                    pother[i] = std::exp(c_other);
                    bound += c_other * -0x1.71547652b82fep+0;
                    if (steps_ref > 0xffffu || steps_alt > 0xffffu) bound = 1e300;
                    if (!(bound >= 0.0)) bound = 1e300;
                    bound_max = std::max(bound_max, bound);
                    eff_pd[i] = (steps_ref & 0xffffu) | (steps_alt << 16);
                }
                n_read += depth;
                eff_all[i] = eff;
            }
            reads_t[t] = n_read;
            other_t[t] = n_other;
            bound_t[t] = bound_max;
        });
        for (int t = 0; t < nthr; ++t) {
            max_bound = std::max(max_bound, bound_t[t]);
            num_read += reads_t[t];
            num_other += other_t[t];
            for (int c2 = 0; c2 < kMaxCode; ++c2) code_hist[c2] += hist_t[t][c2];
        }
    }
    const auto t_pass1 = tnow();
    for (int i = 0; i < M; ++i)
        if (eff_all[i] >= 0) {
            active.push_back(i);
            eff_depth.push_back(eff_all[i]);
        }
    const int64_t m_active = (int64_t)active.size();
    const auto t_act = tnow();

    // ---- dictionary: the observed codes, in idx order (see above) ----
    std::vector<int> order;                                   // dictionary index -> class * kNumQual + quality
    std::vector<uint8_t> dict_of(kMaxCode, (uint8_t)kPadCode); // idx -> dictionary index
    for (int idx = 0; idx < kMaxCode; ++idx)
        if (code_hist[idx] > 0) {
            dict_of[idx] = (uint8_t)order.size();
            order.push_back((idx & 1) * kNumQual + qof[idx >> 1]);
        }
    const int num_code_seen = (int)order.size();              // distinct (class, quality) codes of the data
    // The probability-domain layout is taken when every counted marker's likelihood is bound to stay a normal double far from
    // the bottom of the range (max_bound <= kPdMaxBound: lk >= 2^-(900 + 27)).  The products of UNLIKELY genotype pairs may
    // well underflow in it -- gradually, to subnormals and 0, where the reference's exp() of their sums of logarithms
    // underflows too: either is nothing beside a likelihood of 2^-927 or more (below 2^-95 of it), so the `markerLK > 0` rule
    // (h:310) never decides and the sum's bits are the likely pairs'.  Deep data -- about 850 reads per marker at the usual
    // qualities -- takes the run words and sums of logarithms.  And the table's rows must fit 16-bit offsets.
    int pd_rows = 0, pd_prod_rows = 0, pd_prod2 = 0, pd_per_win = 0, pd_nwin = 0;
    auto pd_has_rows = [&](int r) { return r >= (int)dict.qp && code_hist[2 * r] + code_hist[2 * r + 1] > 0; };
    if (dict.qp > 0) {
        pd_per_win = 1;
        for (int i = 0; i < dict.w; ++i) pd_per_win *= dict.e + 1;
        pd_per_win -= 1;
        pd_nwin = ((int)dict.qp + dict.w - 1) / dict.w;
        pd_rows = pd_nwin * pd_per_win;
    }
    for (int r = 0; r < kNumQual; ++r)
        if (pd_has_rows(r)) pd_rows += kpow[r];
    const bool pd = pd_wanted && m_active > 0 && max_bound <= kPdMaxBound && pd_rows <= kMaxWideCodes && tn.force_narrow == 0;
    int num_code = num_code_seen;                             // rows of the per-alpha table (the padding row not counted)
    std::vector<double> dict_perr;
    std::vector<double2> prim;
    if (!pd) {
        dict_perr.resize(num_code);
        for (int d = 0; d < num_code; ++d) {
            const double pe = phred[order[d] % kNumQual];
            dict_perr[d] = (order[d] / kNumQual) ? -pe : pe;      // sign carries the class
        }
        // ---- primary codes of the per-alpha table: all ref codes (with the alt code of the same
        // quality as twin, if that occurs) and the alt codes without a ref partner ----
        auto prim_rec = [&](int d, uint32_t twin) {
            const unsigned long long bits = (unsigned long long)((uint32_t)d | (twin << 16));
            double y;
            std::memcpy(&y, &bits, sizeof(y));
            prim.push_back(make_double2(dict_perr[d], y));
        };
        for (int idx = 0; idx < kMaxCode; ++idx) {
            const int d = dict_of[idx];
            if (d == kPadCode) continue;
            if ((idx & 1) == 0) {
                const int t = dict_of[idx | 1];
                prim_rec(d, t == kPadCode ? 0xffffu : (uint32_t)t);
            } else if (dict_of[idx & ~1] == kPadCode) {
                prim_rec(d, 0xffffu);
            }
        }
    } else {
        // rows: quality after quality in rank order, P^1 .. P^K of each -- a marker's steps walk the table upwards
        num_code = pd_rows;
        auto bits_of = [](unsigned long long bits) {
            double y;
            std::memcpy(&y, &bits, sizeof(y));
            return y;
        };
        // the windows' rows first (window after window), then the other qualities' P^1 .. P^K
        dict_perr.assign((size_t)pd_rows, 0.0);
        const int radix = dict.e + 1;
        std::vector<double2> prod1, prod2;                     // product records: of two power rows; of rows one of which is a product
        auto prod_rec = [&](uint32_t ra, uint32_t rb, uint32_t dst) {
            return make_double2(bits_of((unsigned long long)(ra | (rb << 16))), bits_of((unsigned long long)dst));
        };
        int next_row = 0;
        for (int t = 0; t < pd_nwin; ++t) {
            const uint32_t base = (uint32_t)next_row;
            dict.win_base[t] = (uint16_t)base;
            int mul[kPdMaxWin + 1];
            mul[0] = 1;
            for (int i = 0; i < kPdMaxWin; ++i) mul[i + 1] = mul[i] * radix;
            const int wt = std::min<int>(dict.w, (int)dict.qp - t * dict.w);      // (the last window may hold fewer qualities: its other rows are never read)
            for (int i = 0; i < wt; ++i) {
                const int r = t * dict.w + i;
                // {pErr, first row | K << 16 | rows between P^n and P^(n+1) << 24}: the quality's own powers inside the window
                // (an odd window's rows count downwards: pd_win_row -- the record's stride is a signed byte)
                const uint32_t first = pd_win_row(dict, (uint32_t)t, (uint32_t)mul[i]);
                const uint32_t stride8 = (uint32_t)(uint8_t)(int8_t)((t & 1) ? -mul[i] : mul[i]);
                prim.push_back(make_double2(phred[qof[r]], bits_of((unsigned long long)(first | ((uint32_t)dict.e << 16) | (stride8 << 24)))));
            }
            for (int idx = 1; idx <= pd_per_win; ++idx) {
                int ex[kPdMaxWin], nz = 0, first_nz[kPdMaxWin];
                for (int i = 0; i < kPdMaxWin; ++i) {
                    ex[i] = i < dict.w ? (idx / mul[i]) % radix : 0;
                    if (ex[i]) first_nz[nz++] = i;
                }
                bool in_window = true;
                for (int i = wt; i < kPdMaxWin; ++i) in_window = in_window && ex[i] == 0;
                if (nz < 2 || !in_window) continue;
                auto row_of = [&](int i0, int i1) {          // the row holding the exponents of positions [i0, i1) of first_nz only
                    int v = 0;
                    for (int q = i0; q < i1; ++q) v += ex[first_nz[q]] * mul[first_nz[q]];
                    return pd_win_row(dict, (uint32_t)t, (uint32_t)v);
                };
                const uint32_t dst = pd_win_row(dict, (uint32_t)t, (uint32_t)idx);
                if (nz == 2) prod1.push_back(prod_rec(row_of(0, 1), row_of(1, 2), dst));
                else if (nz == 3) prod2.push_back(prod_rec(row_of(0, 2), row_of(2, 3), dst));
                else prod2.push_back(prod_rec(row_of(0, 2), row_of(2, 4), dst));
            }
            next_row += pd_per_win;
        }
        for (int r = 0; r < kNumQual; ++r) {
            if (!pd_has_rows(r)) continue;
            prim.push_back(make_double2(phred[qof[r]], bits_of((unsigned long long)((uint32_t)next_row | ((uint32_t)kpow[r] << 16) | (1u << 24)))));
            dict.single_row[r] = (uint16_t)next_row;
            for (int n = 1; n <= kpow[r]; ++n) dict_perr[(size_t)next_row++] = phred[qof[r]];
        }
        pd_prod_rows = (int)(prod1.size() + prod2.size());
        pd_prod2 = (int)prod2.size();
        prim.insert(prim.end(), prod1.begin(), prod1.end());
        prim.insert(prim.end(), prod2.begin(), prod2.end());
    }
    const int num_pair = pd ? pd_prod_rows : 0;

    // ---- sort markers by effective depth (descending, stable); 16-marker micro-tiles ----
    // counting sort = the stable descending sort by run count (ties keep panel order)
    std::vector<int64_t> perm(m_active);
    int num_mt = (int)((m_active + kMtMarkers - 1) / kMtMarkers);
    std::vector<uint32_t> mt_row_off, mt_rows, mt_rec_y;
    uint64_t total_rows = 0;
    if (!pd) {
        int32_t dmax = 0;
        for (int64_t a = 0; a < m_active; ++a) dmax = std::max(dmax, eff_depth[a]);
        std::vector<int64_t> start((size_t)dmax + 2, 0);
        for (int64_t a = 0; a < m_active; ++a) ++start[(size_t)(dmax - eff_depth[a]) + 1];
        for (size_t d = 1; d < start.size(); ++d) start[d] += start[d - 1];
        for (int64_t a = 0; a < m_active; ++a) perm[start[(size_t)(dmax - eff_depth[a])]++] = a;
        // (Measured and dropped, round 4.  Workgroup b owns the micro-tiles b, b + grid, ...: in this plainly descending list it gets
        // the deepest tile of every stripe of `grid` tiles and the last workgroup the shallowest, ~4 % more rows at C3.  (a) Every
        // other stripe of num_cu tiles in ASCENDING order, a snake that evens the workgroups out: 48-point launch 65.17 / 65.47 us
        // plain, 65.30 / 65.27 us snaked on the same box, OptimizeLLK 6.12 / 6.10 ms either way -- the 61-67 us over which the
        // workgroups of a launch finish their tiles follow the CUs (two or three XCDs of a box run slower), not the tile list.
        // (b) The snake plus position 0 of every stripe = the stripe's SHALLOWEST tile, so that workgroup 0 -- it hosts the wave
        // that runs the simplex and is the last to have its block sums in a search round, 9.6 us against a median of 8.3 -- has
        // the least tile work: its block sums were as late as before (its lateness is not tile work) and OptimizeLLK went
        // 6.12 -> 6.22-6.28 ms.)
        mt_row_off.resize(num_mt);
        mt_rows.resize(num_mt);
        for (int t = 0; t < num_mt; ++t) {
            const int32_t dm = eff_depth[perm[(int64_t)t * kMtMarkers]];   // first lane is deepest
            mt_row_off[t] = (uint32_t)total_rows;
            mt_rows[t] = (uint32_t)((dm + 1) / 2);             // two runs per dword
            total_rows += mt_rows[t];
        }
        mt_rec_y = mt_rows;
    } else {
        // Two phases per tile: its markers' ref steps, then their alt steps (class alt reads the table rows mirrored: the
        // second loop of the kernels names the accumulators the other way round).  A tile's phases are as long as its
        // longest marker's, so the markers are sorted by (alt rows, ref steps) -- descending, stable -- and cut into tiles of
        // 16 neighbours; the FULL tiles are then put in descending order of their rows (ties: more alt rows first), so that
        // workgroups and waves take them longest first and the two tiles a paired wave shape walks side by side have the
        // same phases; the last, partial tile stays last (positions past the last marker are at the end of every array).
        // (scratch a thread keeps from one create to the next: fresh pages cost more than the passes that fill them)
        static thread_local std::vector<uint32_t> sref, salt;
        sref.resize(m_active);
        salt.resize(m_active);
        uint32_t ra_max = 0, sr_max = 0;
        for (int64_t a = 0; a < m_active; ++a) {
            const uint32_t e = eff_pd[active[a]];
            sref[a] = e & 0xffffu;
            salt[a] = e >> 16;
            ra_max = std::max(ra_max, salt[a]);
            sr_max = std::max(sr_max, sref[a]);
        }
        static thread_local std::vector<int64_t> p1;
        p1.resize(m_active);
        {
            const uint64_t width = (uint64_t)sr_max + 1, nkey = ((uint64_t)ra_max + 1) * width;
            auto key = [&](int64_t a) { return (uint64_t)salt[a] * width + sref[a]; };
            if (nkey <= (1u << 22)) {
                std::vector<int64_t> start((size_t)nkey + 1, 0);
                for (int64_t a = 0; a < m_active; ++a) ++start[(size_t)(nkey - 1 - key(a)) + 1];
                for (size_t d = 1; d < start.size(); ++d) start[d] += start[d - 1];
                for (int64_t a = 0; a < m_active; ++a) p1[start[(size_t)(nkey - 1 - key(a))]++] = a;
            } else {
                std::iota(p1.begin(), p1.end(), (int64_t)0);
                std::stable_sort(p1.begin(), p1.end(), [&](int64_t x, int64_t y) { return key(x) > key(y); });
            }
        }
        const int full = (int)(m_active / kMtMarkers), tiles = num_mt;
        // (rr: a tile's ref steps, ra: its alt steps -- the longest marker's of either; rows = two steps each)
        std::vector<uint32_t> rr(tiles, 0), ra(tiles, 0);
        for (int t = 0; t < tiles; ++t)
            for (int64_t m = (int64_t)t * kMtMarkers; m < std::min<int64_t>(m_active, (int64_t)(t + 1) * kMtMarkers); ++m) {
                rr[t] = std::max(rr[t], sref[p1[m]]);
                ra[t] = std::max(ra[t], salt[p1[m]]);
            }
        std::vector<int> torder(tiles);
        std::iota(torder.begin(), torder.end(), 0);
        {   // stable, descending by (rows, alt steps): a counting sort when the keys are few (they are), else std::stable_sort
            uint32_t rows_max = 0, ra_mx = 0;
            for (int t = 0; t < full; ++t) {
                rows_max = std::max(rows_max, (rr[t] + ra[t] + 1) / 2);
                ra_mx = std::max(ra_mx, ra[t]);
            }
            const uint64_t wd = (uint64_t)ra_mx + 1, nk = ((uint64_t)rows_max + 1) * wd;
            auto tkey = [&](int t) { return (uint64_t)((rr[t] + ra[t] + 1) / 2) * wd + ra[t]; };
            if (nk <= (1u << 20)) {
                std::vector<int> start((size_t)nk + 1, 0);
                for (int t = 0; t < full; ++t) ++start[(size_t)(nk - 1 - tkey(t)) + 1];
                for (size_t d = 1; d < start.size(); ++d) start[d] += start[d - 1];
                for (int t = 0; t < full; ++t) torder[start[(size_t)(nk - 1 - tkey(t))]++] = t;
            } else {
                std::stable_sort(torder.begin(), torder.begin() + full, [&](int x, int y) { return tkey(x) > tkey(y); });
            }
        }
        if (num_mt & 1) ++num_mt;                               // (a workgroup owns PAIRS of tiles: llk_kernels.h, owned_tile)
        mt_row_off.resize(num_mt);
        mt_rows.assign(num_mt, 0u);
        mt_rec_y.assign(num_mt, 0u);
        for (int t = 0; t < num_mt; ++t) {
            mt_row_off[t] = (uint32_t)total_rows;
            if (t >= tiles) continue;
            const int src_t = torder[t];
            if (rr[src_t] + ra[src_t] > 0xffffu) {     // (ruled out by pass A's bound: steps beyond 16 bits)
                set_error("vb2_ctx_create: a marker needs more than 65535 steps");
                return VB2_ERR_INVALID;
            }
            mt_rows[t] = (rr[src_t] + ra[src_t] + 1) / 2;
            mt_rec_y[t] = rr[src_t] | ((rr[src_t] + ra[src_t]) << 16);      // {ref steps, all steps}
            total_rows += mt_rows[t];
            for (int64_t l = 0; l < kMtMarkers; ++l) {
                const int64_t from = (int64_t)src_t * kMtMarkers + l;
                if (from < m_active) perm[(int64_t)t * kMtMarkers + l] = p1[from];
            }
        }
    }
    const int64_t m_pad = (int64_t)num_mt * kMtMarkers;
    if (total_rows + kCodeSlackRows >= (1ull << 25)) {       // (rows of 128 bytes, addressed by 32-bit byte offsets in the kernels)
        set_error("vb2_ctx_create: input too large for 32-bit row offsets");
        return VB2_ERR_INVALID;
    }

    const auto t_sort = tnow();
    if (timing)
        std::fprintf(stderr, "  create: K choice %.3f ms, active list %.3f ms, dictionary + sort + tiles %.3f ms\n",
                     tms(t_k0, t_k1), tms(t_pass1, t_act), tms(t_act, t_sort));
    const int num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // A run is one dword: low half = byte offset of the code's row in the LDS table (pre-multiplied:
    // the kernel adds it to the table's address), high half = the top 16 bits of the IEEE double
    // `count` (sign, exponent, 4 mantissa bits: exact for 1..31; the kernel masks it into the high
    // word of a double whose low word is 0).  Unused slots: the padding row (zeros) with count +0.0.
    // Two runs per (row, marker) entry.  16-bit offsets reach 163 wide rows; a bigger dictionary
    // gets the narrow rows (and 4-point launches only).
    const bool force_narrow = tn.force_narrow != 0;
    const int row_bytes = (pd || (num_code <= kMaxWideCodes && !force_narrow)) ? kRowBytesWide : kRowBytesNarrow;
    auto run_word = [&](int d, uint32_t n) {
        const double nd = (double)n;
        unsigned long long bits;
        std::memcpy(&bits, &nd, sizeof(bits));
        return (uint32_t)(d * row_bytes) | ((uint32_t)(bits >> 48) << 16);
    };
    const uint32_t pad4 = run_word(num_code, 0);
    const uint32_t pad_off = (uint32_t)(num_code * row_bytes);        // probability domain: the row of ones
    uint32_t row_of_idx[kMaxCode], hi_of_count[kMaxRunCount + 1];     // the two halves of a run word, by table
    for (int idx = 0; idx < kMaxCode; ++idx) row_of_idx[idx] = dict_of[idx] == kPadCode ? 0u : (uint32_t)(dict_of[idx] * row_bytes);
    for (int n = 0; n <= kMaxRunCount; ++n) hi_of_count[n] = run_word(0, (uint32_t)n);
    // wide quality alphabets: a tile's runs are placed by schedule_tile (tile_sched.h) instead of in plain dictionary order
    // (Tunables::run_sched 0: plain order always; 1: scheduled whatever the dictionary's size)
    const bool run_sched = pd ? false : tn.run_sched < 0 ? num_code > kSchedMinCodes : tn.run_sched != 0;
    const bool pd_sched = pd && tn.run_sched != 0;            // (probability domain: a phase's steps are always placed; 0: plain order)

    // ---- device memory: ONE allocation per context, carved into 256-byte aligned pieces.
    // (A cohort creates contexts from many host threads; allocation calls go through driver
    // ioctls under a process-wide lock and were 12 ms per context there, against 0.6 ms alone.)
    // The data arrays come first, in one block: they are written ONCE, straight into a pinned
    // staging slab with the same layout, and go to HBM as a single hipMemcpyAsync on the
    // context's stream (BASELINE.json north_star: "flatten ... into pinned SoA arrays that are
    // hipMemcpyAsync'd to HBM").
    DeviceLayout& L = c->L;
    std::memset(&L, 0, sizeof(L));
    const int nb = kMaxGridPerCU * num_cu;
    const size_t relay_words = (size_t)resident_relay_words(k);
    const bool want_stamps = tn.stamps != 0;
    size_t dev_total = 0;
    auto carve = [&](size_t bytes) {
        const size_t off = (dev_total + 255) & ~(size_t)255;
        dev_total = off + bytes;
        return off;
    };
    const size_t n_codes = (size_t)(total_rows + kCodeSlackRows) * kMtMarkers * (pd ? 1 : 2);   // + prefetch slack (pd: two 16-bit steps per word)
    const size_t o_codes = carve(n_codes * sizeof(uint32_t));
    const size_t o_rec = carve((size_t)num_mt * sizeof(uint2));
    const size_t o_ud = carve(in->known_af ? 0 : (size_t)k * m_pad * sizeof(double));
    const size_t o_mu = carve(in->known_af ? 0 : (size_t)m_pad * sizeof(double));
    const size_t o_kaf = carve(in->known_af ? (size_t)m_pad * sizeof(double) : 0);
    const size_t o_cd = carve((size_t)4 * m_pad * sizeof(double));
    const size_t o_dpe = carve(dict_perr.size() * sizeof(double));
    const size_t o_prim = carve(prim.size() * sizeof(double2));
    // cohort-step run lists (VB2_OPT_COHORT_LAYOUT): the tile records travel with the data block, the
    // 16-bit words themselves are made on the device from `codes` (pack_codes16_kernel)
    const bool want16 = !pd && opt && (opt->flags & VB2_OPT_COHORT_LAYOUT) && num_mt > 0;
    std::vector<uint2> rec16;
    uint64_t total_rows16 = 0;
    if (want16) {
        rec16.resize(num_mt);
        for (int t = 0; t < num_mt; ++t) {
            const uint32_t r16 = (mt_rows[t] + 1) / 2;                  // four runs per row
            rec16[t] = make_uint2((uint32_t)total_rows16, r16);
            total_rows16 += r16;
        }
    }
    const size_t o_rec16 = carve(want16 ? (size_t)num_mt * sizeof(uint2) : 0);
    const size_t data_bytes = (dev_total + 255) & ~(size_t)255;
    const size_t o_codes16 = carve(want16 ? (size_t)(total_rows16 + kCodeSlackRows) * kMtMarkers * sizeof(uint2) : 0);
    const size_t o_part = carve(sizeof(double) * (size_t)(kMaxPointsPerLaunch + 1) * nb);
    const size_t o_ticket = carve(sizeof(unsigned int) * kTicketWords);
    const size_t o_relay = carve(sizeof(unsigned long long) * relay_words);
    const size_t o_stamps = carve(want_stamps ? sizeof(unsigned long long) * 8 * nb : 0);
    // room for the static schedules of the nine launch shapes (filled on first use)
    size_t o_sched[9], sched_bytes[9];
    for (int slot = 0; slot < 9; ++slot) {
        const int tpu = slot == 7 ? 2 : slot == 8 ? 4 : 1, ngrp = slot < 6 ? slot + 1 : 1;
        const size_t items = (size_t)((num_mt + tpu - 1) / tpu + nb) * ngrp;     // (+ per-workgroup round-up)
        sched_bytes[slot] = (((size_t)nb * kMaxBlockWaves + 1) * sizeof(uint32_t) + 15) / 16 * 16 + items * sizeof(uint16_t);
        o_sched[slot] = carve(sched_bytes[slot]);
    }
    dev_total = (dev_total + 255) & ~(size_t)255;

    // pinned staging slab (recycled through the cache like the other slabs: hipHostMalloc /
    // hipHostFree take milliseconds and synchronise)
    const bool device_pack = device_pack_wanted && m_active > 0;
    const size_t stage_need = data_bytes;
    size_t stage_bytes = 0;
    char* stage = dry ? static_cast<char*>(std::malloc(stage_need))
                      : static_cast<char*>(slab_cache().take(slab_cache().stage, stage_need, dev, &stage_bytes));
    if (!stage) {
        if (dry) { set_error("out of host memory"); return VB2_ERR_NOMEM; }
        VB2_HIP(hipHostMalloc((void**)&stage, stage_need, hipHostMallocDefault));
        stage_bytes = stage_need;
    }
    struct StageGuard {                       // back to the cache (or the driver) on every way out
        char* p; size_t bytes; int dev; bool dry;
        ~StageGuard()
        {
            if (dry) std::free(p);
            else if (p && !slab_cache().give(slab_cache().stage, p, bytes, dev)) (void)hipHostFree(p);
        }
    } stage_guard{stage, stage_bytes, dev, dry};
    uint32_t* const codes = reinterpret_cast<uint32_t*>(stage + o_codes);
    uint2* const mt_rec = reinterpret_cast<uint2*>(stage + o_rec);
    double* const ud_s = reinterpret_cast<double*>(stage + o_ud);
    double* const mu_s = reinterpret_cast<double*>(stage + o_mu);
    double* const kaf_s = reinterpret_cast<double*>(stage + o_kaf);
    double* const cdiag = reinterpret_cast<double*>(stage + o_cd);
    for (int t = 0; t < num_mt; ++t) mt_rec[t] = make_uint2(mt_row_off[t], mt_rec_y[t]);
    if (!dict_perr.empty()) std::memcpy(stage + o_dpe, dict_perr.data(), dict_perr.size() * sizeof(double));
    if (!prim.empty()) std::memcpy(stage + o_prim, prim.data(), prim.size() * sizeof(double2));
    if (want16) std::memcpy(stage + o_rec16, rec16.data(), rec16.size() * sizeof(uint2));
    // padding: the slack rows behind the last tile, and the (< 16) marker positions past the last active one
    if (!device_pack) {
        if (pd) std::fill(codes + (size_t)total_rows * kMtMarkers, codes + n_codes, pad_off | (pad_off << 16));
        else std::fill(codes + (size_t)total_rows * kMtMarkers * 2, codes + n_codes, pad4);
    }
    for (int64_t m = device_pack ? m_pad : m_active; m < m_pad; ++m) {
        if (in->known_af) kaf_s[m] = 0.0;
        else {
            for (int kk = 0; kk < k; ++kk) ud_s[(size_t)kk * m_pad + m] = 0.0;
            mu_s[m] = 0.0;
        }
        for (int q = 0; q < 4; ++q) cdiag[(size_t)q * m_pad + m] = 0.0;
        const int t = (int)(m / kMtMarkers), lane = (int)(m % kMtMarkers);
        if (pd) {
            uint32_t* row0 = codes + (size_t)mt_row_off[t] * kMtMarkers;
            {   // (a lane without a marker: the padding row in either phase)
                const uint32_t s1 = mt_rec_y[t] & 0xffffu;
                uint16_t* h16 = reinterpret_cast<uint16_t*>(row0);
                for (uint32_t g = 0; g < 2u * mt_rows[t]; ++g)
                    h16[((size_t)(g >> 1) * kMtMarkers + lane) * 2 + (g & 1u)] = (uint16_t)(pad_off + (g >= s1 ? (uint32_t)kPdAltOffset : 0u));
            }
            continue;
        }
        uint32_t* row0 = codes + (size_t)mt_row_off[t] * kMtMarkers * 2;
        for (size_t j = 0; j < (size_t)mt_rows[t] * 2; ++j) row0[((j >> 1) * kMtMarkers + lane) * 2 + (j & 1)] = pad4;
    }
    // ---- pass B (kernel order: markers sorted by run count, i.e. scattered reads of the panel
    // order arrays -- prefetched): run words, panel rows, diagonal terms into the staging slab ----
    if (device_pack) {
        char* const inp = in_stage.p;            // (run lists and constants are there already, or on the device)
        if (device_flatten) {
        } else if (in->known_af) std::memcpy(inp + i_kaf, in->known_af, (size_t)M * sizeof(double));
        else {
            std::memcpy(inp + i_ud, in->ud, (size_t)M * k * sizeof(double));
            std::memcpy(inp + i_mu, in->means, (size_t)M * sizeof(double));
        }
        uint32_t* const s_src = reinterpret_cast<uint32_t*>(inp + i_src);
        uint32_t* const s_eff = reinterpret_cast<uint32_t*>(inp + i_eff);
        int32_t* const s_pidx = reinterpret_cast<int32_t*>(inp + i_pidx);
        parallel_for(m_active, [&](int, int64_t m0, int64_t m1) {
            for (int64_t m = m0; m < m1; ++m) {
                const int i = active[perm[m]];
                s_src[m] = (uint32_t)(in->read_off[i] - read_base);
                s_eff[m] = (uint32_t)eff_all[i];
                s_pidx[m] = i;
            }
        });
    } else
    parallel_for(m_active, [&](int, int64_t m0, int64_t m1) {
    constexpr int64_t kAhead = 12;
    for (int64_t m = m0; m < m1; ++m) {
        if (m + 2 * kAhead < m1) {                    // (its read_off entry first: the run list is found through it)
            const int i2 = active[perm[m + 2 * kAhead]];
            __builtin_prefetch(&in->read_off[i2]);
            __builtin_prefetch(cd_tmp + (size_t)i2 * 4);
            if (in->known_af) __builtin_prefetch(&in->known_af[i2]);
            else {
                __builtin_prefetch(&in->ud[(size_t)i2 * k]);
                __builtin_prefetch(&in->means[i2]);
            }
        }
        if (m + kAhead < m1) {
            const int i1 = active[perm[m + kAhead]];
            __builtin_prefetch(runs + (in->read_off[i1] - read_base));
        }
        const int i = active[perm[m]];
        const uint16_t* src = runs + (in->read_off[i] - read_base);
        const size_t eff = (size_t)eff_all[i];
        // runs in dictionary order: lanes of a wave then tend to hit the same or
        // neighbouring LDS table rows at the same step (bank-friendly)
        const int t = (int)(m / kMtMarkers), lane = (int)(m % kMtMarkers);
        if (pd && pd_sched) {
            // (the tiles' steps are written below, placed by schedule_tile)
        } else if (pd) {
            // the marker's steps (pack_pd_kernel, statement for statement): ref runs, then alt runs, a run of count c as
            // ceil(c / K) row offsets; either phase padded to the tile's rows with the row of ones
            uint32_t* out = codes + (size_t)mt_row_off[t] * kMtMarkers + lane;
            const uint32_t s1 = mt_rec_y[t] & 0xffffu, s2 = mt_rec_y[t] >> 16;
            uint32_t step = 0, cur = 0;
            auto put = [&](uint32_t off) {
                if (step & 1u) out[(size_t)(step >> 1) * kMtMarkers] = cur | (off << 16);
                else cur = off;
                ++step;
            };
            for (uint32_t cls = 0; cls < 2; ++cls) {
                PdWin win{0u, 0u, 0ull};
                auto put_row = [&](uint32_t row) { put(row * (uint32_t)row_bytes + cls * (uint32_t)kPdAltOffset); };
                for (size_t j = 0; j < eff; ++j) {
                    const uint32_t rw = src[j], idx = rw & 0xffu;
                    if ((idx & 1u) != cls) continue;
                    pd_run(dict, win, idx >> 1, rw >> 8, put_row);
                }
                pd_flush(dict, win, put_row);
                const uint32_t end = cls == 0 ? s1 : 2u * ((s2 + 1u) >> 1);
                while (step < end) put(pad_off + cls * (uint32_t)kPdAltOffset);
            }
        } else {
        uint32_t* row0 = codes + (size_t)mt_row_off[t] * kMtMarkers * 2;
        const size_t slots = (size_t)mt_rows[t] * 2;
        size_t j = run_sched ? slots : 0;                  // (scheduled: the tiles' run words are written below)
        for (; j < eff; ++j) {
            const uint32_t rw = src[j];
            row0[((j >> 1) * kMtMarkers + lane) * 2 + (j & 1)] = row_of_idx[rw & 0xffu] | hi_of_count[rw >> 8];
        }
        for (; j < slots; ++j) row0[((j >> 1) * kMtMarkers + lane) * 2 + (j & 1)] = pad4;
        }
        if (in->known_af) {
            kaf_s[m] = in->known_af[i];
        } else {
            for (int kk = 0; kk < k; ++kk) ud_s[(size_t)kk * m_pad + m] = in->ud[(size_t)i * k + kk];
            mu_s[m] = in->means[i];
        }
        const double* cd = cd_tmp + (size_t)i * 4;
        cdiag[m] = pd ? pother[i] : cd[0];
        cdiag[m_pad + m] = cd[1];
        cdiag[2 * m_pad + m] = cd[2];
        cdiag[3 * m_pad + m] = cd[3];
    }
    });

    if (!device_pack && pd && pd_sched)
        parallel_for(num_mt, [&](int, int64_t t0, int64_t t1) {
            // pack_pd_sched_kernel's work, tile after tile: a phase's steps as row indices, placed by schedule_tile
            TileSched S;
            struct Ident { int operator[](uint32_t i) const { return (int)i; } } ident;
            std::vector<uint16_t> lst[kMtMarkers];
            for (int64_t t = t0; t < t1; ++t) {
                uint32_t first_step = 0;
                const uint32_t s1 = mt_rec_y[t] & 0xffffu, s2 = mt_rec_y[t] >> 16;
                uint16_t* const out16 = reinterpret_cast<uint16_t*>(codes + (size_t)mt_row_off[t] * kMtMarkers);
                auto half = [&](int l, uint32_t g, uint32_t off) {          // step g of lane l (pack_pd_sched_kernel: put_step)
                    out16[((size_t)(g >> 1) * kMtMarkers + l) * 2 + (g & 1u)] = (uint16_t)off;
                };
                for (uint32_t cls = 0; cls < 2; ++cls) {
                    const int steps = (int)(cls == 0 ? s1 : s2 - s1);
                    uint32_t eff16[kMtMarkers];
                    for (int l = 0; l < kMtMarkers; ++l) {
                        lst[l].clear();
                        const int64_t m = t * kMtMarkers + l;
                        if (m < m_active) {
                            const int i = active[perm[m]];
                            const uint16_t* src = runs + (in->read_off[i] - read_base);
                            PdWin win{0u, 0u, 0ull};
                            auto push_row = [&](uint32_t row) { lst[l].push_back((uint16_t)row); };
                            for (int32_t j = 0; j < eff_all[i]; ++j) {
                                const uint32_t rw = src[j], idx = rw & 0xffu;
                                if ((idx & 1u) != cls) continue;
                                pd_run(dict, win, idx >> 1, rw >> 8, push_row);
                            }
                            pd_flush(dict, win, push_row);
                        }
                        eff16[l] = (uint32_t)lst[l].size();
                    }
                    schedule_tile(S, eff16, steps, num_code, ident,
                                  [&](int l, int j) -> uint32_t { return lst[l][j]; },
                                  [&](int l, int c, uint32_t rw) { half(l, first_step + (uint32_t)c, rw * (uint32_t)row_bytes + cls * (uint32_t)kPdAltOffset); },
                                  [&](int l, int c) { half(l, first_step + (uint32_t)c, pad_off + cls * (uint32_t)kPdAltOffset); });
                    first_step += (uint32_t)steps;
                }
                if (s2 & 1u)
                    for (int l = 0; l < kMtMarkers; ++l) half(l, s2, pad_off + (uint32_t)kPdAltOffset);
            }
        });
    if (!device_pack && run_sched)
        parallel_for(num_mt, [&](int, int64_t t0, int64_t t1) {
            TileSched S;
            for (int64_t t = t0; t < t1; ++t) {
                uint32_t eff16[kMtMarkers];
                const uint16_t* src16[kMtMarkers];
                for (int l = 0; l < kMtMarkers; ++l) {
                    const int64_t m = t * kMtMarkers + l;
                    const bool have = m < m_active;
                    const int i = have ? active[perm[m]] : 0;
                    eff16[l] = have ? (uint32_t)eff_all[i] : 0u;
                    src16[l] = have ? runs + (in->read_off[i] - read_base) : runs;
                }
                uint32_t* row0 = codes + (size_t)mt_row_off[t] * kMtMarkers * 2;
                schedule_tile(S, eff16, (int)(2u * mt_rows[t]), num_code, dict_of.data(),
                              [&](int l, int j) -> uint32_t { return src16[l][j]; },
                              [&](int l, int c, uint32_t rw) {
                                  row0[((size_t)(c >> 1) * kMtMarkers + l) * 2 + (c & 1)] = row_of_idx[rw & 0xffu] | hi_of_count[rw >> 8];
                              },
                              [&](int l, int c) { row0[((size_t)(c >> 1) * kMtMarkers + l) * 2 + (c & 1)] = pad4; });
            }
        });

    const auto t_flat = tnow();
    if (dry && t_digest_out) {
        // digest of everything the flatten produced (the defined parts of the staging slab, not its alignment gaps)
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&](const void* ptr, size_t bytes) {
            const unsigned char* q = static_cast<const unsigned char*>(ptr);
            for (size_t i = 0; i < bytes; ++i) hsh = (hsh ^ q[i]) * 1099511628211ull;
        };
        // (Tunables::digest_multiset, a test aid: the run words enter as a SUM of word hashes per micro-tile -- the same for
        // any order of a tile's words: what schedule_tile may change and nothing else)
        if (tn.digest_multiset && !pd) {
            for (int t = 0; t < num_mt; ++t) {
                uint64_t sum = 0;
                const uint32_t* w = codes + (size_t)mt_row_off[t] * kMtMarkers * 2;
                for (size_t j = 0; j < (size_t)mt_rows[t] * kMtMarkers * 2; ++j) {
                    // (per lane: the lane index enters, so words may move between steps but not between markers)
                    const uint64_t x = ((uint64_t)((j >> 1) % kMtMarkers) << 32 | w[j]) * 0x9E3779B97F4A7C15ull;
                    sum += x ^ (x >> 29);
                }
                mix(&sum, sizeof(sum));
            }
        } else {
            mix(codes, n_codes * sizeof(uint32_t));
        }
        mix(mt_rec, (size_t)num_mt * sizeof(uint2));
        if (in->known_af) mix(kaf_s, (size_t)m_pad * sizeof(double));
        else { mix(ud_s, (size_t)k * m_pad * sizeof(double)); mix(mu_s, (size_t)m_pad * sizeof(double)); }
        mix(cdiag, (size_t)4 * m_pad * sizeof(double));
        mix(stage + o_dpe, dict_perr.size() * sizeof(double));
        mix(stage + o_prim, prim.size() * sizeof(double2));
        const int64_t counts[4] = {num_read, num_other, (int64_t)num_code, m_active};
        mix(counts, sizeof(counts));
        *t_digest_out = hsh;
    }
    if (dry) {
        if (timing)
            std::fprintf(stderr, "flatten (dry): %.1f ms (classify %.1f, dictionary+sort %.1f, pack %.1f; %d threads)\n",
                         tms(t_start, t_flat), tms(t_start, t_pass1), tms(t_pass1, t_sort), tms(t_sort, t_flat), nthr);
        if (timing && pd) {
            // what the read loops will see (a diagnostic of the layout): steps per marker, padding included, and LDS passes
            // per step -- the 16 markers of a tile read one table row each; different rows whose indices agree mod 16 start
            // in the same bank group and are served one after the other
            const uint16_t* h16 = reinterpret_cast<const uint16_t*>(codes);
            uint64_t steps = 0, passes = 0, pads = 0;
            for (int t = 0; t < num_mt; ++t) {
                const uint32_t s1 = mt_rec_y[t] & 0xffffu;
                for (uint32_t g = 0; g < 2u * mt_rows[t]; ++g) {
                    int nrow_in[16][4], nin[16];
                    std::fill(nin, nin + 16, 0);
                    int worst = 1;
                    for (int l = 0; l < kMtMarkers; ++l) {
                        uint32_t off = h16[(((size_t)mt_row_off[t] + (g >> 1)) * kMtMarkers + l) * 2 + (g & 1u)];
                        if (g >= s1) off -= (uint32_t)kPdAltOffset;
                        const int row = (int)(off / (uint32_t)row_bytes), r = row & 15;
                        if (row == num_code) ++pads;
                        bool seen = false;
                        for (int q = 0; q < nin[r]; ++q) seen = seen || nrow_in[r][q] == row;
                        if (!seen && nin[r] < 4) nrow_in[r][nin[r]++] = row;
                        worst = std::max(worst, nin[r]);
                    }
                    passes += (uint64_t)worst;
                    ++steps;
                }
            }
            std::fprintf(stderr, "  probability domain: %d table rows, %.2f steps per marker (%.2f of them padding), %.3f LDS passes per step\n",
                         num_code, 16.0 * (double)steps / (double)std::max<int64_t>(1, m_active),
                         (double)pads / (double)std::max<int64_t>(1, m_active), steps ? (double)passes / (double)steps : 0.0);
        }
        return VB2_OK;
    }
    c->d_slab = slab_cache().take(slab_cache().dev, dev_total, dev, &c->d_slab_bytes);
    if (!c->d_slab) {
        VB2_HIP(hipMalloc((void**)&c->d_slab, dev_total));
        c->d_slab_bytes = dev_total;
    }
    char* const dbase = static_cast<char*>(c->d_slab);
    // the data arrays in ONE asynchronous copy; partial sums, ticket, relay, stamps and schedule space start as zeros
    if (device_pack) {
        char* const din = static_cast<char*>(d_in.p);
        if (device_flatten)     // (the reads, the panel rows and pass A's results are there: the three words per sorted marker follow)
            VB2_HIP(hipMemcpyAsync(din + up1_end, in_stage.p + up1_end, up2_end - up1_end, hipMemcpyHostToDevice, c->stream));
        else
            VB2_HIP(hipMemcpyAsync(din, in_stage.p, in_total, hipMemcpyHostToDevice, c->stream));
        // the small tables of the data block, each to its place
        VB2_HIP(hipMemcpyAsync(dbase + o_rec, stage + o_rec, (size_t)num_mt * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
        if (!dict_perr.empty())
            VB2_HIP(hipMemcpyAsync(dbase + o_dpe, stage + o_dpe, dict_perr.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (!prim.empty())
            VB2_HIP(hipMemcpyAsync(dbase + o_prim, stage + o_prim, prim.size() * sizeof(double2), hipMemcpyHostToDevice, c->stream));
        if (want16)
            VB2_HIP(hipMemcpyAsync(dbase + o_rec16, stage + o_rec16, rec16.size() * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
        if (pd) {
            PackPdArgs pp;
            std::memset(&pp, 0, sizeof(pp));
            pp.runs = reinterpret_cast<const uint16_t*>(din + i_runs);
            pp.src_off = reinterpret_cast<const uint32_t*>(din + i_src);
            pp.nrun = reinterpret_cast<const uint32_t*>(din + i_eff);
            pp.pidx = reinterpret_cast<const int32_t*>(din + i_pidx);
            pp.cd = reinterpret_cast<const double*>(din + i_cd);
            pp.pother = reinterpret_cast<const double*>(din + i_pother);
            pp.ud = in->known_af ? nullptr : reinterpret_cast<const double*>(din + i_ud);
            pp.mu = in->known_af ? nullptr : reinterpret_cast<const double*>(din + i_mu);
            pp.kaf = in->known_af ? reinterpret_cast<const double*>(din + i_kaf) : nullptr;
            pp.mt_rec = reinterpret_cast<const uint2*>(dbase + o_rec);
            pp.codes = reinterpret_cast<uint32_t*>(dbase + o_codes);
            pp.ud_s = reinterpret_cast<double*>(dbase + o_ud);
            pp.mu_s = reinterpret_cast<double*>(dbase + o_mu);
            pp.kaf_s = in->known_af ? reinterpret_cast<double*>(dbase + o_kaf) : nullptr;
            pp.cdiag = reinterpret_cast<double*>(dbase + o_cd);
            pp.m_active = m_active;
            pp.m_pad = m_pad;
            pp.k = k;
            pp.num_mt = num_mt;
            pp.total_rows = (uint32_t)total_rows;
            pp.slack_rows = (uint32_t)kCodeSlackRows;
            pp.pad_off = pad_off;
            pp.dict = dict;
            pp.sched = pd_sched ? 1 : 0;
            pp.num_code = num_code;
            pp.row_bytes = row_bytes;
            VB2_HIP(launch_pack_pd(pp, c->stream));
        } else {
        PackArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.runs = reinterpret_cast<const uint16_t*>(din + i_runs);
        pa.src_off = reinterpret_cast<const uint32_t*>(din + i_src);
        pa.eff = reinterpret_cast<const uint32_t*>(din + i_eff);
        pa.pidx = reinterpret_cast<const int32_t*>(din + i_pidx);
        pa.cd = reinterpret_cast<const double*>(din + i_cd);
        pa.ud = in->known_af ? nullptr : reinterpret_cast<const double*>(din + i_ud);
        pa.mu = in->known_af ? nullptr : reinterpret_cast<const double*>(din + i_mu);
        pa.kaf = in->known_af ? reinterpret_cast<const double*>(din + i_kaf) : nullptr;
        pa.mt_rec = reinterpret_cast<const uint2*>(dbase + o_rec);
        pa.codes = reinterpret_cast<uint2*>(dbase + o_codes);
        pa.ud_s = reinterpret_cast<double*>(dbase + o_ud);
        pa.mu_s = reinterpret_cast<double*>(dbase + o_mu);
        pa.kaf_s = in->known_af ? reinterpret_cast<double*>(dbase + o_kaf) : nullptr;
        pa.cdiag = reinterpret_cast<double*>(dbase + o_cd);
        pa.m_active = m_active;
        pa.m_pad = m_pad;
        pa.k = k;
        pa.num_mt = num_mt;
        pa.total_rows = (uint32_t)total_rows;
        pa.slack_rows = (uint32_t)kCodeSlackRows;
        pa.pad4 = pad4;
        for (int idx = 0; idx < kMaxCode; ++idx) pa.row_of_idx[idx] = row_of_idx[idx];
        for (int n = 0; n <= kMaxRunCount; ++n) pa.hi_of_count[n] = hi_of_count[n];
        pa.sched = run_sched ? 1 : 0;
        pa.num_code = num_code;
        for (int idx = 0; idx < kMaxCode; ++idx) pa.dict_of[idx] = dict_of[idx];
        VB2_HIP(launch_pack_layout(pa, c->stream));
        }
    } else {
        VB2_HIP(hipMemcpyAsync(dbase, stage, data_bytes, hipMemcpyHostToDevice, c->stream));
    }
    VB2_HIP(hipMemsetAsync(dbase + o_part, 0, dev_total - o_part, c->stream));
    if (!in->known_af) {
        L.ud = reinterpret_cast<const double*>(dbase + o_ud);
        L.mu = reinterpret_cast<const double*>(dbase + o_mu);
    } else {
        L.known_af = reinterpret_cast<const double*>(dbase + o_kaf);
    }
    L.codes = reinterpret_cast<const uint2*>(dbase + o_codes);
    L.mt_rec = reinterpret_cast<const uint2*>(dbase + o_rec);
    L.ediag = reinterpret_cast<const double*>(dbase + o_cd);
    L.dict_perr = reinterpret_cast<const double*>(dbase + o_dpe);
    L.prim = reinterpret_cast<const double2*>(dbase + o_prim);
    c->d_partials = reinterpret_cast<double*>(dbase + o_part);
    c->d_ticket = reinterpret_cast<unsigned int*>(dbase + o_ticket);
    c->d_relay = reinterpret_cast<unsigned long long*>(dbase + o_relay);
    if (want_stamps) {
        c->d_stamps = reinterpret_cast<unsigned long long*>(dbase + o_stamps);
        L.stamps = c->d_stamps;
    }
    for (int slot = 0; slot < 9; ++slot) {
        c->sched_[slot].d_base = dbase + o_sched[slot];
        c->sched_[slot].bytes = sched_bytes[slot];
    }
    c->h_mt_rows.assign(mt_rows.begin(), mt_rows.end());
    if (pd) c->h_mt_rec_y.assign(mt_rec_y.begin(), mt_rec_y.end());
    c->dbg_regions = {{o_codes, n_codes * sizeof(uint32_t)}, {o_rec, (size_t)num_mt * sizeof(uint2)}};
    if (in->known_af) c->dbg_regions.push_back({o_kaf, (size_t)m_pad * sizeof(double)});
    else {
        c->dbg_regions.push_back({o_ud, (size_t)k * m_pad * sizeof(double)});
        c->dbg_regions.push_back({o_mu, (size_t)m_pad * sizeof(double)});
    }
    c->dbg_regions.push_back({o_cd, (size_t)4 * m_pad * sizeof(double)});
    c->dbg_regions.push_back({o_dpe, dict_perr.size() * sizeof(double)});
    c->dbg_regions.push_back({o_prim, prim.size() * sizeof(double2)});
    c->dbg_counts[0] = num_read; c->dbg_counts[1] = num_other; c->dbg_counts[2] = (int64_t)num_code; c->dbg_counts[3] = m_active;
    c->sched_enabled = tn.sched != 0;
    c->device_bytes = (int64_t)dev_total;
    L.num_prim = (int32_t)prim.size();
    L.num_pair = num_pair;
    L.num_pair2 = pd ? pd_prod2 : 0;
    L.reserved1 = 0;
    L.pd = pd ? 1 : 0;
    c->num_code_seen = num_code_seen;
    L.num_code = num_code;
    L.row_bytes = row_bytes;
    L.num_mt = num_mt;
    L.num_cu = num_cu;
    L.dyn_limit = tn.dyn_tiles;
    L.stagger = tn.stagger;
    L.num_pc = k;
    L.num_active = m_active;
    L.m_pad = m_pad;
    // bytes one cohort step reads of this sample: run lists + tile records + panel rows + diagonal terms
    const int64_t panel_bytes = (int64_t)(in->known_af ? 1 : k + 1) * m_pad * 8 + 4 * m_pad * 8 + (int64_t)num_mt * 8;
    c->cohort_bytes = (int64_t)total_rows * kMtMarkers * (pd ? 4 : 8) + panel_bytes;
    if (want16) {
        L.mt_rec16 = reinterpret_cast<const uint2*>(dbase + o_rec16);
        uint2* d16 = reinterpret_cast<uint2*>(dbase + o_codes16);
        VB2_HIP(launch_pack_codes16(L, d16, L.mt_rec16, (uint32_t)total_rows16, c->stream));
        L.codes16 = d16;
        c->cohort_bytes = (int64_t)total_rows16 * kMtMarkers * 8 + panel_bytes;
    }

    c->num_read = num_read;
    c->num_read_other = num_other;
    c->algorithmic_bytes = 2 * num_read + m_active * (8 * (int64_t)k + 12);

    // Host <-> device hand-off of the (tiny) parameter and result vectors goes through
    // pinned, device-mapped host memory that the kernels access directly: no copy
    // commands on the evaluation path.
    // (one pinned allocation for all of them, for the same reason as the device slab)
    const size_t pt_bytes = sizeof(double) * (size_t)kStagePoints * (2 * k + 1);
    size_t pin_total = 0;
    auto pcarve = [&](size_t bytes) {
        const size_t off = (pin_total + 127) & ~(size_t)127;
        pin_total = off + bytes;
        return off;
    };
    const size_t p_points = pcarve(pt_bytes);
    const size_t p_out = pcarve(sizeof(double) * kStagePoints);
    const size_t p_done = pcarve(sizeof(unsigned long long) * 8);        // [0] sequence number (+ spare words)
    const size_t p_cmd = pcarve(sizeof(unsigned long long) * relay_words);
    const size_t p_state = pcarve(sizeof(unsigned int));
    const size_t p_result = pcarve(sizeof(double) * (8 + kDeviceSimplexMaxDim + 2 * VB2_MAX_PC));
    c->h_slab = slab_cache().take(slab_cache().pin, pin_total, dev, &c->h_slab_bytes);
    if (!c->h_slab) {
        VB2_HIP(hipHostMalloc((void**)&c->h_slab, pin_total, hipHostMallocMapped));
        c->h_slab_bytes = pin_total;
    }
    std::memset(c->h_slab, 0, pin_total);
    char* hbase = static_cast<char*>(c->h_slab);
    char* hdev = nullptr;
    VB2_HIP(hipHostGetDevicePointer((void**)&hdev, c->h_slab, 0));
    c->h_points = reinterpret_cast<double*>(hbase + p_points);
    c->d_points = reinterpret_cast<double*>(hdev + p_points);
    c->h_out = reinterpret_cast<double*>(hbase + p_out);
    c->d_out = reinterpret_cast<double*>(hdev + p_out);
    c->h_done = reinterpret_cast<unsigned long long*>(hbase + p_done);
    c->d_done = reinterpret_cast<unsigned long long*>(hdev + p_done);
    c->h_cmd = reinterpret_cast<unsigned long long*>(hbase + p_cmd);
    c->d_cmd = reinterpret_cast<unsigned long long*>(hdev + p_cmd);
    c->h_state = reinterpret_cast<unsigned int*>(hbase + p_state);
    c->d_state = reinterpret_cast<unsigned int*>(hdev + p_state);
    c->h_result = reinterpret_cast<double*>(hbase + p_result);
    c->d_result = reinterpret_cast<double*>(hdev + p_result);
    c->device_simplex_enabled = tn.device_simplex != 0 && !(opt && (opt->flags & VB2_OPT_HOST_SEARCH));
    c->spin_wait = tn.spin_wait != 0;
    c->resident_enabled = tn.resident != 0 && !(opt && (opt->flags & VB2_OPT_LAUNCH_PER_STEP));
    c->plain_launch = tn.coop == 0 || (opt && (opt->flags & VB2_OPT_PLAIN_LAUNCH)) || profiler_attached();
    c->dbg_timing = timing;
    // the upload has left the staging slab (which goes back to the cache now): wait for THIS
    // context's stream only -- other contexts' streams and the null stream are not touched
    VB2_HIP(hipStreamSynchronize(c->stream));
    if (timing) {
        std::fprintf(stderr, "vb2_ctx_create: flatten %.2f ms (%s %.2f, dictionary+sort %.2f, pack %.2f; "
                     "%d threads), device alloc+upload%s %.2f ms\n",
                     tms(t_start, t_flat), device_flatten ? "stage" : "classify",
                     device_flatten ? tms(t_start, t_staged) : tms(t_start, t_pass1), tms(t_pass1, t_sort), tms(t_sort, t_flat), nthr,
                     device_pack ? "+pack kernels" : "", tms(t_flat, tnow()));
        if (device_flatten) std::fprintf(stderr, "  upload + classify_kernel + read-back: %.2f ms\n", tms(t_staged, t_pass1));
    }
    *out = c.release();
    return VB2_OK;
}

int Context::eval_device(int num_point, const double* d_pts, double* d_llk, hipStream_t s,
                         unsigned long long* done_flag, unsigned long long done_seq,
                         const double* h_pts, int reduce_override)
{
    if (num_point <= 0) return VB2_OK;
    VB2_HIP(hipSetDevice(device));
    if (!s) s = stream;
    if (L.num_mt == 0) {
        VB2_HIP(launch_fill_zero(d_llk, num_point, s));
        return VB2_OK;
    }
    VB2_HIP(launch_llk_eval(L, num_point, d_pts, h_pts, d_partials, d_llk, d_ticket, done_flag, done_seq,
                            &done_seq_, s, reduce_override, this));
    return VB2_OK;
}

Schedule Context::get(int mode, int ngrp, int grid, int block_waves)
{
    SchedSlot& sl = sched_[sched_slot(mode, ngrp)];
    if (sl.tried || !sched_enabled) return sl.s;
    sl.tried = true;                                     // a failure below leaves the snake deal in place
    std::vector<uint32_t> off;
    std::vector<uint16_t> item;
    const int tpu = mode == 3 ? 2 : mode == 4 ? 4 : 1;
    if (!build_schedule(h_mt_rows.data(), L.num_mt, grid, block_waves, tpu, ngrp, &off, &item, L.pd ? 1 : 0)) return sl.s;
    const size_t off_bytes = (off.size() * sizeof(uint32_t) + 15) / 16 * 16;
    if (off_bytes + item.size() * sizeof(uint16_t) > sl.bytes) return sl.s;
    if (hipMemcpyAsync(sl.d_base, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(sl.d_base + off_bytes, item.data(), item.size() * sizeof(uint16_t), hipMemcpyHostToDevice,
                       stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {   // (pageable sources: the copies are staged before this returns)
        (void)hipGetLastError();
        return sl.s;
    }
    sl.s.off = reinterpret_cast<const uint32_t*>(sl.d_base);
    sl.s.item = reinterpret_cast<const uint16_t*>(sl.d_base + off_bytes);
    return sl.s;
}

int Context::cohort_schedules(int bps, int block_waves, Schedule out[4], bool full)
{
    for (int sh = 0; sh < 4; ++sh) out[sh] = Schedule{nullptr, nullptr};
    if (!sched_enabled || L.num_mt == 0 || eval_takes_the_queue(L, bps, block_waves, 1)) return VB2_OK;
    std::lock_guard<std::mutex> lk(cohort_mu_);
    for (const CohortSched& e : cohort_sched_)
        if (e.bps == bps && e.block_waves == block_waves && (e.full || !full)) {
            for (int sh = full ? 0 : 2; sh < 4; ++sh) out[sh] = e.s[sh];      // (!full: never the other two, whoever built them)
            return VB2_OK;
        }
    // micro-tiles a wave takes per item: two for <= 4 points (when paired), one for 8 points, four for one or two points.
    // Shapes with the same number share ONE schedule; !full: only the one- and two-point shapes (a search's steps but for
    // its first and its shrinks, which then take the in-kernel snake deal).
    const int tpu[4] = {2, 1, 4, 4};
    std::vector<char> blob;
    size_t where[4] = {0, 0, 0, 0};
    bool built[4] = {false, false, false, false};
    const size_t ob = (((size_t)bps * block_waves + 1) * sizeof(uint32_t) + 15) / 16 * 16;
    CohortSched e;
    e.bps = bps;
    e.block_waves = block_waves;
    e.full = full;
    for (int sh = 0; sh < 4; ++sh) {
        if (!full && sh < 2) continue;
        int same = -1;
        for (int prev = 0; prev < sh; ++prev)
            if (built[prev] && tpu[prev] == tpu[sh]) same = prev;
        if (same >= 0) {
            where[sh] = where[same];
            built[sh] = true;
            continue;
        }
        std::vector<uint32_t> off;
        std::vector<uint16_t> item;
        if (!build_schedule(h_mt_rows.data(), L.num_mt, bps, block_waves, tpu[sh], 1, &off, &item, L.pd ? 1 : 0)) {
            e.full = true;                                  // (nothing more to be had for this geometry)
            cohort_sched_.push_back(e);                     // (every shape: the snake deal)
            return VB2_OK;
        }
        where[sh] = blob.size();
        built[sh] = true;
        blob.resize(blob.size() + ob + (item.size() * sizeof(uint16_t) + 15) / 16 * 16);
        std::memcpy(blob.data() + where[sh], off.data(), off.size() * sizeof(uint32_t));
        std::memcpy(blob.data() + where[sh] + ob, item.data(), item.size() * sizeof(uint16_t));
    }
    VB2_HIP(hipSetDevice(device));
    e.d_mem = slab_cache().take(slab_cache().dev, blob.size(), device, &e.bytes);
    if (!e.d_mem) {
        VB2_HIP(hipMalloc(&e.d_mem, blob.size()));
        e.bytes = blob.size();
    }
    if (hipMemcpyAsync(e.d_mem, blob.data(), blob.size(), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {        // (a pageable source: staged before this returns)
        (void)hipGetLastError();
        if (!slab_cache().give(slab_cache().dev, e.d_mem, e.bytes, device)) (void)hipFree(e.d_mem);
        set_error("cohort schedule upload failed");
        return VB2_ERR_HIP;
    }
    for (int sh = 0; sh < 4; ++sh) {
        if (!built[sh]) continue;
        const char* base = static_cast<const char*>(e.d_mem) + where[sh];
        e.s[sh] = Schedule{reinterpret_cast<const uint32_t*>(base), reinterpret_cast<const uint16_t*>(base + ob)};
        out[sh] = e.s[sh];
    }
    cohort_sched_.push_back(e);
    return VB2_OK;
}

// One resident search per device at a time (all its workgroups must be on the CUs together).
static std::atomic<int> g_resident_busy[64];

bool Context::resident_begin()
{
    if (!resident_enabled || resident_active || L.num_mt == 0 || !spin_wait || device < 0 || device >= 64)
        return false;
    int expected = 0;
    if (!g_resident_busy[device].compare_exchange_strong(expected, 1)) return false;
    bool ok = hipSetDevice(device) == hipSuccess;
    const size_t words = (size_t)resident_words(num_pc);
    if (ok) {
        std::memset(h_cmd, 0, sizeof(unsigned long long) * words);
        __atomic_store_n(h_state, 0u, __ATOMIC_RELEASE);
        ok = hipMemsetAsync(d_relay, 0, sizeof(unsigned long long) * (size_t)resident_relay_words(num_pc), stream) == hipSuccess;
    }
    if (ok) {
        ResidentArgs ra;
        ra.h_cmd = d_cmd;
        ra.relay = d_relay;
        ra.relay_reps = 8;                                // (1 .. kRelayReps measured: 8 copies, 32 workgroups polling each)
        ra.extra_ctl = 0;
        ra.spec = d_relay + resident_words(num_pc);
        ra.h_out = d_out;
        ra.h_done = d_done;
        ra.h_state = d_state;
        ra.first_seq = done_seq_ + 1;
        ra.timeout_ticks = 100000000ull;                  // 1 s without a command: give up
        ra.h_result = d_result;
        ra.epoch = ++resident_epoch_;
        // room for the on-device simplex of every model of this context: dimension <= 2k + 1
        ra.state_nmax = (device_simplex_enabled && 2 * num_pc + 1 <= kDeviceSimplexMaxDim) ? 2 * num_pc + 1 : 0;
        ra.state_off = 0;
        {
            const LaunchGeom gm = launch_geom(L, 1);
            ra.sched_multi = get(3, 1, gm.grid, gm.block_waves);
            ra.sched_single = get(4, 1, gm.grid, gm.block_waves);
            // the workgroups' own run lists in LDS for the whole search, if they fit (Tunables::lds_cache 0: the A/B knob)
            ra.cache_tiles = (int32_t)owned_most(L.pd ? 1 : 0, (uint32_t)L.num_mt, (uint32_t)gm.grid);
            ra.cache_rows = tunables().lds_cache ? (int32_t)resident_cache_rows(h_mt_rows.data(), L.num_mt, gm.grid, L.pd ? 1 : 0) : 0;
        }
        bool coop = !plain_launch;
        ok = launch_llk_resident(L, &ra, d_partials, d_ticket, stream, &coop) == hipSuccess;
        resident_cooperative = ok && coop;
        if (!ok) (void)hipGetLastError();
        resident_nmax = ok ? ra.state_nmax : 0;
    }
    if (!ok) {
        g_resident_busy[device].store(0);
        return false;
    }
    resident_active = true;
    return true;
}

// Posts rows (or the exit command, n = 0) under sequence number seq.
static void resident_post(unsigned long long* cmd, int words, unsigned long long seq, int n, int stride,
                          const double* rows)
{
    unsigned long long x = 0;
    if (n > 0) std::memcpy(cmd + 2, rows, sizeof(double) * (size_t)n * stride);
    cmd[1] = (unsigned long long)n;
    for (int w = 1; w < words - 1; ++w) x ^= word_hash(cmd[w], (unsigned)w);
    cmd[words - 1] = x ^ resident_mix(seq);
    __atomic_store_n(&cmd[0], seq, __ATOMIC_RELEASE);
}

void Context::resident_end()
{
    if (!resident_active) return;
    (void)hipSetDevice(device);
    resident_post(h_cmd, resident_words(num_pc), ++done_seq_, 0, 2 * num_pc + 1, nullptr);
    (void)hipStreamSynchronize(stream);                    // the kernel leaves on the exit command
    if (dbg_timing && dbg_cmds > 0) {
        std::fprintf(stderr, "resident search: %lld commands, host between result and next post %.2f us avg, "
                     "post -> result seen %.2f us avg\n", (long long)dbg_cmds,
                     1e-3 * dbg_host_ns / dbg_cmds, 1e-3 * dbg_wait_ns / dbg_cmds);
        dbg_cmds = 0; dbg_host_ns = dbg_wait_ns = 0; dbg_have_prev = false;
    }
    resident_active = false;
    g_resident_busy[device].store(0);
}

// The results of a step come back through mapped host memory as relaxed system-scope stores followed -- after the
// storing lanes have their acknowledgements -- by the relaxed store of the sequence number the host spins on (round 4:
// the release fences this replaced were half of an empty step's 20 us).  The memory model does not promise that the
// acknowledged stores are visible to the HOST before the flag is, so the host does not rely on it: the
// result words are set to NaN before every step, and a NaN found behind the flag is first taken for a store still on its
// way (re-read for a few microseconds) and only then for the kernels' own "a workgroup never reported" marker.
bool settle_results(const double* out, int n)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool any_nan = false;
        for (int b = 0; b < n; ++b) any_nan |= std::isnan(reinterpret_cast<const volatile double*>(out)[b]);
        if (!any_nan) return true;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) return false;
        __builtin_ia32_pause();
    }
}

void Context::resident_submit(int n, const double* rows)
{
    for (int b = 0; b < n; ++b) h_out[b] = std::numeric_limits<double>::quiet_NaN();
    resident_post(h_cmd, resident_words(num_pc), ++done_seq_, n, 2 * num_pc + 1, rows);
    dbg_t_post = std::chrono::steady_clock::now();
    if (dbg_timing && dbg_have_prev)
        dbg_host_ns += std::chrono::duration<double, std::nano>(dbg_t_post - dbg_prev_seen).count();
}

bool Context::resident_collect(int n, double* out)
{
    const unsigned long long seq = done_seq_;
    const int stride = 2 * num_pc + 1;
    bool seen = false;
    const auto t0 = dbg_t_post;
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
        if ((spins & 0x3ff) == 0x3ff) {
            if (__atomic_load_n(h_state, __ATOMIC_ACQUIRE) == 3u) break;       // kernel gave up
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) break;
        }
        __builtin_ia32_pause();
    }
    // A NaN can only be the kernel's own "a workgroup never reported" marker (the partial-sum
    // hand-off gave up after a quarter of a second): part of the grid is not on the CUs.  Same treatment as no
    // answer at all.
    if (seen && !settle_results(h_out, n)) seen = false;
    if (!seen) {
        // Give up on the mode: tell the kernel to leave (it may already have, on its idle
        // limit), wait for it, and let the caller redo the batch with plain launches.
        (void)hipSetDevice(device);
        resident_post(h_cmd, resident_words(num_pc), ++done_seq_, 0, stride, nullptr);
        (void)hipStreamSynchronize(stream);
        resident_active = false;
        resident_enabled = false;
        g_resident_busy[device].store(0);
        (void)hipMemsetAsync(d_ticket, 0, sizeof(unsigned int) * kTicketWords, stream);
        return false;
    }
    std::memcpy(out, h_out, sizeof(double) * n);
    ++resident_evals;
    if (dbg_timing) {
        dbg_prev_seen = std::chrono::steady_clock::now();
        dbg_wait_ns += std::chrono::duration<double, std::nano>(dbg_prev_seen - t0).count();
        dbg_have_prev = true;
        ++dbg_cmds;
    }
    return true;
}

int Context::device_minimize(MinimizeRequest* req)
{
    const int k = num_pc, n = req->dim;
    if (!resident_active || n < 1 || n > resident_nmax || !req->start) return VB2_ERR_INVALID;
    const int words = resident_words(k);
    // trace staging: rows [alpha, llk, pc1(k), pc2(k)] in mapped host memory (reserve_trace, before
    // the resident kernel went up: no allocation call while it runs), scattered into the caller's
    // arrays afterwards
    double* h_trace = h_trace_stage;
    double* d_trace = d_trace_stage;
    long long cap = 0;
    if (req->trace && req->trace->capacity > req->trace->count && h_trace)
        cap = std::min<long long>(req->trace->capacity - req->trace->count, trace_stage_rows);
    if (req->trace && cap == 0 && req->trace->capacity > req->trace->count) return VB2_ERR_INVALID;   // no staging: host search
    // ---- the request (resident_kernel.inc: word offsets of the MINIMIZE command) ----
    unsigned long long* cmd = h_cmd;
    auto put_d = [&](int w, double v) { std::memcpy(&cmd[w], &v, sizeof(v)); };
    for (int w = 1; w < words; ++w) cmd[w] = 0;
    cmd[1] = kResidentMinimize;
    cmd[2] = (unsigned long long)n | ((unsigned long long)req->kind << 8) | ((unsigned long long)k << 16);
    cmd[3] = (unsigned long long)req->cycle_max;
    put_d(4, req->ftol);
    put_d(5, req->llk1);
    put_d(6, req->fix_alpha);
    put_d(7, req->g_alpha);
    cmd[8] = (unsigned long long)d_trace;
    cmd[9] = (unsigned long long)cap;
    for (int j = 0; j < n; ++j) put_d(10 + j, req->start[j]);
    for (int j = 0; j < k; ++j) {
        put_d(10 + n + j, req->fix_pc[j]);
        put_d(10 + n + k + j, req->fix_pc2[j]);
        put_d(10 + n + 2 * k + j, req->g_pc[j]);
        put_d(10 + n + 3 * k + j, req->g_pc2[j]);
    }
    const unsigned long long seq = ++done_seq_;
    unsigned long long x = 0;
    for (int w = 1; w < words - 1; ++w) x ^= word_hash(cmd[w], (unsigned)w);
    cmd[words - 1] = x ^ resident_mix(seq);
    __atomic_store_n(&cmd[0], seq, __ATOMIC_RELEASE);
    // ---- one wait for the whole search (cycleMax evaluations of ~15 us each at most) ----
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
        if ((spins & 0x3ff) == 0x3ff) {
            if (__atomic_load_n(h_state, __ATOMIC_ACQUIRE) == 3u) break;       // kernel gave up
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;
        }
        __builtin_ia32_pause();
    }
    int rc = VB2_OK;
    if (!seen || std::isnan(h_result[1]) || std::isnan(h_result[5])) {
        // no answer (or a "workgroup never reported" NaN inside): leave the mode, the caller's host
        // optimiser redoes the search with plain launches
        (void)hipSetDevice(device);
        resident_post(h_cmd, words, ++done_seq_, 0, 2 * k + 1, nullptr);
        (void)hipStreamSynchronize(stream);
        resident_active = false;
        resident_enabled = false;
        g_resident_busy[device].store(0);
        (void)hipMemsetAsync(d_ticket, 0, sizeof(unsigned int) * kTicketWords, stream);
        rc = VB2_ERR_HIP;
    } else {
        req->status = (int)h_result[0];
        req->ret = h_result[1];
        req->cycle_count = (long)h_result[2];
        req->num_eval = (long)h_result[3];
        req->num_point = (long)h_result[4];
        req->out_llk1 = h_result[5];
        req->out_g_alpha = h_result[6];
        const long long ntrace = (long long)h_result[7];
        std::memcpy(req->point, h_result + 8, sizeof(double) * n);
        std::memcpy(req->out_g_pc, h_result + 8 + n, sizeof(double) * k);
        std::memcpy(req->out_g_pc2, h_result + 8 + n + k, sizeof(double) * k);
        if (req->status != 3) {
            ++device_minimizes;
            if (req->trace) {
                vb2_trace* t = req->trace;
                for (long long i = 0; i < ntrace; ++i) {
                    if (i < cap && t->count < t->capacity) {
                        const double* row = h_trace + (size_t)i * (2 * k + 2);
                        const int64_t r = t->count;
                        t->alpha[r] = row[0];
                        t->llk[r] = row[1];
                        std::memcpy(t->pc1 + r * k, row + 2, sizeof(double) * k);
                        std::memcpy(t->pc2 + r * k, row + 2 + k, sizeof(double) * k);
                    }
                    t->count++;
                }
            }
        }
    }
    return rc;
}

int Context::reserve_trace(int64_t rows)
{
    if (rows <= trace_stage_rows) return VB2_OK;
    VB2_HIP(hipSetDevice(device));
    if (h_trace_stage) (void)hipHostFree(h_trace_stage);
    h_trace_stage = d_trace_stage = nullptr;
    trace_stage_rows = 0;
    VB2_HIP(hipHostMalloc((void**)&h_trace_stage, sizeof(double) * (size_t)rows * (2 * num_pc + 2), hipHostMallocMapped));
    VB2_HIP(hipHostGetDevicePointer((void**)&d_trace_stage, h_trace_stage, 0));
    trace_stage_rows = rows;
    return VB2_OK;
}

int Context::eval_host(int num_point, const double* pc1, const double* pc2, const double* alpha,
                       double* llk_out)
{
    if (num_point < 0 || (num_point > 0 && (!pc1 || !pc2 || !alpha || !llk_out))) {
        set_error("vb2_llk_eval_batch: invalid argument");
        return VB2_ERR_INVALID;
    }
    VB2_HIP(hipSetDevice(device));
    const int k = num_pc, stride = 2 * k + 1;
    const double *const pc1_all = pc1, *const pc2_all = pc2, *const alpha_all = alpha;
    double* const llk_all = llk_out;
    int served = 0;
    while (resident_active && served < num_point) {
        // the search kernel is already on the CUs: post <= 4 rows, spin on the sequence number
        const int n = std::min(4, num_point - served);
        double rows[4 * (2 * VB2_MAX_PC + 1)];
        for (int b = 0; b < n; ++b) {
            double* row = rows + (size_t)b * stride;
            std::memcpy(row, pc1 + (size_t)(served + b) * k, sizeof(double) * k);
            std::memcpy(row + k, pc2 + (size_t)(served + b) * k, sizeof(double) * k);
            row[2 * k] = alpha[served + b];
        }
        resident_submit(n, rows);
        if (!resident_collect(n, llk_out + served)) break;       // plain launches below
        served += n;
    }
    pc1 += (size_t)served * k;
    pc2 += (size_t)served * k;
    alpha += served;
    llk_out += served;
    num_point -= served;
    for (int done = 0; done < num_point; done += kStagePoints) {
        const int n = std::min(kStagePoints, num_point - done);
        for (int b = 0; b < n; ++b) {
            double* row = h_points + (size_t)b * stride;
            std::memcpy(row, pc1 + (size_t)(done + b) * k, sizeof(double) * k);
            std::memcpy(row + k, pc2 + (size_t)(done + b) * k, sizeof(double) * k);
            row[2 * k] = alpha[done + b];
        }
        // The kernel that produces the last result also publishes a sequence number to
        // mapped host memory; spinning on it costs a few microseconds less per call than
        // hipStreamSynchronize (which matters: a search is ~350 dependent calls).
        // A NaN result is the tagged hand-off's "a workgroup never reported" marker (the grid was
        // not co-resident, e.g. a device shared with another process).  It must never reach the
        // optimiser -- every Nelder-Mead comparison with a NaN is false -- so the launch is redone
        // once with the arrival-ticket hand-off, which does not need co-residency.  A NaN that
        // survives that is the input's own (NaN parameters) and is passed on as the value it is.
        for (int attempt = 0; attempt < 2; ++attempt) {
            const unsigned long long seq = ++done_seq_;
            for (int b = 0; b < n; ++b) h_out[b] = std::numeric_limits<double>::quiet_NaN();
            int rc = eval_device(n, d_points, d_out, stream, L.num_mt > 0 && spin_wait ? d_done : nullptr, seq,
                                 h_points, attempt == 0 ? 0 : 1);
            if (rc) return rc;
            bool seen = false;
            if (L.num_mt > 0 && spin_wait) {
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned spins = 0;; ++spins) {
                    if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
                    if ((spins & 0x3ff) == 0x3ff &&
                        std::chrono::steady_clock::now() - t0 > std::chrono::seconds(4)) break;
                    __builtin_ia32_pause();
                }
            }
            if (!seen) VB2_HIP(hipStreamSynchronize(stream));
            if (settle_results(h_out, n)) break;
            if (attempt == 0) ++nan_retries;
        }
        std::memcpy(llk_out + done, h_out, sizeof(double) * n);
    }
    // NaN parameters: the reference's rule (context.h: params_hold_nan)
    for (int b = 0; b < num_point + served; ++b)
        if (L.num_mt > 0 && params_hold_nan(pc1_all + (size_t)b * k, pc2_all + (size_t)b * k, alpha_all[b], k, L.known_af != nullptr))
            llk_all[b] = 0.0;
    return VB2_OK;
}

int Context::ensure_codes16()
{
    //  two threads may build batches over one context: one of them makes the copy, the other waits
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (L.codes16 || L.num_mt == 0) return VB2_OK;
    // the pack kernel goes on this context's stream: behind a resident search kernel it would only start when that
    // kernel gives up after its idle second (and end the session without a word) -- end the session first
    if (resident_active) resident_end();
    VB2_HIP(hipSetDevice(device));
    std::vector<uint2> rec16(L.num_mt);
    uint64_t total16 = 0;
    for (int t = 0; t < L.num_mt; ++t) {
        // (probability domain: 8-bit steps, four to a 32-bit word -- the record keeps the tile's {ref steps | all steps << 16})
        const uint32_t r16 = L.pd ? ((h_mt_rec_y[t] >> 16) + 3u) / 4u : (h_mt_rows[t] + 1) / 2;
        rec16[t] = make_uint2((uint32_t)total16, L.pd ? h_mt_rec_y[t] : r16);
        total16 += r16;
    }
    const size_t rec_bytes = ((size_t)L.num_mt * sizeof(uint2) + 255) & ~(size_t)255;
    const size_t bytes = rec_bytes + (size_t)(total16 + kCodeSlackRows) * kMtMarkers * (L.pd ? sizeof(uint32_t) : sizeof(uint2));
    VB2_HIP(hipMalloc(&d_codes16_own, bytes));
    char* base = static_cast<char*>(d_codes16_own);
    const uint2* d_rec = reinterpret_cast<const uint2*>(base);
    uint2* d16 = reinterpret_cast<uint2*>(base + rec_bytes);
    {
        // a failure after the allocation must not leave the block behind (the next call would allocate again)
        hipError_t e = hipMemcpyAsync(base, rec16.data(), rec16.size() * sizeof(uint2), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess)
            e = L.pd ? launch_pack_pd_codes8(L, reinterpret_cast<uint32_t*>(d16), d_rec, (uint32_t)total16, stream)
                     : launch_pack_codes16(L, d16, d_rec, (uint32_t)total16, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);   // (rec16 is a pageable source; and the lists must be complete)
        if (e != hipSuccess) {
            (void)hipFree(d_codes16_own);
            d_codes16_own = nullptr;
            set_error(std::string("vb2_batch_create: building the 16-bit run lists failed: ") + hipGetErrorString(e));
            return VB2_ERR_HIP;
        }
    }
    L.mt_rec16 = d_rec;
    L.codes16 = d16;
    int64_t rows32 = 0;
    for (int t = 0; t < L.num_mt; ++t) rows32 += h_mt_rows[t];
    // (what a cohort step streams of this sample: the shorter lists instead of the context's own)
    cohort_bytes += L.pd ? ((int64_t)total16 - rows32) * kMtMarkers * 4 : ((int64_t)total16 - rows32) * kMtMarkers * 8;
    device_bytes += (int64_t)bytes;
    return VB2_OK;
}

int Context::layout_digest(unsigned long long* digest)
{
    // the same regions, in the same order, as the dry flatten's digest (flatten_digest) -- read back from the device
    VB2_HIP(hipSetDevice(device));
    resident_end();
    VB2_HIP(hipStreamSynchronize(stream));
    uint64_t hsh = 1469598103934665603ull;
    std::vector<unsigned char> buf;
    for (const auto& r : dbg_regions) {
        buf.resize(r.second);
        if (r.second) VB2_HIP(hipMemcpy(buf.data(), static_cast<const char*>(d_slab) + r.first, r.second, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < r.second; ++i) hsh = (hsh ^ buf[i]) * 1099511628211ull;
    }
    const unsigned char* q = reinterpret_cast<const unsigned char*>(dbg_counts);
    for (size_t i = 0; i < sizeof(dbg_counts); ++i) hsh = (hsh ^ q[i]) * 1099511628211ull;
    *digest = hsh;
    return VB2_OK;
}

int Context::read_stamps(unsigned long long* out, int max_blocks)
{
    if (!d_stamps) return 0;
    const int nb = std::min(max_blocks, max_grid(L));
    if (hipMemcpy(out, d_stamps, sizeof(unsigned long long) * 8 * nb, hipMemcpyDeviceToHost) != hipSuccess)
        return 0;
    return nb;
}

void Context::fill_info(vb2_info* info) const
{
    std::memset(info, 0, sizeof(*info));
    info->abi_version = VB2_ABI_VERSION;
    info->device = device;
    info->num_marker = num_marker;
    info->num_pc = num_pc;
    info->num_active_marker = L.num_active;
    info->num_read = num_read;
    info->num_read_other = num_read_other;
    info->num_code = num_code_seen;
    info->num_tile = L.num_mt;
    info->device_bytes = device_bytes;
    info->algorithmic_bytes_per_eval = algorithmic_bytes;
    info->cohort_step_bytes = cohort_bytes;
    info->layout = L.pd;
    info->num_table_row = L.num_code;
    int64_t rows = 0;
    for (uint32_t r : h_mt_rows) rows += r;
    info->num_step = rows * 2 * kMtMarkers;
    std::snprintf(info->device_name, sizeof(info->device_name), "%s", device_name);
    std::snprintf(info->arch, sizeof(info->arch), "%s", arch);
}

}  // namespace vb2

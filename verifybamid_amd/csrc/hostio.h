// hostio.h -- host-side input adapters and writers for the --SVDPrefix /
// --PileupFile flow: the "flatten once" stage that feeds vb2_ctx_create.
//
// Mirrors (file:line relative to the reference root):
//   ReadChooseBed / ReadMatrixUD / ReadMean / ReadAF   ContaminationEstimator.cpp:342-487
//   SimplePileupViewer::ReadPileup                     SimplePileupViewer.cpp:711-833
//   BuildResolvedMarkers                               ContaminationEstimator.cpp:67-86
//   IsSanityCheckOK                                    ContaminationEstimator.cpp:543-587
//   .Ancestry / .selfSM / .Pileup writers              ContaminationEstimator.cpp:168-188, main.cpp:336-411
#ifndef VB2_HOSTIO_H_
#define VB2_HOSTIO_H_

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/vb2_abi.h"

namespace vb2 {

// The .bed rows name positions; a position listed twice is ONE entry of the reference's
// ChooseBed[chr][pos] map (the later row's alleles win, cpp:433).  Here every distinct (chr, pos)
// gets a dense id in order of first appearance -- a "slot" -- so that everything per position is a
// flat array: one hash lookup per pileup line, none per marker afterwards.
struct Panel {
    int numPC = 0;
    uint32_t NumMarker = 0;                               // rows of .UD (cpp:366)
    std::vector<double> UD;                               // NumMarker x numPC
    std::vector<double> means;
    std::vector<std::pair<std::string, int>> PosVec;      // cpp:432 (pos is 1-based)
    // ChooseBed (cpp:433) as slots
    std::unordered_map<std::string, std::unordered_map<int, int32_t>> slotOf;   // (chr, pos) -> slot
    std::vector<int32_t> rowSlot;                         // PosVec row -> slot
    std::vector<char> slotRef, slotAlt;                   // ChooseBed[chr][pos].first / .second
    std::vector<int32_t> slotPos, slotChr;                // the slot's position; index into chrNames
    std::vector<std::string> chrNames;
    bool isAFknown = false;
    std::unordered_map<std::string, std::unordered_map<uint32_t, double>> knownAF;
    std::vector<double> slotAF;                           // knownAF[chr][pos] per slot (0 where the AF file has no row)
    size_t num_slot() const { return slotPos.size(); }
    int32_t find_slot(const std::string& chr, int pos) const
    {
        auto c = slotOf.find(chr);
        if (c == slotOf.end()) return -1;
        auto s = c->second.find(pos);
        return s == c->second.end() ? -1 : s->second;
    }
    void finish();                                        // after read_bed (+ read_known_af): slotAF
};

// SimplePileupViewer.h:84-147: baseInfo / qualInfo / posIndex.  A "site" is a pileup line whose
// position is in the .bed, numbered in order of appearance (the reference's global index); its
// parsed bases and qualities sit back to back in two pools instead of one std::string each.
struct PileupViewer {
    std::string basePool, qualPool;
    std::vector<uint32_t> siteOff{0};                     // site s owns [siteOff[s], siteOff[s + 1])
    std::vector<int32_t> siteOfSlot;                      // panel slot -> site, -1 = not in the pileup (posIndex)
    std::string SEQ_SM = "DefaultSampleName";
    int numBases = 0;
    int effectiveNumSite = 0;
    double avgDepth = 0;
    double sdDepth = 0;
    void init(const Panel& p)
    {
        siteOfSlot.assign(p.num_slot(), -1);
        siteOff.assign(1, 0u);
        basePool.clear();
        qualPool.clear();
    }
    int num_site() const { return (int)siteOff.size() - 1; }
    int32_t site_of(int32_t slot) const { return slot >= 0 && (size_t)slot < siteOfSlot.size() ? siteOfSlot[slot] : -1; }
    uint32_t depth(int32_t site) const { return siteOff[site + 1] - siteOff[site]; }
    const char* bases(int32_t site) const { return basePool.data() + siteOff[site]; }
    const char* quals(int32_t site) const { return qualPool.data() + siteOff[site]; }
    // a new site for `slot` with n parsed (base, quality) pairs
    void add_site(int32_t slot, const char* b, const char* q, size_t n)
    {
        if (basePool.size() + n > 0xffffffffull)      // (32-bit site offsets; the callers turn this into VB2_ERR_INVALID)
            throw std::length_error("pileup too large: more than 4 GiB of bases at the panel's sites");
        siteOfSlot[slot] = num_site();
        basePool.append(b, n);
        qualPool.append(q, n);
        siteOff.push_back((uint32_t)basePool.size());
    }
};

int read_bed(const std::string& path, Panel* p);
int read_ud(const std::string& path, Panel* p);
int read_mean(const std::string& path, Panel* p);
int read_known_af(const std::string& path, Panel* p);
int read_pileup(const std::string& path, const Panel& panel, PileupViewer* v);
// BAM/CRAM input through htslib (bam_flatten.cpp; VB2_ERR_IO with an explanation when the library was
// built without htslib).  SimplePileupViewer.cpp:172-557 with main.cpp:81-96's defaults.
// mp: the reference's "Pileup Options" (main.cpp:176-187), nullptr or given == 0 = its defaults (main.cpp:81-96)
int read_bam(const std::string& bam_path, const std::string& ref_path, const Panel& panel, PileupViewer* v,
             const vb2_mpileup_opts* mp = nullptr);
bool bam_support();
bool sanity_check(const Panel& p, PileupViewer* v);

}  // namespace vb2

// The flattened, panel-ordered arrays behind a vb2_input (owned here).
struct vb2_flat {
    std::shared_ptr<vb2::Panel> panel_ptr;    // one panel can serve a whole cohort of samples
    vb2::Panel& panel;                        // = *panel_ptr
    vb2_flat() : panel_ptr(std::make_shared<vb2::Panel>()), panel(*panel_ptr) {}
    explicit vb2_flat(std::shared_ptr<vb2::Panel> shared) : panel_ptr(std::move(shared)), panel(*panel_ptr) {}
    vb2_flat(const vb2_flat&) = delete;
    vb2_flat& operator=(const vb2_flat&) = delete;
    vb2::PileupViewer viewer;
    std::vector<int64_t> read_off;
    std::string bases, quals;
    std::vector<char> alt_base;
    std::vector<double> known_af;
    int32_t num_site = 0;
    bool sanity_disabled = true;
    vb2_input input{};
    void resolve();          // BuildResolvedMarkers -> input
};

namespace vb2 {
int write_ancestry(const std::string& prefix, int numPC, const double* pc, const double* pc2);
int write_selfsm(const std::string& prefix, const vb2_flat& f, const vb2_estimate& est,
                 bool pileup_input);
int write_pileup(const std::string& prefix, const vb2_flat& f);
void print_summary(const char* title, int numPC, const vb2_estimate& est);
// The panel files of a run (abi.cpp): .UD and .mu parsed on helper threads while the caller's
// thread reads the .bed (and the AF file); errors in the reference's reading order.
int load_panel(const vb2_run_args* a, Panel* panel);
}  // namespace vb2

#endif

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
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/vb2_abi.h"

namespace vb2 {

typedef std::unordered_map<std::string, std::unordered_map<int, std::pair<char, char>>> BedTable;

struct Panel {
    int numPC = 0;
    uint32_t NumMarker = 0;                               // rows of .UD (cpp:366)
    std::vector<double> UD;                               // NumMarker x numPC
    std::vector<double> means;
    std::vector<std::pair<std::string, int>> PosVec;      // cpp:432
    BedTable ChooseBed;                                   // cpp:433 (pos is 1-based)
    bool isAFknown = false;
    std::unordered_map<std::string, std::unordered_map<uint32_t, double>> knownAF;
};

struct PileupViewer {                                     // SimplePileupViewer.h:84-147
    std::vector<std::string> baseInfo, qualInfo;
    std::unordered_map<std::string, std::unordered_map<int32_t, int32_t>> posIndex;
    std::string SEQ_SM = "DefaultSampleName";
    int numBases = 0;
    int effectiveNumSite = 0;
    double avgDepth = 0;
    double sdDepth = 0;
};

int read_bed(const std::string& path, Panel* p);
int read_ud(const std::string& path, Panel* p);
int read_mean(const std::string& path, Panel* p);
int read_known_af(const std::string& path, Panel* p);
int read_pileup(const std::string& path, const BedTable& bed, PileupViewer* v);
// BAM/CRAM input through htslib (bam_flatten.cpp; VB2_ERR_IO with an explanation when the library was
// built without htslib).  SimplePileupViewer.cpp:172-557 with main.cpp:81-96's defaults.
int read_bam(const std::string& bam_path, const std::string& ref_path, const Panel& panel, PileupViewer* v);
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

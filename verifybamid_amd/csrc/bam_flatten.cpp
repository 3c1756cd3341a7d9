// bam_flatten.cpp -- BAM/CRAM -> per-marker pileup on the host, the
// "SimplePileupViewer runs once on host" stage of BASELINE.json's north_star for --BamFile input.
//
// Behaviour restated from the reference (file:line relative to the reference root):
//   read filter + BAQ + mapQ cap       SimplePileupViewer.cpp:172-237 (mplp_func)
//   per-region pileup over the panel   SimplePileupViewer.cpp:245-557 (SimplePileup)
//   base characters                    SimplePileupViewer.cpp:25-62   (pileup_seq)
//   defaults                           main.cpp:81-96: min-MQ 2, min-BQ 13, adjust-MQ (capQ) 40,
//                                      max depth 8000, REALN (BAQ) + SMART_OVERLAPS, read filter
//                                      UNMAP | SECONDARY | QCFAIL | DUP
// It fills the same PileupViewer (sites in two pools, hostio.h) the text-pileup reader fills, so the
// sanity check, BuildResolvedMarkers and the flattening into pinned SoA arrays are shared.
//
// NEEDS htslib (>= 1.10: hts_pos_t), which neither this build image nor the GPU box has (probed
// in round 3: no htslib/sam.h, no libhts anywhere on either).  The file is compiled only on an
// explicit opt-in, `cmake -DVB2_WITH_HTSLIB=ON` (default OFF since round 3: a machine that happens
// to have libhts must not silently ship an unvalidated reader); otherwise read_bam() reports that
// and --BamFile fails loudly.  STATUS: written against htslib's public API, NOT compiled and NOT
// validated anywhere (no htslib, and the reference's resource/test/test.bam is absent): parity at
// this boundary is UNPINNED -- validate with `--OutputPileup` against the reference's
// expected/result.Pileup where htslib and a BAM exist before relying on it.
//
// Two behaviours of the reference that look like slips are kept on purpose, because the golden
// outputs were produced with them (SimplePileupViewer.cpp:448-476, 502-511):
//   * a read that has a DELETION at the site contributes no base (pileup_seq, :32) but its quality
//     still goes to qualInfo (:462-470), so from that read on the site's bases and qualities are
//     paired off by one -- ComputeMixLLKs walks baseInfo's length (ContaminationEstimator.h:285-298).
//     Here: the qualities of deleted reads are kept in the list and the first bases.size() of them
//     are what add_site pairs with the bases;
//   * every position the pileup engine reports inside a region is indexed (posIndex, :427-438), even
//     if every read failed min-BQ and the site ends up empty; a marker no read covers is never
//     reported, hence never indexed (siteOfSlot stays -1: BuildResolvedMarkers' "absent").
#include "hostio.h"

#include "context.h"   // set_error

#ifdef VB2_WITH_HTSLIB
#include <htslib/faidx.h>
#include <htslib/hts.h>
#include <htslib/sam.h>

#include <cctype>
#include <cstring>
#include <string>
#include <vector>

namespace vb2 {
namespace {

// the reference's mode bits of mplp.flag (SimplePileupViewer.h:14-24) -- `--incl-flags` stores its value THERE
// (main.cpp:185-186), not in a required-read-flags mask: kept
constexpr int kMplpNoOrphan = 1 << 3, kMplpRealn = 1 << 4, kMplpIllumina13 = 1 << 7, kMplpRedoBaq = 1 << 6,
              kMplpSmartOverlaps = 1 << 10;

struct MplpConf {                      // main.cpp:81-96
    int min_mq = 2, min_baseQ = 13, capQ_thres = 40, max_depth = 8000;
    int flag = kMplpRealn | kMplpSmartOverlaps;
    uint32_t rflag_filter = BAM_FUNMAP | BAM_FSECONDARY | BAM_FQCFAIL | BAM_FDUP;
    void apply(const vb2_mpileup_opts* mp)      // main.cpp:176-187, 208-211
    {
        if (!mp || !mp->given) return;
        min_baseQ = mp->min_bq;
        min_mq = mp->min_mq;
        capQ_thres = mp->adjust_mq;
        max_depth = mp->max_depth;
        flag = mp->incl_flags;
        if (mp->no_orphans) flag |= kMplpNoOrphan;
        else flag &= ~kMplpNoOrphan;
        rflag_filter = (uint32_t)mp->excl_flags;
    }
};

struct Aux {
    samFile* fp = nullptr;
    sam_hdr_t* hdr = nullptr;
    hts_itr_t* iter = nullptr;
    faidx_t* fai = nullptr;
    char* ref = nullptr;               // sequence of ref_tid (cached: consecutive markers share it)
    int ref_tid = -1;
    hts_pos_t ref_len = 0;
    MplpConf conf;
};

bool fetch_ref(Aux* a, int tid)
{
    if (!a->fai) return false;
    if (tid == a->ref_tid) return a->ref != nullptr;
    free(a->ref);
    a->ref = nullptr;
    a->ref_tid = tid;
    hts_pos_t len = 0;
    a->ref = faidx_fetch_seq64(a->fai, sam_hdr_tid2name(a->hdr, tid), 0, HTS_POS_MAX, &len);
    a->ref_len = a->ref ? len : 0;
    return a->ref != nullptr;
}

// the pileup engine's read source: same order of tests as mplp_func (SimplePileupViewer.cpp:172-237)
int next_read(void* data, bam1_t* b)
{
    Aux* a = static_cast<Aux*>(data);
    int ret;
    for (;;) {
        ret = a->iter ? sam_itr_next(a->fp, a->iter, b) : sam_read1(a->fp, a->hdr, b);
        if (ret < 0) break;
        if (b->core.tid < 0 || (b->core.flag & BAM_FUNMAP)) continue;
        if (a->conf.rflag_filter & b->core.flag) continue;
        if (a->conf.flag & kMplpIllumina13) {                          // SimplePileupViewer.cpp:203-208
            uint8_t* qual = bam_get_qual(b);
            for (int i = 0; i < b->core.l_qseq; ++i) qual[i] = qual[i] > 31 ? qual[i] - 31 : 0;
        }
        const bool has_ref = fetch_ref(a, b->core.tid);
        if (has_ref && a->ref_len <= b->core.pos) continue;            // read outside the reference sequence
        if (has_ref && (a->conf.flag & kMplpRealn))                    // BAQ
            sam_prob_realn(b, a->ref, a->ref_len, (a->conf.flag & kMplpRedoBaq) ? 7 : 3);
        if (has_ref && a->conf.capQ_thres > 10) {
            const int q = sam_cap_mapq(b, a->ref, a->ref_len, a->conf.capQ_thres);
            if (q < 0) continue;
            if (b->core.qual > q) b->core.qual = (uint8_t)q;
        }
        if (b->core.qual < a->conf.min_mq) continue;
        if ((a->conf.flag & kMplpNoOrphan) && (b->core.flag & BAM_FPAIRED) && !(b->core.flag & BAM_FPROPER_PAIR)) continue;
        break;
    }
    return ret;
}

std::string sample_name(sam_hdr_t* hdr)     // @RG SM: (SimplePileupViewer.cpp:296-314; several samples are an error there)
{
    const char* text = sam_hdr_str(hdr);
    std::string sm;
    for (const char* p = text ? std::strstr(text, "@RG") : nullptr; p; p = std::strstr(p + 3, "@RG")) {
        const char* eol = std::strchr(p, '\n');
        const char* t = std::strstr(p, "\tSM:");
        if (!t || (eol && t > eol)) continue;
        t += 4;
        const char* e = t;
        while (*e && *e != '\t' && *e != '\n') ++e;
        const std::string one(t, e);
        if (sm.empty()) sm = one;
        else if (sm != one) return std::string();       // more than one sample
    }
    return sm.empty() ? std::string("DefaultSampleName") : sm;
}

}  // namespace

int read_bam(const std::string& bam_path, const std::string& ref_path, const Panel& panel, PileupViewer* v,
             const vb2_mpileup_opts* mp)
{
    Aux a;
    a.conf.apply(mp);
    hts_idx_t* idx = nullptr;
    struct Cleanup {                       // every exit path releases what was opened so far
        Aux& a;
        hts_idx_t*& idx;
        ~Cleanup()
        {
            if (a.iter) hts_itr_destroy(a.iter);
            free(a.ref);
            if (a.fai) fai_destroy(a.fai);
            if (idx) hts_idx_destroy(idx);
            if (a.hdr) sam_hdr_destroy(a.hdr);
            if (a.fp) sam_close(a.fp);
        }
    } cleanup{a, idx};
    a.fp = sam_open(bam_path.c_str(), "rb");
    if (!a.fp) { set_error("failed to open " + bam_path); return VB2_ERR_IO; }
    if (hts_set_fai_filename(a.fp, ref_path.c_str()) != 0) { set_error("failed to process " + ref_path); return VB2_ERR_IO; }
    a.hdr = sam_hdr_read(a.fp);
    if (!a.hdr) { set_error("fail to read the header of " + bam_path); return VB2_ERR_IO; }
    idx = sam_index_load(a.fp, bam_path.c_str());
    if (!idx) { set_error("fail to load index for " + bam_path); return VB2_ERR_IO; }
    a.fai = fai_load(ref_path.c_str());
    v->SEQ_SM = sample_name(a.hdr);
    if (v->SEQ_SM.empty()) {
        set_error("This BAM or CRAM file contains more than 1 sample, please demultiplex or separate first!");
        return VB2_ERR_INVALID;
    }
    v->init(panel);
    v->numBases = 0;
    std::string bases, quals;
    // one region per panel marker, in .bed order (the reference jumps region by region too)
    for (size_t row = 0; row < panel.PosVec.size(); ++row) {
        const auto& marker = panel.PosVec[row];
        const std::string& chr = marker.first;
        const int pos1 = marker.second;                                   // 1-based
        const int tid = sam_hdr_name2tid(a.hdr, chr.c_str());
        if (tid < 0) continue;
        const int32_t slot = panel.rowSlot[row];
        if (v->siteOfSlot[slot] >= 0) continue;                           // duplicated marker: skipped (cpp:424-427)
        bases.clear();
        quals.clear();
        bool reported = false;             // the pileup engine produced this position (>= 1 alignment over it)
        a.iter = sam_itr_queryi(idx, tid, pos1 - 1, pos1);
        if (a.iter) {
            bam_plp_t plp = bam_plp_init(next_read, &a);
            bam_plp_set_maxcnt(plp, a.conf.max_depth);
            if (a.conf.flag & kMplpSmartOverlaps) bam_plp_init_overlaps(plp);
            int ptid, ppos, n;
            const bam_pileup1_t* pl;
            while ((pl = bam_plp_auto(plp, &ptid, &ppos, &n)) != nullptr) {
                if (ptid != tid || ppos != pos1 - 1) continue;
                reported = true;
                const bool has_ref = fetch_ref(&a, tid);
                for (int j = 0; j < n; ++j) {
                    const bam_pileup1_t* p = pl + j;
                    const int q = p->qpos < p->b->core.l_qseq ? bam_get_qual(p->b)[p->qpos] : 0;
                    if (q < a.conf.min_baseQ) continue;                   // SimplePileupViewer.cpp:457, 467
                    // the quality of EVERY read that passes min-BQ, deleted ones included (:462-470) ...
                    quals.push_back((char)(q + 33 < 126 ? q + 33 : 126));
                    if (p->is_del) continue;                              // ... but a base only where there is one (:32)
                    int c = p->qpos < p->b->core.l_qseq ? seq_nt16_str[bam_seqi(bam_get_seq(p->b), p->qpos)] : 'N';
                    const bool rev = bam_is_rev(p->b);
                    if (has_ref) {                                        // pileup_seq, SimplePileupViewer.cpp:32-40
                        const int rb = ppos < a.ref_len ? a.ref[ppos] : 'N';
                        if (c == '=' || seq_nt16_table[c] == seq_nt16_table[rb]) c = rev ? ',' : '.';
                        else c = rev ? std::tolower(c) : std::toupper(c);
                    } else {
                        c = c == '=' ? (rev ? ',' : '.') : (rev ? std::tolower(c) : std::toupper(c));
                    }
                    bases.push_back((char)c);
                }
            }
            bam_plp_destroy(plp);
            hts_itr_destroy(a.iter);
            a.iter = nullptr;
        }
        if (!reported) continue;           // never seen by the reference's loop: not indexed (posIndex, :427-438)
        if (!bases.empty()) {
            v->effectiveNumSite++;
            v->numBases += (int)bases.size();
        }
        // bases.size() <= quals.size(): the pairing is by index, like the reference's (header of this file)
        v->add_site(slot, bases.data(), quals.data(), bases.size());
    }
    v->avgDepth = v->effectiveNumSite ? (double)v->numBases / v->effectiveNumSite : 0.0;
    return VB2_OK;
}

bool bam_support() { return true; }

}  // namespace vb2

#else   // ---- built without htslib ----

namespace vb2 {

int read_bam(const std::string&, const std::string&, const Panel&, PileupViewer*, const vb2_mpileup_opts*)
{
    set_error("--BamFile needs htslib, which this build does not have (configure with CMake where libhts is "
              "installed); run the reference once with --OutputPileup and pass the result with --PileupFile");
    return VB2_ERR_IO;
}

bool bam_support() { return false; }

}  // namespace vb2

#endif

// hostio.cpp -- see hostio.h.
#include "hostio.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "context.h"   // set_error

namespace vb2 {

namespace {

// The reference reads panel files through statgen's InputFile::readLine, which
// reports EOF (and so drops the data) for a last line that lacks '\n'
// (statgen/InputFile.cpp:112-130).  Same here.
bool terminated_lines(const std::string& path, std::vector<std::string>* lines)
{
    std::ifstream fin(path, std::ios::binary);
    if (!fin.is_open()) return false;
    std::string all((std::istreambuf_iterator<char>(fin)), std::istreambuf_iterator<char>());
    size_t beg = 0;
    for (;;) {
        const size_t nl = all.find('\n', beg);
        if (nl == std::string::npos) break;
        lines->emplace_back(all, beg, nl - beg);
        beg = nl + 1;
    }
    return true;
}

int io_error(const std::string& what)
{
    set_error(what);
    return VB2_ERR_IO;
}

}  // namespace

// ContaminationEstimator.cpp:413-438.  ref/alt are single chars: only the first
// character of an allele column is kept ("A,G" -> 'A').
int read_bed(const std::string& path, Panel* p)
{
    std::vector<std::string> lines;
    if (!terminated_lines(path, &lines)) return io_error("Open file:" + path + "\t failed");
    std::string chr;
    int pos = 0;
    char ref = 0, alt = 0;
    for (const std::string& line : lines) {
        std::stringstream ss(line);
        ss >> chr >> pos >> pos;
        ss >> ref >> alt;
        p->PosVec.push_back(std::make_pair(chr, pos));
        p->ChooseBed[chr][pos] = std::make_pair(ref, alt);
    }
    return VB2_OK;
}

// ContaminationEstimator.cpp:342-373: first numPC columns; fewer is fatal.
int read_ud(const std::string& path, Panel* p)
{
    std::vector<std::string> lines;
    if (!terminated_lines(path, &lines)) return io_error("Open file:" + path + "\t failed");
    std::vector<double> row(p->numPC, 0.);
    for (const std::string& line : lines) {
        std::stringstream ss(line);
        int index = 0;
        while (index < p->numPC && ss >> row[index]) index++;
        if (index < p->numPC) {
            char msg[256];
            std::snprintf(msg, sizeof(msg),
                          "--NumPC should be less than or equal to the number of PCs in SVD files "
                          "provided by --SVDPrefix! (Expected:%d vs Observed:%d)", p->numPC, index);
            set_error(msg);
            return VB2_ERR_INVALID;
        }
        p->UD.insert(p->UD.end(), row.begin(), row.end());
        p->NumMarker++;
    }
    return VB2_OK;
}

// ContaminationEstimator.cpp:440-459: second column.
int read_mean(const std::string& path, Panel* p)
{
    std::vector<std::string> lines;
    if (!terminated_lines(path, &lines)) return io_error("Open file:" + path + "\t failed");
    double mu = 0;
    std::string name;
    for (const std::string& line : lines) {
        std::stringstream ss(line);
        ss >> name;
        ss >> mu;
        p->means.push_back(mu);
    }
    return VB2_OK;
}

// ContaminationEstimator.cpp:461-487 (std::getline: an unterminated last line counts).
int read_known_af(const std::string& path, Panel* p)
{
    std::ifstream fin(path);
    if (!fin.is_open()) return io_error("Open file:" + path + "\t failed");
    std::string line, chr;
    uint32_t pos = 0;
    double af = 0;
    char ref = 0, alt = 0;
    while (std::getline(fin, line)) {
        std::stringstream ss(line);
        ss >> chr;
        ss >> pos >> pos;
        ss >> ref >> alt;
        ss >> af;
        p->knownAF[chr][pos] = af;
    }
    p->isAFknown = true;
    return VB2_OK;
}

namespace {
// SimplePileupViewer.cpp:711-746: keep ". , A C G T N a c g t n" (one quality
// each), drop "*" "#" (they still consume a quality), skip "^x", "+N..."/"-N...",
// ignore everything else ("$", ...).
void parse_bases(const std::string& seq, const std::string& qual, std::string* pseq,
                 std::string* pqual)
{
    pseq->clear();
    pqual->clear();
    size_t iq = 0;
    for (size_t i = 0; i < seq.size(); ++i) {
        const char c = seq[i];
        if (c == '+' || c == '-') {
            size_t j = i + 1;
            while (j != seq.size() && std::isdigit((unsigned char)seq[j])) j++;
            const size_t digit_len = j - (i + 1);
            const int clip = digit_len ? std::stoi(seq.substr(i + 1, digit_len)) : 0;
            i += digit_len + clip;
        } else if (c == '^') {
            i += 1;
        } else if (c == '.' || c == ',' || c == 'A' || c == 'G' || c == 'C' || c == 'T' || c == 'N' ||
                   c == 'a' || c == 'g' || c == 'c' || c == 't' || c == 'n') {
            *pseq += c;
            *pqual += iq < qual.size() ? qual[iq] : '!';
            ++iq;
        } else if (c == '*' || c == '#') {
            ++iq;
        }
    }
}
}  // namespace

// SimplePileupViewer.cpp:748-833.
int read_pileup(const std::string& path, const BedTable& bed, PileupViewer* v)
{
    std::ifstream fin(path);
    if (!fin.is_open()) return io_error("open file " + path + " failed!");
    int global_index = 0;
    std::string chr, ref, seq, qual, line, pseq, pqual;
    int pos = 0, depth = 0;
    v->numBases = 0;
    while (std::getline(fin, line)) {
        std::stringstream ss(line);
        ss >> chr >> pos >> ref >> depth >> seq >> qual;   // fields persist across malformed lines
        if (seq.find_first_of(".,") != std::string::npos && ref == ".") {
            set_error("Pileup format error: cannot find ref allele, exit!");
            return VB2_ERR_INVALID;
        }
        parse_bases(seq, qual, &pseq, &pqual);
        depth = (int)pqual.length();                        // SNP bases only
        auto bc = bed.find(chr);
        if (bc == bed.end() || bc->second.find(pos) == bc->second.end()) continue;

        bool existed = false;
        auto pc = v->posIndex.find(chr);
        if (pc != v->posIndex.end() && pc->second.find(pos) != pc->second.end()) existed = true;
        else v->posIndex[chr][pos] = global_index++;
        if (existed) {
            std::cerr << "[WARNING] The pileup file has duplicated lines! Merged here" << std::endl;
            // the reference builds a merged copy and then drops it (quirk vii)
        } else {
            v->baseInfo.push_back(pseq);
            v->qualInfo.push_back(pqual);
        }
        v->numBases += depth;
        depth = 0;
        seq = "";
        qual = "";
        v->effectiveNumSite++;
    }
    v->avgDepth = (double)v->numBases / v->effectiveNumSite;
    return VB2_OK;
}

// ContaminationEstimator.cpp:543-587.
bool sanity_check(const Panel& p, PileupViewer* v)
{
    std::fprintf(stderr, "NOTICE - Number of marker in Reference Matrix:%d\n", (int)p.NumMarker);
    std::fprintf(stderr, "NOTICE - Number of marker shared with input file:%d\n", v->effectiveNumSite);
    auto depth_at = [&](size_t i, int* d) {
        if (i >= p.PosVec.size()) return false;
        auto c = v->posIndex.find(p.PosVec[i].first);
        if (c == v->posIndex.end()) return false;
        auto s = c->second.find(p.PosVec[i].second);
        if (s == c->second.end()) return false;
        *d = (int)v->baseInfo[s->second].size();
        return true;
    };
    int d = 0;
    for (size_t i = 0; i < p.NumMarker; ++i)
        if (depth_at(i, &d)) v->sdDepth += d * d;
    v->sdDepth = std::sqrt(v->sdDepth / v->effectiveNumSite - v->avgDepth * v->avgDepth);
    v->effectiveNumSite = 0;
    for (size_t i = 0; i < p.NumMarker; ++i) {
        if (!depth_at(i, &d)) continue;
        if (d == 0 || d < (v->avgDepth - 3 * v->sdDepth) || d > (v->avgDepth + 3 * v->sdDepth)) continue;
        v->effectiveNumSite++;
    }
    std::fprintf(stderr, "NOTICE - Mean Depth:%f\n", v->avgDepth);
    std::fprintf(stderr, "NOTICE - SD Depth:%f\n", v->sdDepth);
    std::fprintf(stderr, "NOTICE - %d SNP markers remained after sanity check.\n", v->effectiveNumSite);
    return v->effectiveNumSite > 1000 && v->effectiveNumSite > (p.NumMarker * 0.1);
}

// ContaminationEstimator.cpp:168-188 (default ostream formatting, 6 significant digits).
int write_ancestry(const std::string& prefix, int numPC, const double* pc, const double* pc2)
{
    const std::string name(prefix + ".Ancestry");
    std::ofstream fout(name);
    if (!fout.is_open()) return io_error("Open file " + name + " failed!");
    fout << "PC\tContaminatingSample\tIntendedSample" << std::endl;
    for (int i = 0; i < numPC; ++i) fout << i + 1 << "\t" << pc[i] << "\t" << pc2[i] << std::endl;
    fout.close();
    if (!fout) return io_error("Errors detected when writing to file " + name + " !");
    return VB2_OK;
}

// main.cpp:386-411.
int write_selfsm(const std::string& prefix, const vb2_flat& f, const vb2_estimate& est,
                 bool pileup_input)
{
    const std::string name(prefix + ".selfSM");
    std::ofstream fout(name);
    if (!fout.is_open()) return io_error("Open file " + name + " failed!");
    fout << "#SEQ_ID\tRG\tCHIP_ID\t#SNPS\t#READS\tAVG_DP\tFREEMIX\tFREELK1\tFREELK0\tFREE_RH\tFREE_RA\t"
            "CHIPMIX\tCHIPLK1\tCHIPLK0\tCHIP_RH\tCHIP_RA\tDPREF\tRDPHET\tRDPALT"
         << std::endl;
    fout << f.viewer.SEQ_SM << "\tNA\tNA\t" << f.panel.NumMarker << "\t";
    if (pileup_input) fout << "NA";
    else fout << f.viewer.numBases;
    fout << "\t" << f.viewer.avgDepth << "\t" << ((est.alpha < 0.5) ? est.alpha : (1.f - est.alpha))
         << "\t" << -est.llk1 << "\t" << -est.llk0 << "\t"
         << "NA\tNA\t" << "NA\tNA\tNA\tNA\tNA\t" << "NA\tNA\tNA" << std::endl;
    fout.close();
    if (!fout) return io_error("Errors detected when writing to file " + name + " !");
    return VB2_OK;
}

// main.cpp:336-369: sites with depth > 0 in bed order.
int write_pileup(const std::string& prefix, const vb2_flat& f)
{
    const std::string name(prefix + ".Pileup");
    std::ofstream fout(name);
    if (!fout.is_open()) return io_error("Open file " + name + " failed!");
    for (const auto& item : f.panel.PosVec) {
        auto c = f.viewer.posIndex.find(item.first);
        if (c == f.viewer.posIndex.end()) continue;
        auto s = c->second.find(item.second);
        if (s == c->second.end()) continue;
        const std::string& b = f.viewer.baseInfo[s->second];
        if (b.empty()) continue;
        fout << item.first << "\t" << item.second << "\t"
             << f.panel.ChooseBed.at(item.first).at(item.second).first << "\t" << b.size() << "\t"
             << b << "\t" << f.viewer.qualInfo[s->second] << std::endl;
    }
    fout.close();
    if (!fout) return io_error("Errors detected when writing to file " + name + " !");
    return VB2_OK;
}

// stdout block of ContaminationEstimator.cpp:100-166.
void print_summary(const char* title, int numPC, const vb2_estimate& est)
{
    if (title) std::cout << title << std::endl;
    std::cout << "Contaminating Sample ";
    for (int i = 0; i < numPC; ++i) std::cout << "PC" << i + 1 << ":" << est.pc[i] << "\t";
    std::cout << std::endl;
    std::cout << "Intended Sample ";
    for (int i = 0; i < numPC; ++i) std::cout << "PC" << i + 1 << ":" << est.pc2[i] << "\t";
    std::cout << std::endl;
    std::cout << "FREEMIX(Alpha):" << (est.alpha < 0.5 ? est.alpha : (1 - est.alpha)) << std::endl;
}

}  // namespace vb2

// ContaminationEstimator.cpp:67-86, emitting panel-ordered offsets instead of
// an index into the viewer: marker i owns [read_off[i], read_off[i+1]).
void vb2_flat::resolve()
{
    const uint32_t M = panel.NumMarker;
    read_off.assign((size_t)M + 1, 0);
    alt_base.assign(M, 0);
    if (panel.isAFknown) known_af.assign(M, 0.0);
    bases.clear();
    quals.clear();
    num_site = 0;
    for (uint32_t i = 0; i < M; ++i) {
        read_off[i] = (int64_t)bases.size();
        if (i >= panel.PosVec.size()) continue;
        const std::string& chr = panel.PosVec[i].first;
        const int pos = panel.PosVec[i].second;
        auto c = viewer.posIndex.find(chr);
        if (c == viewer.posIndex.end()) continue;
        auto s = c->second.find(pos);
        if (s == c->second.end()) continue;
        ++num_site;
        bases += viewer.baseInfo[s->second];
        quals += viewer.qualInfo[s->second];
        alt_base[i] = panel.ChooseBed[chr][pos].second;
        if (panel.isAFknown) known_af[i] = panel.knownAF[chr][(uint32_t)pos];
    }
    read_off[M] = (int64_t)bases.size();
    input = vb2_input{};
    input.num_marker = (int32_t)M;
    input.num_pc = panel.numPC;
    input.ud = panel.UD.data();
    input.means = panel.means.data();
    input.read_off = read_off.data();
    input.bases = bases.data();
    input.quals = quals.data();
    input.alt_base = alt_base.data();
    input.known_af = panel.isAFknown ? known_af.data() : nullptr;
    input.avg_depth = viewer.avgDepth;
    input.sd_depth = viewer.sdDepth;
    input.sanity_disabled = sanity_disabled ? 1 : 0;
}

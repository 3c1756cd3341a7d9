// cli_main.cpp -- command line with the reference's flag names, defaults and
// output files for the contamination path (main.cpp:56-414), on top of the C-ABI.
//
//   VerifyBamID --SVDPrefix P --PileupFile F --Reference R [--NumPC k] [--Output o]
//               [--WithinAncestry] [--FixPC a:b:..] [--FixAlpha x] [--KnownAF f]
//               [--Epsilon e] [--DisableSanityCheck] [--OutputPileup] [--Verbose]
//               [--NumThread n] [--Seed s]      (+ deprecated --UDPath/--MeanPath/--BedPath)
//               [--min-BQ n] [--min-MQ n] [--adjust-MQ n] [--max-depth n] [--no-orphans]
//               [--incl-flags n] [--excl-flags n]      (the reference's "Pileup Options", main.cpp:176-187: --BamFile only)
//
// --BamFile needs a build with htslib (CMake finds it: bam_flatten.cpp); a build without it --
// this image's -- reports that and suggests the reference's own --OutputPileup file with --PileupFile.
// Extensions: --Device n selects the GPU; --Devices a,b,.. uses several -- one sample's markers
// are sharded over them (partial log-likelihoods met in one RCCL all-reduce), a --PileupList
// cohort is dealt to them group by group; --PileupList F runs many samples against one panel.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/vb2_abi.h"

namespace {

struct Flag {
    enum Kind { kBool, kInt, kDouble, kString } kind;
    void* dst;
    bool seen;
};

void fatal(const char* msg)
{
    std::fprintf(stderr, "\nFATAL ERROR - \n%s\n\n", msg);
    std::exit(EXIT_FAILURE);
}

}  // namespace

int main(int argc, char** argv)
{
    std::fprintf(stderr,
                 "VerifyBamID2 (MI355X-native likelihood core): DNA contamination estimation from "
                 "sequence reads using ancestry-agnostic method.\n\n");

    // defaults: main.cpp:58-79
    std::string UDPath("Empty"), MeanPath("Empty"), BedPath("Empty"), BamFile("Empty"),
        RefPath("Empty"), outputPrefix("result"), PileupFile("Empty"), SVDPrefix("Empty"),
        knownAF("Empty"), fixPC("Empty"), PileupList("Empty"), Devices("Empty");
    double fixAlpha = -1., epsilon = 1e-8;
    bool withinAncestry = false, outputPileup = false, verbose = false, disableSanityCheck = false;
    int seed = 12345, nPC = 2, nthread = 4, device = -1, numStart = 1;
    bool lineSearch = false;
    // "Pileup Options" (main.cpp:176-187), defaults main.cpp:81-96 (MPLP_REALN | MPLP_SMART_OVERLAPS; UNMAP | SECONDARY |
    // QCFAIL | DUP): they shape what --BamFile input becomes and, as in the reference, do nothing to --PileupFile input
    int minBQ = 13, minMQ = 2, adjustMQ = 40, maxDepth = 8000, inclFlags = (1 << 4) | (1 << 10), exclFlags = 0x4 | 0x100 | 0x200 | 0x400;
    bool noOrphans = false;

    std::map<std::string, Flag> flags = {
        {"BamFile", {Flag::kString, &BamFile, false}},
        {"PileupFile", {Flag::kString, &PileupFile, false}},
        {"Reference", {Flag::kString, &RefPath, false}},
        {"SVDPrefix", {Flag::kString, &SVDPrefix, false}},
        {"Output", {Flag::kString, &outputPrefix, false}},
        {"WithinAncestry", {Flag::kBool, &withinAncestry, false}},
        {"DisableSanityCheck", {Flag::kBool, &disableSanityCheck, false}},
        {"NumPC", {Flag::kInt, &nPC, false}},
        {"FixPC", {Flag::kString, &fixPC, false}},
        {"FixAlpha", {Flag::kDouble, &fixAlpha, false}},
        {"KnownAF", {Flag::kString, &knownAF, false}},
        {"NumThread", {Flag::kInt, &nthread, false}},
        {"Seed", {Flag::kInt, &seed, false}},
        {"Epsilon", {Flag::kDouble, &epsilon, false}},
        {"OutputPileup", {Flag::kBool, &outputPileup, false}},
        {"Verbose", {Flag::kBool, &verbose, false}},
        {"UDPath", {Flag::kString, &UDPath, false}},
        {"MeanPath", {Flag::kString, &MeanPath, false}},
        {"BedPath", {Flag::kString, &BedPath, false}},
        {"min-BQ", {Flag::kInt, &minBQ, false}},
        {"min-MQ", {Flag::kInt, &minMQ, false}},
        {"adjust-MQ", {Flag::kInt, &adjustMQ, false}},
        {"max-depth", {Flag::kInt, &maxDepth, false}},
        {"no-orphans", {Flag::kBool, &noOrphans, false}},
        {"incl-flags", {Flag::kInt, &inclFlags, false}},
        {"excl-flags", {Flag::kInt, &exclFlags, false}},
        {"Device", {Flag::kInt, &device, false}},
        {"Devices", {Flag::kString, &Devices, false}},
        // not in the reference: a cohort against one panel.  File of lines "<pileup>\t<output prefix>";
        // the panel is read once and the samples are searched in lock-step groups (vb2_cohort_run).
        {"PileupList", {Flag::kString, &PileupList, false}},
        // not in the reference: optimiser variants (vb2_search_opts).  --NumStart n searches from n
        // starting points in lock-step (start 0 = the reference's; the others add noise drawn from
        // --Seed) and reports the best; --LineSearch uses Brent's method (the reference's unused
        // MathGold) for the one-parameter models (--FixPC / --KnownAF).
        {"NumStart", {Flag::kInt, &numStart, false}},
        {"LineSearch", {Flag::kBool, &lineSearch, false}},
    };
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (std::strncmp(a, "--", 2) != 0) {
            std::fprintf(stderr, "WARNING - ignoring stray argument %s\n", a);
            continue;
        }
        auto it = flags.find(a + 2);
        if (it == flags.end()) {
            std::fprintf(stderr, "WARNING - unknown option %s ignored\n", a);
            continue;
        }
        Flag& f = it->second;
        if (f.seen) {   // params.cpp:114-185: an option may be given once
            std::string m = std::string("Option ") + a + " specified more than once";
            fatal(m.c_str());
        }
        f.seen = true;
        if (f.kind == Flag::kBool) {
            *static_cast<bool*>(f.dst) = true;
            continue;
        }
        if (i + 1 >= argc) {
            std::string m = std::string("Option ") + a + " needs a value";
            fatal(m.c_str());
        }
        const char* v = argv[++i];
        if (f.kind == Flag::kInt) *static_cast<int*>(f.dst) = std::atoi(v);
        else if (f.kind == Flag::kDouble) *static_cast<double*>(f.dst) = std::atof(v);
        else *static_cast<std::string*>(f.dst) = v;
    }
    // --Seed: parsed and never used by the reference (main.cpp:137,286); here it seeds --NumStart's starting points
    (void)nthread;  // the likelihood runs on the GPU; kept for command-line compatibility

    // main.cpp:214-232
    if (SVDPrefix == "Empty") {
        if (UDPath == "Empty") fatal("--UDPath is required when --RefVCF is absent");
        if (MeanPath == "Empty") fatal("--MeanPath is required when --RefVCF is absent");
        if (BedPath == "Empty") fatal("--BedPath is required when --RefVCF is absent");
    } else {
        UDPath = SVDPrefix + ".UD";
        MeanPath = SVDPrefix + ".mu";
        BedPath = SVDPrefix + ".bed";
    }
    if (RefPath == "Empty") fatal("--Reference is required");          // main.cpp:263-266
    if (PileupFile == "Empty" && PileupList == "Empty" && BamFile == "Empty")
        fatal("--BamFile or --PileupFile is required");                // main.cpp:278-281

    vb2_run_args args;
    std::memset(&args, 0, sizeof(args));
    args.ud_path = UDPath.c_str();
    args.mean_path = MeanPath.c_str();
    args.bed_path = BedPath.c_str();
    args.pileup_path = PileupFile == "Empty" ? nullptr : PileupFile.c_str();
    args.bam_path = BamFile == "Empty" ? nullptr : BamFile.c_str();       // needs an htslib build (bam_flatten.cpp)
    args.reference_path = RefPath.c_str();
    args.known_af_path = knownAF == "Empty" ? nullptr : knownAF.c_str();
    args.output_prefix = outputPrefix.c_str();
    args.num_pc = nPC;
    args.disable_sanity = disableSanityCheck;
    args.output_pileup = outputPileup;
    args.search.num_start = numStart;
    args.search.seed = (uint32_t)seed;
    args.search.line_search = lineSearch ? 1 : 0;
    args.device = device;
    std::vector<int32_t> devs;
    if (Devices != "Empty") {
        std::stringstream ds(Devices);
        std::string tok;
        while (std::getline(ds, tok, ',')) {
            if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos)
                fatal("--Devices takes a comma-separated list of device ordinals, e.g. 0,1,2,3");
            devs.push_back(std::atoi(tok.c_str()));
        }
        if (devs.empty() || devs.size() > 64) fatal("--Devices names no device (or more than 64)");
        args.devices = devs.data();
        args.num_device = (int32_t)devs.size();
        args.device = devs[0];
    }
    static const char* const kPileupOpts[] = {"min-BQ", "min-MQ", "adjust-MQ", "max-depth", "no-orphans", "incl-flags", "excl-flags"};
    for (const char* name : kPileupOpts) args.mpileup.given |= flags[name].seen ? 1 : 0;
    args.mpileup.min_bq = minBQ;
    args.mpileup.min_mq = minMQ;
    args.mpileup.adjust_mq = adjustMQ;
    args.mpileup.max_depth = maxDepth;
    args.mpileup.no_orphans = noOrphans ? 1 : 0;
    args.mpileup.incl_flags = inclFlags;
    args.mpileup.excl_flags = exclFlags;
    if (args.mpileup.given && BamFile == "Empty")
        std::fprintf(stderr, "NOTICE - pileup options (--min-BQ, --min-MQ, ...) apply to --BamFile input only; "
                             "the text pileup is taken as it is (as in the reference)\n");
    args.model.is_heter = !withinAncestry;
    args.model.epsilon = epsilon;
    args.model.verbose = verbose;
    args.model.notices = 1;

    std::vector<double> tmpPC;
    if (fixPC != "Empty") {                                            // main.cpp:291-308
        std::fprintf(stderr, "NOTICE - you specified --fixPC, this will overide dynamic estimation of PCs\n");
        std::stringstream ss(fixPC);
        std::string token;
        while (std::getline(ss, token, ':')) tmpPC.push_back(std::atof(token.c_str()));
        if ((int)tmpPC.size() > nPC)
            std::fprintf(stderr, "WARNING - parameter --fixPC provided larger dimension than parameter "
                                 "--numPC(default value 2) and hence will be truncated\n");
        if ((int)tmpPC.size() < nPC)
            fatal("parameter --fixPC provided smaller dimension than parameter --numPC(default value 2)");
        args.model.is_pc_fixed = 1;
        args.model.fix_pc = tmpPC.data();
    } else if (std::fabs(fixAlpha + 1.) > std::numeric_limits<double>::epsilon()) {   // main.cpp:309-313
        std::fprintf(stderr, "NOTICE - you specified --fixAlpha, this will overide dynamic estimation of alpha\n");
        args.model.is_alpha_fixed = 1;
        args.model.fix_alpha = fixAlpha;
    }
    if (args.known_af_path) args.model.is_af_known = 1;               // main.cpp:314-319

    if (PileupList != "Empty") {
        std::ifstream fl(PileupList);
        if (!fl.is_open()) fatal("cannot open --PileupList file");
        std::vector<std::string> pile, pref;
        std::string line;
        while (std::getline(fl, line)) {
            if (line.empty()) continue;
            const size_t tab = line.find('\t');
            pile.push_back(line.substr(0, tab));
            pref.push_back(tab == std::string::npos ? line + ".vb2" : line.substr(tab + 1));
        }
        if (pile.empty()) fatal("--PileupList file names no sample");
        std::vector<const char*> cpile, cpref;
        for (size_t i = 0; i < pile.size(); ++i) { cpile.push_back(pile[i].c_str()); cpref.push_back(pref[i].c_str()); }
        vb2_cohort_args ca;
        std::memset(&ca, 0, sizeof(ca));
        ca.base = args;
        ca.num_sample = (int32_t)pile.size();
        ca.pileup_paths = cpile.data();
        ca.output_prefixes = cpref.data();
        ca.num_host_thread = nthread > 4 ? nthread : 0;               // --NumThread above its default: reader threads
        std::vector<vb2_run_result> cres(pile.size());
        std::vector<int32_t> cst(pile.size());
        const int rcc = vb2_cohort_run(&ca, cres.data(), cst.data());
        if (rcc != VB2_OK) {
            std::fprintf(stderr, "\nFATAL ERROR - \n%s\n\n", vb2_last_error());
            return EXIT_FAILURE;
        }
        int bad = 0;
        std::printf("#PILEUP\tOUTPUT\tSTATUS\tFREEMIX\tFREELK1\tFREELK0\tAVG_DP\n");
        for (size_t i = 0; i < pile.size(); ++i) {
            const double al = cres[i].est.alpha;
            std::printf("%s\t%s\t%d\t%g\t%g\t%g\t%g\n", pile[i].c_str(), pref[i].c_str(), (int)cst[i],
                        al < 0.5 ? al : 1 - al, cres[i].est.llk1, cres[i].est.llk0, cres[i].avg_depth);
            if (cst[i] != VB2_OK) ++bad;
        }
        std::fprintf(stderr, "NOTICE - cohort of %zu samples done, %d failed their own checks\n", pile.size(), bad);
        return bad ? EXIT_FAILURE : 0;
    }

    vb2_run_result res;
    const int rc = vb2_run(&args, &res);
    if (rc != VB2_OK) {
        if (rc == VB2_ERR_SANITY) std::fprintf(stderr, "WARNING - %s\n", vb2_last_error());
        else std::fprintf(stderr, "\nFATAL ERROR - \n%s\n\n", vb2_last_error());
        return EXIT_FAILURE;
    }
    std::fprintf(stderr, "NOTICE - %lld likelihood evaluations, %lld points launched on the device\n",
                 (long long)res.est.num_eval, (long long)res.est.num_launch_point);
    std::fprintf(stderr, "NOTICE - Success!\n");
    return 0;
}

// hostio.cpp -- see hostio.h.
#include "tunables.h"
#include "hostio.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>
#include <vector>

#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "context.h"   // set_error

namespace vb2 {

namespace {

int io_error(const std::string& what)
{
    set_error(what);
    return VB2_ERR_IO;
}

// Whole file -> memory.  Like the reference's InputFile (statgen/InputFile.cpp: ifopen picks
// GzipFileType by the magic bytes), a gzip'd file is inflated transparently and anything else is
// read as it is -- zlib's gzread does both.
// `pad`: capacity kept free behind the contents (read_pileup appends that many newlines: its AVX2 scanner loads
// whole 32-byte blocks)
bool slurp(const std::string& path, std::string* all, size_t pad = 0)
{
    // plain files (everything but the gzip'd panels the reference's InputFile also accepts) are
    // read in one go; zlib's transparent mode would copy them through its own buffers
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool regular = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        const ssize_t got = regular ? ::pread(fd, magic, 2, 0) : 0;
        if (regular && !(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
            all->reserve((size_t)st.st_size + pad);
            all->resize((size_t)st.st_size);
            size_t done = 0;
            while (done < all->size()) {
                const ssize_t n = ::read(fd, &(*all)[done], all->size() - done);
                if (n < 0) { ::close(fd); return false; }
                if (n == 0) break;                       // (shrank meanwhile)
                done += (size_t)n;
            }
            all->resize(done);
            ::close(fd);
            return true;
        }
        ::close(fd);
    }
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) all->append(buf.data(), (size_t)n);
    const bool ok = n == 0;
    gzclose(f);
    return ok;
}

// ---- fast field scanner -------------------------------------------------------------------
// The readers below restate code that parses every line with `std::stringstream >> field`.
// That is exact but slow (≈2 µs per field).  The scanner handles the lines whose fields are
// plainly formatted -- whitespace-separated tokens, decimal numbers -- with the same results
// (strtod is what libstdc++'s num_get ends in), and reports anything else as "not plain": the
// caller then runs the original stringstream statements on that line, so odd input (short
// lines, "1e", hex, overflow, locale digits, ...) keeps the iostream behaviour bit for bit.
// Tunables::slow_parse = 1: every line takes the stringstream statements -- the
// differential tests compare the two paths on deliberately odd files.
inline bool slow_parse() { return tunables().slow_parse != 0; }

struct WsTable {
    unsigned char ws[256];
    WsTable()
    {
        std::memset(ws, 0, sizeof(ws));
        for (const char* k = " \t\r\v\f\n"; *k; ++k) ws[(unsigned char)*k] = 1;
    }
};
static const WsTable kWs;
inline bool is_ws(char c) { return kWs.ws[(unsigned char)c] != 0; }

struct Scan {
    const char* p;
    const char* end;
    void skip_ws() { while (p < end && is_ws(*p)) ++p; }
    // next whitespace-delimited token; false at end of line
    bool token(const char** b, const char** e)
    {
        skip_ws();
        if (p >= end) return false;
        *b = p;
        while (p < end && !is_ws(*p)) ++p;
        *e = p;
        return true;
    }
    // `ss >> char`: the next non-blank character (tokens do not matter)
    bool one_char(char* c)
    {
        skip_ws();
        if (p >= end) return false;
        *c = *p++;
        return true;
    }
};

// [+-]?digits, at most 9 digits (no overflow question) -> int
inline bool plain_int(const char* b, const char* e, int* out)
{
    const char* q = b;
    bool neg = false;
    if (q < e && (*q == '+' || *q == '-')) neg = (*q++ == '-');
    if (q >= e || e - q > 9) return false;
    int v = 0;
    for (; q < e; ++q) {
        if (*q < '0' || *q > '9') return false;
        v = v * 10 + (*q - '0');
    }
    *out = neg ? -v : v;
    return true;
}

// [+-]?(digits[.digits*] | .digits)([eE][+-]?digits)? -> double.
// Up to 15 significant digits and a decimal exponent within +-22 (what R / awk / printf("%.15g")
// panels contain) take Clinger's exact path: the digit string as an integer (< 2^53) times or
// divided by an exactly representable power of ten is ONE correctly rounded IEEE operation,
// i.e. the strtod result.  Everything else goes to strtod itself.
inline bool plain_double(const char* b, const char* e, double* out)
{
    static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                      1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* q = b;
    bool neg = false;
    if (q < e && (*q == '+' || *q == '-')) neg = (*q++ == '-');
    int nd = 0, nsig = 0, frac = 0;
    unsigned long long sig = 0;
    bool lead = true;
    while (q < e && *q >= '0' && *q <= '9') {
        if (!(lead && *q == '0')) { lead = false; if (nsig < 19) sig = sig * 10 + (unsigned)(*q - '0'); ++nsig; }
        ++q; ++nd;
    }
    if (q < e && *q == '.') {
        ++q;
        while (q < e && *q >= '0' && *q <= '9') {
            if (!(lead && *q == '0')) { lead = false; if (nsig < 19) sig = sig * 10 + (unsigned)(*q - '0'); ++nsig; }
            ++q; ++nd; ++frac;
        }
    }
    if (nd == 0) return false;
    int ex = 0;
    if (q < e && (*q == 'e' || *q == 'E')) {
        ++q;
        bool eneg = false;
        if (q < e && (*q == '+' || *q == '-')) eneg = (*q++ == '-');
        int ne = 0;
        while (q < e && *q >= '0' && *q <= '9') { ex = ex * 10 + (*q - '0'); ++q; ++ne; }
        if (ne == 0 || ne > 3) return false;
        if (eneg) ex = -ex;
    }
    if (q != e || e - b > 60) return false;
    const int e10 = ex - frac;                  // value = sig * 10^e10
    if (nsig <= 15 && e10 >= -22 && e10 <= 22) {
        double v = (double)sig;                 // exact: sig < 10^15 < 2^53
        v = e10 < 0 ? v / kPow10[-e10] : v * kPow10[e10];
        *out = neg ? -v : v;
        return true;
    }
    char tmp[64];
    std::memcpy(tmp, b, (size_t)(e - b));
    tmp[e - b] = 0;
    char* endp = nullptr;
    const double v = std::strtod(tmp, &endp);
    if (endp != tmp + (e - b) || !std::isfinite(v)) return false;
    *out = v;
    return true;
}

// Calls fn(begin, end) for every '\n'-terminated line.  The reference reads panel files
// through statgen's InputFile::readLine, which reports EOF (and so drops the data) for a last
// line that lacks '\n' (statgen/InputFile.cpp:112-130); with_unterminated = true is the
// std::getline behaviour instead (an unterminated last line counts).
template <class F>
void for_each_line_n(const char* p, size_t len, bool with_unterminated, F fn)
{
    const char* end = p + len;
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
        if (!nl) {
            if (with_unterminated) fn(p, end);
            break;
        }
        fn(p, nl);
        p = nl + 1;
    }
}
template <class F>
void for_each_line(const std::string& all, bool with_unterminated, F fn)
{
    const char* p = all.data();
    const char* end = p + all.size();
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
        if (!nl) {
            if (with_unterminated) fn(p, end);
            break;
        }
        fn(p, nl);
        p = nl + 1;
    }
}

}  // namespace

// ContaminationEstimator.cpp:413-438.  ref/alt are single chars: only the first
// character of an allele column is kept ("A,G" -> 'A').
int read_bed(const std::string& path, Panel* p)
{
    std::string all;
    if (!slurp(path, &all)) return io_error("Open file:" + path + "\t failed");
    std::string chr;
    int pos = 0;
    char ref = 0, alt = 0;
    std::unordered_map<int, int32_t>* inner = nullptr;    // slotOf[chr] of the last line
    std::string inner_chr;
    int32_t inner_id = -1;
    const bool slow = slow_parse();
    for_each_line(all, false, [&](const char* b, const char* e) {
        Scan sc{b, e};
        const char *t0, *t1, *u0, *u1, *v0, *v1;
        int p1 = 0, p2 = 0;
        char r = 0, a = 0;
        if (!slow && sc.token(&t0, &t1) && sc.token(&u0, &u1) && plain_int(u0, u1, &p1) && sc.token(&v0, &v1) &&
            plain_int(v0, v1, &p2) && sc.one_char(&r) && sc.one_char(&a)) {
            chr.assign(t0, t1);
            pos = p2;
            ref = r;
            alt = a;
        } else {                                   // not plain: the original statements
            std::stringstream ss(std::string(b, e));
            ss >> chr >> pos >> pos;
            ss >> ref >> alt;
        }
        p->PosVec.push_back(std::make_pair(chr, pos));
        if (!inner || inner_chr != chr) {
            const bool fresh = p->slotOf.find(chr) == p->slotOf.end();
            inner = &p->slotOf[chr];
            inner_chr = chr;
            if (fresh) {
                inner_id = (int32_t)p->chrNames.size();
                p->chrNames.push_back(chr);
            } else {
                inner_id = 0;
                while (p->chrNames[inner_id] != chr) ++inner_id;
            }
        }
        // ChooseBed[chr][pos] = (ref, alt): a repeated position keeps its slot, the later alleles win
        auto ins = inner->emplace(pos, (int32_t)p->slotPos.size());
        const int32_t slot = ins.first->second;
        if (ins.second) {
            p->slotPos.push_back(pos);
            p->slotChr.push_back(inner_id);
            p->slotRef.push_back(ref);
            p->slotAlt.push_back(alt);
        } else {
            p->slotRef[slot] = ref;
            p->slotAlt[slot] = alt;
        }
        p->rowSlot.push_back(slot);
    });
    return VB2_OK;
}

void Panel::finish()
{
    slotAF.assign(num_slot(), 0.0);
    if (!isAFknown) return;
    // the reference's knownAF[chr][pos] yields 0 for a site the AF file does not list
    for (size_t s = 0; s < num_slot(); ++s) {
        auto ac = knownAF.find(chrNames[slotChr[s]]);
        if (ac == knownAF.end()) continue;
        auto ap = ac->second.find((uint32_t)slotPos[s]);
        if (ap != ac->second.end()) slotAF[s] = ap->second;
    }
}

// ContaminationEstimator.cpp:342-373: first numPC columns; fewer is fatal.
int read_ud(const std::string& path, Panel* p)
{
    std::string all;
    if (!slurp(path, &all)) return io_error("Open file:" + path + "\t failed");
    const int numPC = p->numPC;
    const bool slow = slow_parse();
    // The rows are independent (a short row is fatal, so nothing persists from line to line): big
    // files are cut at line ends and parsed by a few threads -- 17-digit values, which need strtod,
    // cost ~90 ns each, 37 ms for a 100 000 x 4 panel on one core.
    struct Part {
        std::vector<double> ud;
        uint32_t rows = 0;
        int rc = VB2_OK;
        std::string err;
    };
    auto parse = [&](const char* from, const char* to, bool with_tail, Part* out) {
        std::vector<double> row(numPC, 0.);
        auto one_line = [&](const char* b, const char* e) {
            if (out->rc) return;
            int index = 0;
            Scan sc{b, e};
            const char *t0, *t1;
            bool plain = !slow;
            while (plain && index < numPC) {
                if (!sc.token(&t0, &t1)) break;                     // short line: the error below
                if (!plain_double(t0, t1, &row[index])) { plain = false; break; }
                index++;
            }
            if (!plain) {
                std::stringstream ss(std::string(b, e));
                index = 0;
                while (index < numPC && ss >> row[index]) index++;
            }
            if (index < numPC) {
                char msg[256];
                std::snprintf(msg, sizeof(msg),
                              "--NumPC should be less than or equal to the number of PCs in SVD files "
                              "provided by --SVDPrefix! (Expected:%d vs Observed:%d)", numPC, index);
                out->err = msg;
                out->rc = VB2_ERR_INVALID;
                return;
            }
            out->ud.insert(out->ud.end(), row.begin(), row.end());
            out->rows++;
        };
        const char* q = from;
        while (q < to) {
            const char* nl = static_cast<const char*>(std::memchr(q, '\n', (size_t)(to - q)));
            if (!nl) {
                (void)with_tail;            // (an unterminated last line is dropped, like for_each_line(.., false, ..))
                break;
            }
            one_line(q, nl);
            q = nl + 1;
        }
    };
    int nthr = all.size() > (1u << 20) ? std::max(1, std::min(8, usable_cpu_count() / 2)) : 1;
    std::vector<Part> parts(nthr);
    if (nthr == 1) {
        parse(all.data(), all.data() + all.size(), true, &parts[0]);
    } else {
        std::vector<const char*> cut(nthr + 1);
        const char* base = all.data();
        const char* end = base + all.size();
        cut[0] = base;
        cut[nthr] = end;
        for (int t = 1; t < nthr; ++t) {
            const char* guess = base + all.size() * (size_t)t / (size_t)nthr;
            if (guess < cut[t - 1]) guess = cut[t - 1];
            const char* nl = static_cast<const char*>(std::memchr(guess, '\n', (size_t)(end - guess)));
            cut[t] = nl ? nl + 1 : end;
        }
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(parse, cut[t], cut[t + 1], t == nthr - 1, &parts[t]);
        for (auto& x : th) x.join();
    }
    for (Part& part : parts) {                       // in file order: the first bad row decides
        if (part.rc) {
            p->UD.insert(p->UD.end(), part.ud.begin(), part.ud.end());
            p->NumMarker += part.rows;
            set_error(part.err);
            return part.rc;
        }
        p->UD.insert(p->UD.end(), part.ud.begin(), part.ud.end());
        p->NumMarker += part.rows;
    }
    return VB2_OK;
}

// ContaminationEstimator.cpp:440-459: second column.
int read_mean(const std::string& path, Panel* p)
{
    std::string all;
    if (!slurp(path, &all)) return io_error("Open file:" + path + "\t failed");
    double mu = 0;
    std::string name;
    const bool slow = slow_parse();
    for_each_line(all, false, [&](const char* b, const char* e) {
        Scan sc{b, e};
        const char *t0, *t1, *u0, *u1;
        double v = 0;
        if (!slow && sc.token(&t0, &t1) && sc.token(&u0, &u1) && plain_double(u0, u1, &v)) {
            mu = v;                                 // (the name is not used)
        } else {
            std::stringstream ss(std::string(b, e));
            ss >> name;
            ss >> mu;
        }
        p->means.push_back(mu);
    });
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
// Returns false where the reference's std::stoi throws (an indel marker without a length, or a
// length that does not fit an int): the reference dies on that line, this reader reports it.
// Core on character ranges: ps / pq must have room for n characters; *out = pairs kept.
bool parse_bases_raw(const char* seq, size_t n, const char* qual, size_t nq, char* ps, char* pq, size_t* out)
{
    // character classes: 1 keep (one quality each), 2 '*' '#' (consume a quality), 3 '+' '-'
    // (indel: skip the length and that many characters), 4 '^' (skip the next character)
    static const struct Table {
        unsigned char cls[256];
        Table()
        {
            std::memset(cls, 0, sizeof(cls));
            for (const char* k = ".,AGCTNagctn"; *k; ++k) cls[(unsigned char)*k] = 1;
            cls[(unsigned char)'*'] = cls[(unsigned char)'#'] = 2;
            cls[(unsigned char)'+'] = cls[(unsigned char)'-'] = 3;
            cls[(unsigned char)'^'] = 4;
        }
    } table;
    size_t iq = 0, o = 0;
    for (size_t i = 0; i < n; ++i) {
        const char c = seq[i];
        switch (table.cls[(unsigned char)c]) {
        case 1:
            ps[o] = c;
            pq[o] = iq < nq ? qual[iq] : '!';
            ++o;
            ++iq;
            break;
        case 2:
            ++iq;
            break;
        case 3: {
            size_t j = i + 1;
            while (j != n && std::isdigit((unsigned char)seq[j])) j++;
            const size_t digit_len = j - (i + 1);
            if (digit_len == 0) return false;                  // stoi(""): invalid_argument
            long long clip = 0;
            for (size_t d = i + 1; d < j; ++d) {
                clip = clip * 10 + (seq[d] - '0');
                if (clip > 2147483647ll) return false;         // stoi: out_of_range
            }
            i += digit_len + (size_t)clip;
            break;
        }
        case 4:
            i += 1;
            break;
        default:
            break;
        }
    }
    *out = o;
    return true;
}

bool parse_bases(const std::string& seq, const std::string& qual, std::string* pseq,
                 std::string* pqual)
{
    const size_t n = seq.size();
    pseq->resize(n);
    pqual->resize(n);
    size_t o = 0;
    if (!parse_bases_raw(seq.data(), n, qual.data(), qual.size(), n ? &(*pseq)[0] : nullptr, n ? &(*pqual)[0] : nullptr, &o))
        return false;
    pseq->resize(o);
    pqual->resize(o);
    return true;
}

#if defined(__x86_64__)
// ---- the same with AVX2 (run-time dispatch; VB2_SCALAR_PARSE=1 keeps the scalar statements) ------------------
// A pileup is ~80 % bases and qualities.  The kept characters (". , A C G T N a c g t n") come in long runs, so the
// bases column is classified 32 characters at a time (two nibble look-ups) and every run of kept characters is
// copied as a block together with its qualities; a character of any other class goes through the scalar
// statements above, one at a time.  Loads may run up to 31 bytes past a field: the caller pads its buffer.
inline bool cpu_has_avx2()
{
    static const bool hw = __builtin_cpu_supports("avx2");
    return hw && tunables().scalar_parse == 0;                  // (the differential tests switch it)
}

__attribute__((target("avx2"))) inline unsigned nonkeep_mask32(__m256i x)
{
    // keep iff (lo_tab[low nibble] & hi_tab[high nibble]) != 0; classes: bit0 0x2_, bit1 0x4_ / 0x6_, bit2 0x5_ / 0x7_
    const __m256i lo_tab = _mm256_setr_epi8(0, 2, 0, 2, 4, 0, 0, 2, 0, 0, 0, 0, 1, 0, 3, 0,
                                            0, 2, 0, 2, 4, 0, 0, 2, 0, 0, 0, 0, 1, 0, 3, 0);
    const __m256i hi_tab = _mm256_setr_epi8(0, 0, 1, 0, 2, 4, 2, 4, 0, 0, 0, 0, 0, 0, 0, 0,
                                            0, 0, 1, 0, 2, 4, 2, 4, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i nib = _mm256_set1_epi8(0x0f);
    const __m256i lo = _mm256_shuffle_epi8(lo_tab, _mm256_and_si256(x, nib));
    const __m256i hi = _mm256_shuffle_epi8(hi_tab, _mm256_and_si256(_mm256_srli_epi16(x, 4), nib));
    const __m256i k = _mm256_and_si256(lo, hi);
    return (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(k, _mm256_setzero_si256()));
}

// bytes <= 0x20 (every whitespace character of the "C" locale is one) among the 32 at p
__attribute__((target("avx2"))) inline unsigned blank_mask32(const char* p)
{
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
    const __m256i sp = _mm256_set1_epi8(0x20);
    return (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(x, sp), sp));
}

// parse_bases_raw's result; ps / pq need room for n + 32 characters, seq / qual 31 readable bytes past their ends
__attribute__((target("avx2")))
bool parse_bases_avx2(const char* seq, size_t n, const char* qual, size_t nq, char* ps, char* pq, size_t* out)
{
    size_t i = 0, iq = 0, o = 0;
    while (i < n) {
        const size_t rem = n - i;
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(seq + i));
        unsigned nk = nonkeep_mask32(x);
        if (rem < 32) nk |= ~0u << rem;
        const unsigned run = nk ? (unsigned)__builtin_ctz(nk) : 32u;
        if (run) {
            _mm256_storeu_si256(reinterpret_cast<__m256i*>(ps + o), x);
            if (iq + run <= nq) {
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(pq + o),
                                    _mm256_loadu_si256(reinterpret_cast<const __m256i*>(qual + iq)));
            } else {                                           // fewer qualities than bases: '!' (the scalar rule)
                const size_t have = iq < nq ? nq - iq : 0;
                std::memcpy(pq + o, qual + iq, have);
                std::memset(pq + o + have, '!', run - have);
            }
            o += run;
            i += run;
            iq += run;
        }
        if (run < 32 && i < n) {                               // one character of another class
            const char c = seq[i];
            if (c == '*' || c == '#') {
                ++iq;
            } else if (c == '+' || c == '-') {
                size_t j = i + 1;
                while (j != n && std::isdigit((unsigned char)seq[j])) j++;
                const size_t digit_len = j - (i + 1);
                if (digit_len == 0) return false;              // stoi(""): invalid_argument
                long long clip = 0;
                for (size_t d = i + 1; d < j; ++d) {
                    clip = clip * 10 + (seq[d] - '0');
                    if (clip > 2147483647ll) return false;     // stoi: out_of_range
                }
                i += digit_len + (size_t)clip;
            } else if (c == '^') {
                i += 1;
            }
            ++i;
        }
    }
    *out = o;
    return true;
}

// A line of exactly six tab-separated, non-empty fields without any other blank: their bounds.  False: not such
// a line (the caller's general statements decide).  31 readable bytes past `e` required.
__attribute__((target("avx2")))
bool split_six_fields(const char* b, const char* e, const char** f0, const char** f1)
{
    const char* p = b;
    for (int k = 0; k < 4; ++k) {                              // chromosome, position, reference base, depth: short
        f0[k] = p;
        while (p < e && (unsigned char)*p > 0x20) ++p;
        f1[k] = p;
        if (p >= e || *p != '\t' || f1[k] == f0[k]) return false;
        ++p;
    }
    f0[4] = p;                                                 // bases: up to the next blank, which must be a tab
    for (;;) {
        if (p >= e) return false;
        const unsigned m = blank_mask32(p);
        if (m) { p += __builtin_ctz(m); break; }
        p += 32;
    }
    if (p >= e || *p != '\t' || p == f0[4]) return false;
    f1[4] = p;
    ++p;
    f0[5] = p;                                                 // qualities: the rest of the line, no blank in it
    if (p >= e) return false;
    for (const char* q = p; q < e; q += 32) {
        unsigned m = blank_mask32(q);
        const size_t rem = (size_t)(e - q);
        if (rem < 32) m &= ~(~0u << rem);
        if (m) return false;
    }
    f1[5] = e;
    return true;
}
#endif
}  // namespace

// SimplePileupViewer.cpp:748-833.
//
// The reference reads every line with `ss >> chr >> pos >> ref >> depth >> seq >> qual` into
// variables that live across lines, so a short or malformed line inherits the fields of the line
// before it (after that line's own processing: parsed strings for a line outside the .bed, empty
// strings and depth 0 for a line inside it).  Plainly formatted lines -- all of them, normally --
// take the scanner below and never touch those variables; the state a following malformed line
// would inherit is rebuilt from the remembered line only when such a line turns up.
int read_pileup(const std::string& path, const Panel& panel, PileupViewer* v)
{
    std::string all;
    constexpr size_t kPad = 64;
    if (!slurp(path, &all, kPad)) return io_error("open file " + path + " failed!");
    const size_t all_len = all.size();
    all.append(kPad, '\n');                                 // (lines are cut from the first all_len bytes only)
    v->init(panel);
    v->basePool.reserve(all_len / 3);
    v->qualPool.reserve(all_len / 3);
    v->siteOff.reserve(panel.num_slot() + 1);
    std::string chr, ref, seq, qual, pseq, pqual;          // the reference's persistent variables
    int pos = 0, depth = 0;
    v->numBases = 0;
    int rc = VB2_OK;
    // the last plainly formatted line, not yet folded into the variables above
    struct Last {
        bool pending = false, in_bed = false;
        const char *c0, *c1, *r0, *r1, *s0, *s1, *q0, *q1;
        int pos;
    } last;
    auto materialise = [&]() {
        if (!last.pending) return;
        last.pending = false;
        chr.assign(last.c0, last.c1);
        pos = last.pos;
        ref.assign(last.r0, last.r1);
        if (last.in_bed) {                                  // cpp:825-827
            depth = 0;
            seq.clear();
            qual.clear();
        } else {                                            // the parsed strings persist (:785-786)
            seq.assign(last.s0, last.s1);
            qual.assign(last.q0, last.q1);
            parse_bases(seq, qual, &pseq, &pqual);
            seq = pseq;
            qual = pqual;
            depth = (int)pqual.length();
        }
    };
    // per-chromosome table of the last line (consecutive lines share the chromosome); the next
    // line usually is the next slot of the .bed, which needs no table at all
    const std::unordered_map<int, int32_t>* tab = nullptr;
    const char* tab_c0 = nullptr;
    size_t tab_len = 0;
    int32_t next_slot = 0;
    const int32_t nslot = (int32_t)panel.num_slot();
    auto slot_of = [&](const char* c0, const char* c1, int p) -> int32_t {
        const size_t len = (size_t)(c1 - c0);
        if (next_slot < nslot && panel.slotPos[next_slot] == p) {
            const std::string& name = panel.chrNames[panel.slotChr[next_slot]];
            if (name.size() == len && std::memcmp(name.data(), c0, len) == 0) return next_slot;
        }
        if (!tab_c0 || tab_len != len || std::memcmp(tab_c0, c0, len) != 0) {
            auto it = panel.slotOf.find(std::string(c0, c1));
            tab = it == panel.slotOf.end() ? nullptr : &it->second;
            tab_c0 = c0;
            tab_len = len;
        }
        if (!tab) return -1;
        auto s = tab->find(p);
        return s == tab->end() ? -1 : s->second;
    };
    auto record = [&](int32_t slot, const char* b, const char* q, size_t n) {
        if (v->siteOfSlot[slot] >= 0) {
            std::cerr << "[WARNING] The pileup file has duplicated lines! Merged here" << std::endl;
            // the reference builds a merged copy and then drops it (quirk vii)
        } else {
            v->add_site(slot, b, q, n);
        }
        v->numBases += (int)n;
        v->effectiveNumSite++;
        next_slot = slot + 1;
    };
    const bool slow = slow_parse();
#if defined(__x86_64__)
    // The AVX2 path writes the parsed characters straight behind the pools' contents, a block at a time: the pools
    // are kept at a generous size while it is in use (`used` = the real length), and cut back for the statements
    // that go through the string interface.
    const bool simd = !slow && cpu_has_avx2();
    size_t used = 0, cap = 0;
    auto pools_up = [&](size_t need) {                      // room for `need` more characters (+ a block of slack)
        if (used + need + 32 <= cap) return;
        cap = std::max(used + need + 32, std::max<size_t>(cap * 2, all_len / 2 + 4096));
        v->basePool.resize(cap);
        v->qualPool.resize(cap);
    };
    auto pools_down = [&]() {                               // back to the string interface
        if (cap == 0) return;
        v->basePool.resize(used);
        v->qualPool.resize(used);
        cap = 0;
    };
#endif
    const char* const all_b = all.data();
    for_each_line_n(all_b, all_len, true, [&](const char* b, const char* e) {
        if (rc) return;
#if defined(__x86_64__)
        const char *g0[6], *g1[6];
        int ppos_s = 0, pdepth_s = 0;
        if (simd && split_six_fields(b, e, g0, g1) && plain_int(g0[1], g1[1], &ppos_s) && plain_int(g0[3], g1[3], &pdepth_s)) {
            const char *s0 = g0[4], *s1 = g1[4], *q0 = g0[5], *q1 = g1[5];
            const size_t n = (size_t)(s1 - s0);
            if (g1[2] - g0[2] == 1 && *g0[2] == '.' && (std::memchr(s0, '.', n) || std::memchr(s0, ',', n))) {
                set_error("Pileup format error: cannot find ref allele, exit!");
                rc = VB2_ERR_INVALID;
                return;
            }
            if (cap == 0) used = v->basePool.size();
            if (used + n > 0xffffffffull) {                 // (site offsets are 32-bit: 4 GiB of kept bases per sample)
                set_error("pileup too large: more than 4 GiB of bases at the panel's sites");
                rc = VB2_ERR_INVALID;
                return;
            }
            pools_up(n);
            size_t kept = 0;
            if (!parse_bases_avx2(s0, n, q0, (size_t)(q1 - q0), &v->basePool[used], &v->qualPool[used], &kept)) {
                set_error("Pileup format error: indel marker without a valid length in the bases column of " +
                          std::string(g0[0], g1[0]) + ":" + std::to_string(ppos_s));
                rc = VB2_ERR_INVALID;
                return;
            }
            const int32_t slot = slot_of(g0[0], g1[0], ppos_s);
            last.pending = true;
            last.in_bed = slot >= 0;
            last.c0 = g0[0]; last.c1 = g1[0]; last.r0 = g0[2]; last.r1 = g1[2]; last.s0 = s0; last.s1 = s1; last.q0 = q0; last.q1 = q1;
            last.pos = ppos_s;
            if (slot >= 0) {
                if (v->siteOfSlot[slot] < 0) {              // a new site: the parsed characters stay
                    v->siteOfSlot[slot] = v->num_site();
                    used += kept;
                    v->siteOff.push_back((uint32_t)used);
                    v->numBases += (int)kept;
                    v->effectiveNumSite++;
                    next_slot = slot + 1;
                } else {
                    record(slot, nullptr, nullptr, kept);   // a duplicated line: warning, counters (quirk vii)
                }
            }
            return;
        }
        pools_down();
#endif
        Scan sc{b, e};
        const char *c0, *c1, *p0, *p1, *r0, *r1, *d0, *d1, *s0, *s1, *q0, *q1;
        int ppos = 0, pdepth = 0;
        if (!slow && sc.token(&c0, &c1) && sc.token(&p0, &p1) && plain_int(p0, p1, &ppos) && sc.token(&r0, &r1) &&
            sc.token(&d0, &d1) && plain_int(d0, d1, &pdepth) && sc.token(&s0, &s1) && sc.token(&q0, &q1)) {
            const size_t n = (size_t)(s1 - s0);
            if (r1 - r0 == 1 && *r0 == '.' && (std::memchr(s0, '.', n) || std::memchr(s0, ',', n))) {
                set_error("Pileup format error: cannot find ref allele, exit!");
                rc = VB2_ERR_INVALID;
                return;
            }
            // parse straight into the pools' tails; the bytes stay only if the line is a new site
            const size_t tail = v->basePool.size();
            if (tail + n > 0xffffffffull) {                 // (site offsets are 32-bit: 4 GiB of kept bases per sample)
                set_error("pileup too large: more than 4 GiB of bases at the panel's sites");
                rc = VB2_ERR_INVALID;
                return;
            }
            v->basePool.resize(tail + n);
            v->qualPool.resize(tail + n);
            size_t kept = 0;
            if (!parse_bases_raw(s0, n, q0, (size_t)(q1 - q0), &v->basePool[tail], &v->qualPool[tail], &kept)) {
                set_error("Pileup format error: indel marker without a valid length in the bases column of " +
                          std::string(c0, c1) + ":" + std::to_string(ppos));
                rc = VB2_ERR_INVALID;
                return;
            }
            const int32_t slot = slot_of(c0, c1, ppos);
            last.pending = true;
            last.in_bed = slot >= 0;
            last.c0 = c0; last.c1 = c1; last.r0 = r0; last.r1 = r1; last.s0 = s0; last.s1 = s1; last.q0 = q0; last.q1 = q1;
            last.pos = ppos;
            const bool fresh = slot >= 0 && v->siteOfSlot[slot] < 0;
            v->basePool.resize(tail + (fresh ? kept : 0));
            v->qualPool.resize(tail + (fresh ? kept : 0));
            if (slot >= 0) {
                if (fresh) {
                    v->siteOfSlot[slot] = v->num_site();
                    v->siteOff.push_back((uint32_t)v->basePool.size());
                    v->numBases += (int)kept;
                    v->effectiveNumSite++;
                    next_slot = slot + 1;
                } else {
                    record(slot, nullptr, nullptr, kept);       // a duplicated line: warning, counters (quirk vii)
                }
            }
            return;
        }
        // any other line: the original statements, on the variables as the previous lines left them
        materialise();
        {
            std::stringstream ss(std::string(b, e));
            ss >> chr >> pos >> ref >> depth >> seq >> qual;
        }
        if (seq.find_first_of(".,") != std::string::npos && ref == ".") {
            set_error("Pileup format error: cannot find ref allele, exit!");
            rc = VB2_ERR_INVALID;
            return;
        }
        if (!parse_bases(seq, qual, &pseq, &pqual)) {
            set_error("Pileup format error: indel marker without a valid length in the bases column of " +
                      chr + ":" + std::to_string(pos));
            rc = VB2_ERR_INVALID;
            return;
        }
        seq = pseq;                                         // the parsed strings are what persists
        qual = pqual;                                       // into a following short line (:785-786)
        depth = (int)pqual.length();                        // SNP bases only
        const int32_t slot = slot_of(chr.data(), chr.data() + chr.size(), pos);
        tab_c0 = nullptr;                                   // (chr's buffer is not part of `all`)
        if (slot < 0) return;
        record(slot, pseq.data(), pqual.data(), pqual.size());
        depth = 0;
        seq = "";
        qual = "";
    });
#if defined(__x86_64__)
    pools_down();
#endif
    if (rc) return rc;
    v->avgDepth = (double)v->numBases / v->effectiveNumSite;
    return VB2_OK;
}

// ContaminationEstimator.cpp:543-587.
bool sanity_check(const Panel& p, PileupViewer* v)
{
    std::fprintf(stderr, "NOTICE - Number of marker in Reference Matrix:%d\n", (int)p.NumMarker);
    std::fprintf(stderr, "NOTICE - Number of marker shared with input file:%d\n", v->effectiveNumSite);
    auto depth_at = [&](size_t i, int* d) {
        if (i >= p.rowSlot.size()) return false;
        const int32_t site = v->site_of(p.rowSlot[i]);
        if (site < 0) return false;
        *d = (int)v->depth(site);
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
    for (size_t i = 0; i < f.panel.PosVec.size(); ++i) {
        const auto& item = f.panel.PosVec[i];
        const int32_t slot = f.panel.rowSlot[i], site = f.viewer.site_of(slot);
        if (site < 0) continue;
        const uint32_t n = f.viewer.depth(site);
        if (n == 0) continue;
        fout << item.first << "\t" << item.second << "\t" << f.panel.slotRef[slot] << "\t" << n << "\t";
        fout.write(f.viewer.bases(site), n);
        fout << "\t";
        fout.write(f.viewer.quals(site), n);
        fout << std::endl;
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
    bases.reserve(viewer.basePool.size());
    quals.reserve(viewer.qualPool.size());
    for (uint32_t i = 0; i < M; ++i) {
        read_off[i] = (int64_t)bases.size();
        if (i >= panel.rowSlot.size()) continue;
        const int32_t slot = panel.rowSlot[i], site = viewer.site_of(slot);
        if (site < 0) continue;
        ++num_site;
        bases.append(viewer.bases(site), viewer.depth(site));
        quals.append(viewer.quals(site), viewer.depth(site));
        alt_base[i] = panel.slotAlt[slot];
        if (panel.isAFknown && (size_t)slot < panel.slotAF.size()) known_af[i] = panel.slotAF[slot];
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

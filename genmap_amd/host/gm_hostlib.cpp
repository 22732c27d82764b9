// gm_hostlib.cpp -- FASTA ingestion, index directory and output writers of the `genmap` host program.
// See gm_hostlib.h for the byte-compatibility targets (file:line in /root/reference).
#include "gm_hostlib.h"
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <set>
#include <sstream>
#include <thread>
#include <sys/stat.h>

namespace gmh {

// ---- FASTA ------------------------------------------------------------------------------------------------
static inline uint8_t code_of(unsigned char ch)
{
    switch (ch) {   // Dna5 conversion; anything that is not A,C,G,T/U becomes N (src/indexing.hpp:13-20)
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

bool read_fasta(const std::string& path, std::vector<FastaRecord>& out, std::string& err)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::vector<FastaRecord> recs;
    std::string line;
    bool have = false;
    const bool fastq = path.size() > 6 && path.compare(path.size() - 6, 6, ".fastq") == 0;
    if (fastq) {   // 4-line records
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            FastaRecord r; r.id = line.substr(1);
            if (!r.id.empty() && r.id.back() == '\r') r.id.pop_back();
            std::string seq, plus, qual;
            std::getline(in, seq); std::getline(in, plus); std::getline(in, qual);
            for (unsigned char ch : seq) if (!std::isspace(ch)) r.codes.push_back(code_of(ch));
            recs.push_back(std::move(r));
        }
    } else {
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (!line.empty() && line[0] == '>') {
                recs.emplace_back();
                recs.back().id = line.substr(1);
                have = true;
            } else if (have) {
                for (unsigned char ch : line) if (!std::isspace(ch)) recs.back().codes.push_back(code_of(ch));
            }
        }
    }
    // skip empty sequences (src/indexing.hpp:228-231)
    std::vector<FastaRecord> kept;
    for (auto& r : recs) if (!r.codes.empty()) kept.push_back(std::move(r));
    // if shortened ids are still unique, use them instead (src/indexing.hpp:238-266)
    std::vector<std::string> shortIds;
    for (auto& r : kept) {
        size_t w = 0;
        while (w < r.id.size() && !std::isspace((unsigned char)r.id[w])) ++w;
        shortIds.push_back(r.id.substr(0, w));
    }
    std::set<std::string> uniq(shortIds.begin(), shortIds.end());
    if (uniq.size() == shortIds.size())
        for (size_t i = 0; i < kept.size(); ++i) kept[i].id = shortIds[i];
    out = std::move(kept);
    return true;
}

static void scan_dir(const std::string& path, std::vector<std::pair<std::string, std::string>>& files)
{
    static const char* kTypes[] = {"fsa", "fna", "fastq", "fasta", "fas", "faa", "fa"};   // src/indexing.hpp:290
    DIR* d = opendir(path.c_str());
    if (!d) return;
    while (dirent* e = readdir(d)) {
        std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        std::string full = path + "/" + name;
        struct stat st;
        if (stat(full.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) { scan_dir(full, files); continue; }
        size_t dot = name.find_last_of('.');
        std::string ext = dot == std::string::npos ? name : name.substr(dot + 1);
        for (const char* t : kTypes) if (ext == t) { files.push_back({path + "/", name}); break; }
    }
    closedir(d);
}

bool list_fasta_directory(const std::string& dir, std::vector<std::pair<std::string, std::string>>& out, std::string& err)
{
    out.clear();
    scan_dir(dir, out);
    std::sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.second < b.second; });   // src/indexing.hpp:407
    for (size_t i = 0; i + 1 < out.size(); ++i)
        if (out[i].second == out[i + 1].second) {
            err = "ERROR: At least two fasta files with the same filename found (this is not supported)! Please rename them and run again.\n       " +
                  out[i].first + out[i].second + "!\n       " + out[i + 1].first + out[i + 1].second + "!\n";
            return false;
        }
    return true;
}

// ---- index directory --------------------------------------------------------------------------------------
template <class F> static void parallel_ranges(uint64_t n, F f)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::min<uint64_t>(std::min(hw, 64u), std::max<uint64_t>(1, n >> 20));
    if (T <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back(f, n * t / T, n * (t + 1) / T);
    for (auto& x : th) x.join();
}
static bool write_packed4(const std::string& path, const std::vector<uint8_t>& v, std::string& err)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) { err = "cannot write " + path; return false; }
    uint64_t n = v.size();
    f.write((const char*)&n, 8);
    std::vector<uint8_t> buf((n + 1) / 2);
    parallel_ranges(buf.size(), [&](uint64_t b, uint64_t e) {   // a 3.1 Gbp index packs three such streams: all host cores
        for (uint64_t j = b; j < e; ++j) buf[j] = (uint8_t)((v[2 * j] & 15u) | (2 * j + 1 < n ? (v[2 * j + 1] & 15u) << 4 : 0u));
    });
    f.write((const char*)buf.data(), (std::streamsize)buf.size());
    return (bool)f;
}
static bool read_packed4(const std::string& path, std::vector<uint8_t>& v, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot read " + path; return false; }
    uint64_t n = 0;
    f.read((char*)&n, 8);
    std::vector<uint8_t> buf((n + 1) / 2);
    f.read((char*)buf.data(), (std::streamsize)buf.size());
    if (!f) { err = "truncated " + path; return false; }
    v.resize(n);
    parallel_ranges(buf.size(), [&](uint64_t b, uint64_t e) {
        for (uint64_t j = b; j < e; ++j) { v[2 * j] = buf[j] & 15u; if (2 * j + 1 < n) v[2 * j + 1] = buf[j] >> 4; }
    });
    return true;
}

bool write_index_dir(const std::string& dir, const IndexMeta& m, const std::vector<uint8_t>& text, const std::vector<uint8_t>& bf,
                     const std::vector<uint8_t>& br, const SaFiles& sa, std::string& err)
{
    const std::string p = dir + (dir.empty() || dir.back() == '/' ? "" : "/") + "index";
    {
        std::ofstream f(p + ".info");   // keys of src/indexing.hpp:102-112
        if (!f) { err = "cannot write " + p + ".info"; return false; }
        f << "alphabet_size:" << m.alphabetSize << '\n' << "sa_dimensions_i1:" << m.seqNoBits << '\n' << "sa_dimensions_i2:" << m.seqPosBits << '\n'
          << "bwt_dimensions:" << m.bwtBits << '\n' << "sampling_rate:" << m.sampling << '\n' << "fasta_directory:" << (m.directory ? "true" : "false") << '\n'
          << "packed_text:true\n";
    }
    {
        std::ofstream f(p + ".ids");    // rows of src/indexing.hpp:268-274
        if (!f) { err = "cannot write " + p + ".ids"; return false; }
        for (auto& r : m.ids) f << r.file << ';' << r.length << ';' << r.name << '\n';
    }
    if (!write_packed4(p + ".txt4", text, err) || !write_packed4(p + ".bwt4", bf, err) || !write_packed4(p + ".rev.bwt4", br, err)) return false;
    auto put = [&](const std::string& name, const std::vector<uint32_t>& v) {
        std::ofstream f(name, std::ios::binary);
        if (f) f.write((const char*)v.data(), (std::streamsize)(v.size() * 4));
        if (!f) { err = "cannot write " + name; return false; }
        return true;
    };
    if (!sa.full.empty() && !put(p + ".sa", sa.full)) return false;
    if (!sa.marks.empty() && (!put(p + ".sa.marks", sa.marks) || !put(p + ".sa.samples", sa.samples))) return false;
    return true;
}

bool read_index_dir(const std::string& dir, IndexMeta& m, std::vector<uint8_t>& text, std::vector<uint8_t>& bf, std::vector<uint8_t>& br,
                    SaFiles& sa, std::string& err)
{
    const std::string p = dir + (dir.empty() || dir.back() == '/' ? "" : "/") + "index";
    std::ifstream fi(p + ".info");
    if (!fi) { err = "cannot read " + p + ".info"; return false; }
    std::map<std::string, std::string> kv;
    std::string line;
    while (std::getline(fi, line)) { size_t c = line.find(':'); if (c != std::string::npos) kv[line.substr(0, c)] = line.substr(c + 1); }
    auto need = [&](const char* k, std::string& v) {   // retrieve(): src/mappability.hpp:49-66
        auto it = kv.find(k);
        if (it == kv.end()) { err = std::string("ERROR: Malformed index.info file! Could not find key '") + k + "'.\n"; return false; }
        v = it->second; return true;
    };
    std::string v;
    if (!need("alphabet_size", v)) return false;
    try {
    m.alphabetSize = (uint32_t)std::stoi(v);
    if (!need("sa_dimensions_i1", v)) return false;
    m.seqNoBits = (uint32_t)std::stoi(v);
    if (!need("sa_dimensions_i2", v)) return false;
    m.seqPosBits = (uint32_t)std::stoi(v);
    if (!need("bwt_dimensions", v)) return false;
    m.bwtBits = (uint32_t)std::stoi(v);
    if (!need("sampling_rate", v)) return false;
    m.sampling = (uint32_t)std::stoi(v);
    if (!need("fasta_directory", v)) return false;
    m.directory = (v == "true");
    std::ifstream fd(p + ".ids");
    if (!fd) { err = "cannot read " + p + ".ids"; return false; }
    m.ids.clear();
    while (std::getline(fd, line)) {
        if (line.empty()) continue;
        size_t a = line.find(';'), b = line.find(';', a + 1);   // retrieveDirectoryInformationLine: src/common.hpp:10-19
        if (a == std::string::npos || b == std::string::npos) { err = "malformed row in " + p + ".ids"; return false; }
        m.ids.push_back({line.substr(0, a), (uint64_t)std::stoull(line.substr(a + 1, b - a - 1)), line.substr(b + 1)});
    }
    } catch (const std::exception&) { err = "ERROR: Malformed index.info / index.ids file (a number was expected).\n"; return false; }
    if (!read_packed4(p + ".txt4", text, err) || !read_packed4(p + ".bwt4", bf, err) || !read_packed4(p + ".rev.bwt4", br, err)) return false;
    sa.full.clear(); sa.marks.clear(); sa.samples.clear();
    auto get = [&](const std::string& name, std::vector<uint32_t>& v, uint64_t count, bool wholeFile) {   // false: absent or truncated (err set)
        std::ifstream f(name, std::ios::binary);
        if (!f) return false;
        if (wholeFile) { f.seekg(0, std::ios::end); count = (uint64_t)f.tellg() / 4; f.seekg(0); }
        v.resize(count);
        f.read((char*)v.data(), (std::streamsize)(count * 4));
        if (!f) { err = "truncated " + name; v.clear(); return false; }
        return true;
    };
    if (m.sampling <= 1) {   // one entry per row: uint32, or uint64 for indexes of 2^32 - 1 rows and more
        if (!get(p + ".sa", sa.full, 0, true)) { if (!err.empty()) return false; }
        else if (sa.full.size() != bf.size() && sa.full.size() != 2 * bf.size()) { err = "truncated " + p + ".sa"; return false; }
    }
    else {
        const bool a = get(p + ".sa.marks", sa.marks, (bf.size() + 31) / 32, false);
        if (!a && !err.empty()) return false;
        if (a && !get(p + ".sa.samples", sa.samples, 0, true)) { if (err.empty()) err = "cannot read " + p + ".sa.samples"; return false; }
    }
    return true;
}

// ---- writers ----------------------------------------------------------------------------------------------
static inline uint32_t val_at(const void* c, int width, uint64_t i) { return width == 1 ? ((const uint8_t*)c)[i] : ((const uint16_t*)c)[i]; }

// operator<<(ostream&, float) with default flags == printf("%g")
static inline int fmt_float(char* buf, float f) { return snprintf(buf, 32, "%g", (double)f); }
static inline float inverse_of(uint32_t v) { return v != 0 ? 1.0f / static_cast<float>(v) : 0.0f; }

class BufferedFile {
public:
    explicit BufferedFile(const std::string& path, bool append = false) { f_ = fopen(path.c_str(), append ? "ab" : "wb"); if (f_) setvbuf(f_, nullptr, _IOFBF, 1 << 20); }
    ~BufferedFile() { if (f_) fclose(f_); }
    bool ok() const { return f_ != nullptr; }
    void put(char ch) { fputc(ch, f_); }
    void put(const std::string& s) { fwrite(s.data(), 1, s.size(), f_); }
    void put(const char* s, size_t n) { fwrite(s, 1, n, f_); }
    void put_u64(uint64_t v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v); fwrite(b, 1, (size_t)n, f_); }
    void put_float(float f) { char b[32]; int n = fmt_float(b, f); fwrite(b, 1, (size_t)n, f_); }
    void put_value(uint32_t v, bool mappability) { if (mappability) put_float(inverse_of(v)); else put_u64(v); }
private:
    FILE* f_ = nullptr;
};

bool save_raw(const void* c, uint64_t n, int width, const std::string& stem, ValueKind kind, std::string& err)
{
    const char* ext = kind == ValueKind::Mappability ? ".map" : kind == ValueKind::Freq8 ? ".freq8" : ".freq16";   // src/mappability.hpp:104-112
    BufferedFile f(stem + ext);
    if (!f.ok()) { err = "cannot write " + stem + ext; return false; }
    if (kind == ValueKind::Mappability) {   // float32 of 1/v, 0 stays 0 (src/output.hpp:17-24); converted on all host cores
        std::vector<float> buf((size_t)std::min<uint64_t>(n, 64ull << 20));
        for (uint64_t i = 0; i < n; i += buf.size()) {
            const uint64_t m = std::min<uint64_t>(buf.size(), n - i);
            parallel_ranges(m, [&](uint64_t b, uint64_t e) { for (uint64_t j = b; j < e; ++j) buf[j] = inverse_of(val_at(c, width, i + j)); });
            f.put((const char*)buf.data(), m * sizeof(float));
        }
    } else {
        f.put((const char*)c, n * (uint64_t)width);
    }
    return true;
}

bool save_txt(const void* c, uint64_t n, int width, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err)
{
    BufferedFile f(stem + ".txt");
    if (!f.ok()) { err = "cannot write " + stem + ".txt"; return false; }
    // text of every value that occurs (operator<< of a float == "%g"), then the positions of a sequence in batches formatted by
    // all host cores: "v v v ... v\n" (src/output.hpp:43-69)
    std::vector<std::string> valueText(width == 1 ? 256 : 65536);
    {
        std::vector<uint8_t> used(valueText.size(), 0);
        parallel_ranges(n, [&](uint64_t b, uint64_t e) { for (uint64_t j = b; j < e; ++j) used[val_at(c, width, j)] = 1; });   // (racing writers store the same 1)
        for (uint32_t v = 0; v < valueText.size(); ++v) if (used[v]) { char t[32]; int k = mappability ? fmt_float(t, inverse_of(v)) : snprintf(t, sizeof t, "%u", v); valueText[v].assign(t, (size_t)k); }
    }
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    std::vector<std::string> buf(T);
    uint64_t pos = 0;
    for (size_t s = 0; s < seqs.lengths.size(); ++s) {
        f.put('>'); f.put(seqs.names[s]); f.put('\n');
        const uint64_t len = seqs.lengths[s], BATCH = 64ull << 20;
        for (uint64_t k0 = 0; k0 < len; k0 += BATCH) {
            const uint64_t k1 = std::min(len, k0 + BATCH), per = (k1 - k0 + T - 1) / T;
            auto work = [&](unsigned t) {
                std::string& o = buf[t]; o.clear();
                const uint64_t a0 = std::min(k1, k0 + t * per), a1 = std::min(k1, a0 + per);
                o.reserve((size_t)(a1 - a0) * 4);
                for (uint64_t k = a0; k < a1; ++k) { if (k) o.push_back(' '); o += valueText[val_at(c, width, pos + k)]; }
            };
            if (k1 - k0 < (1u << 16)) { for (unsigned t = 0; t < T; ++t) work(t); }
            else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t);
                for (auto& x : th) x.join();
            }
            for (unsigned t = 0; t < T; ++t) f.put(buf[t]);
        }
        pos += len;
        f.put('\n');
    }
    (void)n;
    return true;
}

// Calls fn(seqIndex, startInSeq, runLength, value) for every maximal run of equal values inside one sequence.
template <class Fn> static void for_each_run(const void* c, int width, const SeqTable& seqs, Fn fn)
{
    uint64_t base = 0;
    for (size_t s = 0; s < seqs.lengths.size(); ++s) {
        const uint64_t len = seqs.lengths[s];
        uint64_t k = 0;
        while (k < len) {
            const uint32_t v = val_at(c, width, base + k);
            uint64_t e = k + 1;
            while (e < len && val_at(c, width, base + e) == v) ++e;
            fn(s, k, e - k, v);
            k = e;
        }
        base += len;
    }
}

// Same callback protocol, runs taken from the GPU (only non-zero runs exist there).
template <class Fn> static void for_each_given_run(const RunsInput& runs, const SeqTable& seqs, Fn fn)
{
    size_t s = 0; uint64_t base = 0;
    for (uint64_t r = 0; r < runs.n; ++r) {
        while (s + 1 < seqs.lengths.size() && runs.start[r] >= base + seqs.lengths[s]) { base += seqs.lengths[s]; ++s; }
        fn(s, runs.start[r] - base, runs.length[r], (uint32_t)runs.value[r]);
    }
}

template <class Each>
static bool write_wig(Each each, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err)
{
    {
        BufferedFile f(stem + ".wig");
        if (!f.ok()) { err = "cannot write " + stem + ".wig"; return false; }
        size_t curSeq = (size_t)-1;
        uint64_t lastSpan = 0;
        each([&](size_t s, uint64_t start, uint64_t len, uint32_t v) {
            if (s != curSeq) { curSeq = s; lastSpan = 0; }   // last_occ is per sequence (src/output.hpp:88-90)
            if (v == 0) return;                               // runs of 0 are skipped (:98)
            if (lastSpan != len) { f.put("variableStep chrom="); f.put(seqs.names[s]); f.put(" span="); f.put_u64(len); f.put('\n'); }
            f.put_u64(start + 1); f.put(' '); f.put_value(v, mappability); f.put('\n');   // wig positions start at 1
            lastSpan = len;
        });
    }
    BufferedFile g(stem + ".chrom.sizes");   // src/output.hpp:129-133
    if (!g.ok()) { err = "cannot write " + stem + ".chrom.sizes"; return false; }
    for (size_t s = 0; s < seqs.lengths.size(); ++s) { g.put(seqs.names[s]); g.put('\t'); g.put_u64(seqs.lengths[s]); g.put('\n'); }
    return true;
}

bool save_wig(const void* c, uint64_t n, int width, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err)
{
    (void)n;
    return write_wig([&](auto fn) { for_each_run(c, width, seqs, fn); }, stem, seqs, mappability, err);
}
// ---- run lists from the GPU, formatted on all host cores ---------------------------------------------------------------
// A wig / bedgraph / bed line depends only on its run (and, for wig's "variableStep" header, on the run before it in the same
// sequence), so batches of runs are formatted by all host threads into per-thread buffers and written in order.  At 3.09 Gbp
// the single-threaded writer spent 18.5 s of a 28 s `genmap map -bg` here (profiles/r02/cli_scale_grch38.txt).
namespace {
inline void append_u64(std::string& o, uint64_t v)
{
    char b[24]; int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(b[--n]);
}
enum class RunText { Wig, Bedgraph, Bed };

bool write_runs_parallel(const RunsInput& runs, const SeqTable& seqs, RunText kind, bool mappability, BufferedFile& f)
{
    std::vector<std::string> valueText(65536);   // text of every value (operator<< of float == "%g", of integers == "%u")
    {
        std::vector<uint8_t> used(65536, 0);
        for (uint64_t r = 0; r < runs.n; ++r) used[runs.value[r]] = 1;
        for (uint32_t v = 0; v < 65536; ++v) if (used[v]) { char b[32]; int n = mappability ? fmt_float(b, inverse_of(v)) : snprintf(b, sizeof b, "%u", v); valueText[v].assign(b, (size_t)n); }
    }
    std::vector<uint64_t> cum(seqs.lengths.size() + 1, 0);
    for (size_t s = 0; s < seqs.lengths.size(); ++s) cum[s + 1] = cum[s] + seqs.lengths[s];
    auto seq_of = [&](uint64_t pos) {   // sequence holding slice position pos (empty trailing positions belong to the last one)
        size_t s = (size_t)(std::upper_bound(cum.begin(), cum.end(), pos) - cum.begin());
        s = s ? s - 1 : 0;
        return std::min(s, seqs.lengths.size() - 1);
    };
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const uint64_t BATCH = 16ull << 20;
    std::vector<std::string> buf(T);
    for (uint64_t b0 = 0; b0 < runs.n; b0 += BATCH) {
        const uint64_t b1 = std::min(runs.n, b0 + BATCH), per = (b1 - b0 + T - 1) / T;
        auto work = [&](unsigned t) {
            std::string& o = buf[t]; o.clear();
            const uint64_t r0 = std::min(b1, b0 + t * per), r1 = std::min(b1, r0 + per);
            if (r0 >= r1) return;
            o.reserve((size_t)(r1 - r0) * 40);
            size_t s = seq_of(runs.start[r0]);
            // wig: span of the previous non-zero run of the same sequence (src/output.hpp:88-90)
            // (a run list may hold zero runs, which the serial writer skips without touching last_occ: walk back to the nearest
            // non-zero run of this sequence)
            uint64_t lastSpan = 0;
            if (kind == RunText::Wig)
                for (uint64_t q = r0; q > 0 && runs.start[q - 1] >= cum[s]; --q)
                    if (runs.value[q - 1] != 0) { lastSpan = runs.length[q - 1]; break; }
            for (uint64_t r = r0; r < r1; ++r) {
                const uint64_t st = runs.start[r];
                while (s + 1 < seqs.lengths.size() && st >= cum[s + 1]) { ++s; lastSpan = 0; }
                const uint64_t in = st - cum[s], len = runs.length[r];
                const uint32_t v = runs.value[r];
                if (v == 0) continue;
                if (kind == RunText::Wig) {
                    if (lastSpan != len) { o += "variableStep chrom="; o += seqs.names[s]; o += " span="; append_u64(o, len); o.push_back('\n'); }
                    append_u64(o, in + 1); o.push_back(' '); o += valueText[v]; o.push_back('\n');
                    lastSpan = len;
                } else {
                    o += seqs.names[s]; o.push_back('\t'); append_u64(o, in); o.push_back('\t'); append_u64(o, in + len); o.push_back('\t');
                    if (kind == RunText::Bed) { o.push_back('-'); o.push_back('\t'); }
                    o += valueText[v]; o.push_back('\n');
                }
            }
        };
        if (b1 - b0 < (1u << 16)) { for (unsigned t = 0; t < T; ++t) work(t); }   // small outputs: not worth the threads
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        }
        for (unsigned t = 0; t < T; ++t) f.put(buf[t]);
    }
    return true;
}
}  // namespace

bool save_wig_runs(const RunsInput& runs, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err)
{
    {
        BufferedFile f(stem + ".wig");
        if (!f.ok()) { err = "cannot write " + stem + ".wig"; return false; }
        if (!seqs.lengths.empty()) write_runs_parallel(runs, seqs, RunText::Wig, mappability, f);
    }
    BufferedFile g(stem + ".chrom.sizes");   // src/output.hpp:129-133
    if (!g.ok()) { err = "cannot write " + stem + ".chrom.sizes"; return false; }
    for (size_t s = 0; s < seqs.lengths.size(); ++s) { g.put(seqs.names[s]); g.put('\t'); g.put_u64(seqs.lengths[s]); g.put('\n'); }
    return true;
}

template <class Each>
static bool write_bedgraph(Each each, const std::string& stem, const SeqTable& seqs, bool bedgraph, bool mappability, std::string& err)
{
    BufferedFile f(stem + (bedgraph ? ".bedgraph" : ".bed"));
    if (!f.ok()) { err = "cannot write " + stem; return false; }
    each([&](size_t s, uint64_t start, uint64_t len, uint32_t v) {   // src/output.hpp:150-186
        if (v == 0) return;
        f.put(seqs.names[s]); f.put('\t'); f.put_u64(start); f.put('\t'); f.put_u64(start + len); f.put('\t');
        if (!bedgraph) { f.put('-'); f.put('\t'); }
        f.put_value(v, mappability); f.put('\n');
    });
    return true;
}

bool save_bedgraph(const void* c, uint64_t n, int width, const std::string& stem, const SeqTable& seqs, bool bedgraph, bool mappability, std::string& err)
{
    (void)n;
    return write_bedgraph([&](auto fn) { for_each_run(c, width, seqs, fn); }, stem, seqs, bedgraph, mappability, err);
}
bool save_bedgraph_runs(const RunsInput& runs, const std::string& stem, const SeqTable& seqs, bool bedgraph, bool mappability, std::string& err)
{
    BufferedFile f(stem + (bedgraph ? ".bedgraph" : ".bed"));
    if (!f.ok()) { err = "cannot write " + stem; return false; }
    if (!seqs.lengths.empty()) write_runs_parallel(runs, seqs, bedgraph ? RunText::Bedgraph : RunText::Bed, mappability, f);
    return true;
}

bool save_csv(const std::string& stem, const CsvInput& in, const SeqTable& seqs, uint32_t K, bool revCompl,
              const std::vector<std::string>& fileNames, const std::vector<uint64_t>& seqsPerFile, bool append, std::string& err)
{
    BufferedFile f(stem + ".csv", append);
    if (!f.ok()) { err = "cannot write " + stem + ".csv"; return false; }
    if (!append) {   // header: src/output.hpp:214-222
        f.put("\"k-mer\"");
        for (auto& n : fileNames) { f.put(";\"+ strand "); f.put(n); f.put('"'); }
        if (revCompl) for (auto& n : fileNames) { f.put(";\"- strand "); f.put(n); f.put('"'); }
        f.put('\n');
    }
    std::vector<uint64_t> lastSeqOfFile;   // cumulative number of sequences - 1 per fasta file (:199-211)
    uint64_t acc = 0;
    for (uint64_t c : seqsPerFile) { acc += c; lastSeqOfFile.push_back(acc - 1); }
    std::vector<uint64_t> cum(seqs.lengths.size() + 1, 0);
    for (size_t s = 0; s < seqs.lengths.size(); ++s) cum[s + 1] = cum[s] + seqs.lengths[s];

    // A row depends on its position alone, so shares of the window are formatted by all host cores into per-thread buffers and
    // written in order (the single-threaded writer spent 25 of the 27.7 s of config C5 here, profiles/r02/cli_c5.txt).
    auto put_lists = [&](std::string& o, const uint64_t* list, uint64_t b, uint64_t e) {   // :246-265 / :267-284
        uint64_t i = b, prevSeqs = 0;
        for (size_t fi = 0; fi < lastSeqOfFile.size(); ++fi) {
            o.push_back(';');
            bool first = true;
            while (i < e && (list[i] >> 32) <= lastSeqOfFile[fi]) {
                if (!first) o.push_back('|');
                append_u64(o, (list[i] >> 32) - prevSeqs); o.push_back(','); append_u64(o, list[i] & 0xFFFFFFFFull);
                first = false; ++i;
            }
            prevSeqs = lastSeqOfFile[fi] + 1;
        }
    };
    auto format_range = [&](std::string& o, uint64_t j0, uint64_t j1) {
        if (j0 >= j1) return;
        size_t s = (size_t)(std::upper_bound(cum.begin(), cum.end(), in.posBegin + j0) - cum.begin());
        s = std::min(s ? s - 1 : 0, seqs.lengths.size() - 1);
        for (uint64_t jj = j0; jj < j1; ++jj) {
            const uint64_t pb = in.plusOff[jj], pe = in.plusOff[jj + 1], mb = in.minusOff[jj], me = in.minusOff[jj + 1];
            if (pb == pe && mb == me) continue;                    // "is there at least a hit" (src/algo.hpp:378)
            const uint64_t j = in.posBegin + jj;
            while (s + 1 < seqs.lengths.size() && cum[s + 1] <= j) ++s;   // myPosLocalize (src/common.hpp:21-28)
            const uint64_t off = j - cum[s];
            if ((int64_t)off > (int64_t)seqs.lengths[s] - (int64_t)K) continue;   // k-mer spans two sequences (src/algo.hpp:381)
            append_u64(o, s); o.push_back(','); append_u64(o, off);
            put_lists(o, in.plus, pb, pe);
            if (revCompl) put_lists(o, in.minus, mb, me);
            o.push_back('\n');
        }
    };
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const uint64_t BATCH = 1ull << 21;   // positions per round: bounds the buffered text (a row is ~100 bytes on config C5)
    std::vector<std::string> buf(T);
    for (uint64_t b0 = 0; b0 < in.nPositions; b0 += BATCH) {
        const uint64_t b1 = std::min(in.nPositions, b0 + BATCH), per = (b1 - b0 + T - 1) / T;
        auto work = [&](unsigned t) { buf[t].clear(); format_range(buf[t], std::min(b1, b0 + t * per), std::min(b1, b0 + (t + 1) * per)); };
        if (T == 1 || b1 - b0 < (1u << 14)) { buf[0].clear(); format_range(buf[0], b0, b1); f.put(buf[0]); continue; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
        for (unsigned t = 0; t < T; ++t) f.put(buf[t]);
    }
    return true;
}

uint32_t default_infix_length(uint32_t K, uint32_t E, int32_t xo)
{
    unsigned overlap;
    if (xo >= 0) overlap = (unsigned)xo;
    else if (E == 0) overlap = (unsigned)(K * 0.7);
    else {
        unsigned mm = std::min(std::max(K, 30u), 100u);
        overlap = (unsigned)((K * mm) * std::pow((double)0.7f, (double)E) / 100.0);
    }
    uint64_t maxPossibleOverlap = std::min(K - 1u, K - E - 2u);
    if (overlap > maxPossibleOverlap) { if (xo >= 0) return 0; overlap = (unsigned)maxPossibleOverlap; }
    return K - overlap;
}

}  // namespace gmh

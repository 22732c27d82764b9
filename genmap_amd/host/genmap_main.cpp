// genmap_main.cpp -- the `genmap` host program of the MI355X build: `genmap index` and `genmap map`.
//
// Command line, messages, exit codes and output files follow the reference
//   main / sub-command dispatch   /root/reference/src/genmap.cpp:16-94
//   genmap map                    /root/reference/src/mappability.hpp:409-642 (options :417-466), per-fasta loop :271-365
//   genmap index                  /root/reference/src/indexing.hpp:277-510
// but every computation goes through the C ABI of libgenmap_amd.so (include/genmap_amd.h): the index is
// suffix-sorted on the GPU, computeMappability runs as HIP kernels.  There is no CPU compute path.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include <cmath>
#include <csignal>
#include <execinfo.h>
#include <thread>
#include <vector>
#include "../../include/genmap_amd.h"
#include "gm_hostlib.h"

namespace {

const char* kVersion = "1.3.0-mi355x";

double wall() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec + t.tv_usec * 1e-6; }

struct OptSpec { const char* s; const char* l; bool value; };

// minimal re-statement of the SeqAn ArgumentParser behaviour the reference relies on: -x / --long, values as next token
struct Args {
    std::map<std::string, std::string> val;
    std::map<std::string, bool> flag;
    bool parse(int argc, const char** argv, const std::vector<OptSpec>& specs, std::string& err)
    {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            const OptSpec* sp = nullptr;
            for (auto& o : specs)
                if ((a.size() > 1 && a[0] == '-' && a[1] != '-' && a.substr(1) == o.s) || (a.size() > 2 && a.compare(0, 2, "--") == 0 && a.substr(2) == o.l)) { sp = &o; break; }
            if (!sp) { err = "Unknown option: " + a; return false; }
            if (sp->value) {
                if (i + 1 >= argc) { err = std::string("Option ") + a + " needs a value"; return false; }
                val[sp->l] = argv[++i];
            } else flag[sp->l] = true;
        }
        return true;
    }
    bool has(const char* l) const { return val.count(l) || flag.count(l); }
    std::string get(const char* l, const std::string& d = "") const { auto it = val.find(l); return it == val.end() ? d : it->second; }
};

bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

int fail_gm(const char* what, int rc)
{
    std::cerr << "ERROR: " << what << ": " << gm_status_string(rc) << " (" << gm_last_error() << ")\n";
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------
int index_main(int argc, const char** argv)
{
    const std::vector<OptSpec> specs = {{"F", "fasta-file", true}, {"FD", "fasta-directory", true}, {"I", "index", true}, {"A", "algorithm", true},
                                        {"S", "sampling", true}, {"v", "verbose", false}, {"xa", "seqno", true}, {"xb", "seqpos", true}, {"xc", "bwtlen", true},
                                        {"B", "block-bytes", true}, {"D", "device", true}};
    Args a; std::string err;
    if (!a.parse(argc, argv, specs, err)) { std::cerr << "genmap index: " << err << "\n"; return 1; }
    if (!a.has("index")) { std::cerr << "genmap index: option -I/--index is required\n"; return 1; }
    const bool isFile = a.has("fasta-file"), isDir = a.has("fasta-directory");
    if (isFile && isDir) { std::cerr << "ERROR: You can only use eiher --fasta-file or --fasta-directory, not both.\n"; return 1; }
    if (!isFile && !isDir) { std::cerr << "ERROR: You forgot to specify --fasta-file or --fasta-directory.\n"; return 1; }
    const std::string algo = a.get("algorithm", "divsufsort");
    if (algo != "divsufsort" && algo != "skew") { std::cerr << "genmap index: the value '" << algo << "' of option -A is not one of [divsufsort, skew]\n"; return 1; }
    // the reference defaults to -S 10 to bound host memory; here the array lives in 288 GB of HBM and the full one (1) also lets
    // the search settle narrow nodes by reading the text, so 1 is the default and 2..64 are honoured when asked for
    const int sampling = std::atoi(a.get("sampling", "1").c_str());
    if (sampling < 1 || sampling > 64) { std::cerr << "genmap index: the value of -S must be in [1, 64]\n"; return 1; }
    const bool verbose = a.has("verbose");
    std::string fastaPath = isDir ? a.get("fasta-directory") : a.get("fasta-file");
    if (isDir && !is_dir(fastaPath)) { std::cerr << "ERROR: The fasta directory does not exist!\n"; return 1; }
    if (isFile && !exists(fastaPath)) { std::cerr << "ERROR: The fasta file does not exist!\n"; return 1; }
    std::string indexPath = a.get("index");
    if (exists(indexPath)) { std::cerr << "ERROR: The directory for the index already exists at " << indexPath << "\n       Please remove it, or choose a different location.\n"; return 1; }
    if (mkdir(indexPath.c_str(), 0755)) { std::cerr << "ERROR: Cannot create directory at " << indexPath << '\n'; return 1; }

    gmh::IndexMeta meta; meta.directory = isDir;
    std::vector<uint8_t> text; std::vector<uint64_t> seqLen;
    auto ingest = [&](const std::string& full, const std::string& name) {
        std::vector<gmh::FastaRecord> recs;
        if (!gmh::read_fasta(full, recs, err)) { std::cerr << "ERROR: " << err << "\n"; return false; }
        if (recs.empty()) { std::cerr << "WARNING: The fasta file " << full << " seems to be empty. Excluded from indexing.\n"; return true; }
        for (auto& r : recs) {
            meta.ids.push_back({name, (uint64_t)r.codes.size(), r.id});
            seqLen.push_back(r.codes.size());
            text.insert(text.end(), r.codes.begin(), r.codes.end());
        }
        return true;
    };
    if (isDir) {
        std::vector<std::pair<std::string, std::string>> files;
        if (!gmh::list_fasta_directory(fastaPath, files, err)) { rmdir(indexPath.c_str()); std::cerr << err; return 1; }
        for (auto& f : files) if (!ingest(f.first + f.second, f.second)) return 1;
        if (seqLen.empty()) { rmdir(indexPath.c_str()); std::cerr << "ERROR: No (non-empty) fasta file found!\n"; return 1; }
        std::cout << files.size() << " fasta files have been loaded (run with --verbose to list the files):\n";
        if (verbose) for (auto& f : files) std::cout << f.first << f.second << '\n';
    } else {
        size_t sl = fastaPath.find_last_of('/');
        if (!ingest(fastaPath, sl == std::string::npos ? fastaPath : fastaPath.substr(sl + 1))) return 1;
    }
    if (seqLen.empty()) { rmdir(indexPath.c_str()); std::cerr << "ERROR: There is no non-empty sequence in the fasta file(s).\n"; return 1; }

    // alphabet and index dimensions (src/indexing.hpp:459-470,152-170)
    const bool dna5 = std::find(text.begin(), text.end(), (uint8_t)4) != text.end();
    uint64_t maxLen = 0, total = seqLen.size();
    for (uint64_t l : seqLen) { total += l; maxLen = std::max(maxLen, l); }
    meta.alphabetSize = dna5 ? 5 : 4;
    if (seqLen.size() <= 0xFFFFull && maxLen <= 0xFFFFFFFFull) { meta.seqNoBits = 16; meta.seqPosBits = 32; meta.bwtBits = total <= 0xFFFFFFFFull ? 32 : 64; }
    else if (seqLen.size() <= 0xFFFFFFFFull && maxLen <= 0xFFFFull) { meta.seqNoBits = 32; meta.seqPosBits = 16; meta.bwtBits = 64; }
    else { meta.seqNoBits = 64; meta.seqPosBits = 64; meta.bwtBits = 64; }
    meta.sampling = (uint32_t)sampling;
    if (verbose)
        std::cout << "Index will be constructed using " << (dna5 ? "dna5/rna5" : "dna4/rna4") << " alphabet.\n"
                  << "- The BWT is represented by " << meta.bwtBits << " bit values.\n"
                  << "- The suffix array is sampled at rate " << sampling << " and kept as pairs of " << meta.seqNoBits << " and " << meta.seqPosBits << " bit values.\n";
    std::cout << "Suffix sorting runs on the GPU (prefix doubling, algorithm option '" << algo << "' is accepted for compatibility).\n" << std::flush;

    const double t0 = wall();
    gm_index* ix = nullptr;
    std::cout << "Create fwd Index ... Create bwd Index ... " << std::flush;
    int rc = gm_index_build(text.data(), seqLen.data(), (uint32_t)seqLen.size(), (uint32_t)sampling, (uint32_t)std::atoi(a.get("block-bytes", "0").c_str()),
                            std::atoi(a.get("device", "0").c_str()), &ix);
    if (rc) { rmdir(indexPath.c_str()); return fail_gm("index construction failed", rc); }
    std::cout << "done!\n";
    gm_index_info info; gm_index_get_info(ix, &info);
    std::vector<uint8_t> bf(info.n_rows), br(info.n_rows); gmh::SaFiles sa;
    rc = gm_index_export_bwt(ix, bf.data(), br.data());
    if (!rc && sampling == 1) { sa.full.resize(info.n_rows * (info.row_bits / 32)); rc = gm_index_export_sa(ix, sa.full.data(), info.row_bits / 8); }
    if (!rc && sampling > 1) {
        uint64_t ns = 0;
        rc = gm_index_export_sa_sampled(ix, nullptr, nullptr, 0, &ns);
        if (!rc) { sa.marks.resize((info.n_rows + 31) / 32); sa.samples.resize(ns * (info.row_bits / 32)); rc = gm_index_export_sa_sampled(ix, sa.marks.data(), sa.samples.data(), info.row_bits / 8, &ns); }   // (64-bit rows: two words per sample)
    }
    gm_index_free(ix);
    if (rc) return fail_gm("index export failed", rc);
    if (!gmh::write_index_dir(indexPath, meta, text, bf, br, sa, err)) { std::cerr << "ERROR: " << err << "\n"; return 1; }
    if (verbose) std::cout << "Index of " << info.n_rows << " rows built and written in " << (wall() - t0) << " seconds\n";
    std::cout << "Index created successfully.\n";
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int map_main(int argc, const char** argv)
{
    const std::vector<OptSpec> specs = {
        {"I", "index", true}, {"O", "output", true}, {"E", "errors", true}, {"K", "length", true}, {"S", "selection", true},
        {"nc", "no-reverse-complement", false}, {"ep", "exclude-pseudo", false}, {"fs", "frequency-small", false}, {"fl", "frequency-large", false},
        {"r", "raw", false}, {"t", "txt", false}, {"w", "wig", false}, {"bg", "bedgraph", false}, {"b", "bed", false}, {"d", "csv", false},
        {"m", "memory-mapping", false}, {"T", "threads", true}, {"v", "verbose", false}, {"xo", "overlap", true}, {"D", "device", true}, {"B", "block-bytes", true}};
    Args a; std::string err;
    if (!a.parse(argc, argv, specs, err)) { std::cerr << "genmap map: " << err << "\n"; return 1; }
    for (const char* req : {"index", "output", "length"})
        if (!a.has(req)) { std::cerr << "genmap map: option --" << req << " is required\n"; return 1; }
    const bool wig = a.has("wig"), bg = a.has("bedgraph"), bed = a.has("bed"), raw = a.has("raw"), txt = a.has("txt"), csv = a.has("csv"), verbose = a.has("verbose");
    if (!wig && !bg && !bed && !raw && !txt && !csv) {
        std::cerr << "ERROR: Please choose at least one output format (i.e., --wig, --bedgraph, --bed, --raw, --txt, --csv).\n"; return 1; }
    const bool fs = a.has("frequency-small"), fl = a.has("frequency-large");
    if (fs && fl) { std::cerr << "ERROR: Cannot use both --frequency-small and --frequency-large. Please choose one.\n"; return 1; }
    const gmh::ValueKind kind = fs ? gmh::ValueKind::Freq8 : fl ? gmh::ValueKind::Freq16 : gmh::ValueKind::Mappability;
    const bool mappability = kind == gmh::ValueKind::Mappability;
    const uint32_t K = (uint32_t)std::atoi(a.get("length").c_str());
    const uint32_t E = (uint32_t)std::atoi(a.get("errors", "0").c_str());   // the reference leaves -E uninitialised when absent; 0 here
    const bool revCompl = !a.has("no-reverse-complement"), ep = a.has("exclude-pseudo");
    if (E > 4) { std::cerr << "E > 4 not yet supported.\n"; return 1; }
    const int32_t xo = a.has("overlap") ? std::atoi(a.get("overlap").c_str()) : -1;
    const uint32_t infix = gmh::default_infix_length(K, E, xo);
    if (infix == 0) { std::cerr << "ERROR: overlap cannot be larger than min(K - 1, K - E - 2) = " << std::min(K - 1u, K - E - 2u) << ".\n"; return 1; }

    std::string indexPath = a.get("index");
    gmh::IndexMeta meta; std::vector<uint8_t> text, bf, br; gmh::SaFiles sa;
    const double tRead = wall();
    if (!gmh::read_index_dir(indexPath, meta, text, bf, br, sa, err)) { std::cout << err << (err.empty() || err.back() != '\n' ? "\n" : ""); return 1; }
    const double readSeconds = wall() - tRead;

    // output path: directory, or a file name for single-fasta indices (src/mappability.hpp:562-619)
    std::string outputPath = a.get("output");
    bool outputIncludesFilename = false;
    if (is_dir(outputPath)) { if (outputPath.back() != '/') outputPath += '/'; }
    else if (!meta.directory) {
        if (outputPath.back() == '.') outputPath += '/';
        else {
            size_t sl = outputPath.find_last_of('/');
            std::string parent = sl == std::string::npos ? "." : outputPath.substr(0, sl);
            outputIncludesFilename = true;
            if (!is_dir(parent)) {
                std::cerr << "ERROR: The output cannot be written to the file " << outputPath << ".\n       It seems the directory " << parent << " does not exist.\n";
                return 1;
            }
        }
    } else {
        std::cerr << "ERROR: The output directory " << outputPath << " does not exist.\n"
                  << "       A filename can only be specified for single indexed fasta files (not for indexed fasta directories).\n"
                  << "       Please create it, or choose a different location.\n";
        return 1;
    }
    if (verbose) {
        std::cout << "Index was loaded (dna" << meta.alphabetSize << " alphabet, sampling rate of " << meta.sampling << ").\n"
                  << "- The BWT is represented by " << meta.bwtBits << " bit values.\n"
                  << "- The sampled suffix array is represented by pairs of " << meta.seqNoBits << " and " << meta.seqPosBits << " bit values.\n";
        std::cout << (meta.directory ? "- Index was built on an entire directory.\n" : "- Index was built on a single fasta file.\n") << std::flush;
    }

    // selection (src/mappability.hpp:253-269): BED3 rows keyed by sequence name
    std::map<std::string, std::vector<std::pair<uint64_t, uint64_t>>> selection;
    const bool haveSelection = a.has("selection");
    if (haveSelection) {
        FILE* f = fopen(a.get("selection").c_str(), "r");
        if (!f) { std::cerr << "ERROR: cannot open " << a.get("selection") << "\n"; return 1; }
        // BED3 rows: ref <TAB> begin <TAB> end; the ref field runs to the first TAB (ids may contain spaces when the shortened
        // ids are not unique, src/indexing.hpp:36-61); rows without tabs fall back to blanks as separators
        std::string row; int ch;
        auto flush = [&]() {
            if (row.empty() || row[0] == '#') { row.clear(); return; }
            size_t t1 = row.find('\t'), t2 = t1 == std::string::npos ? t1 : row.find('\t', t1 + 1);
            if (t2 == std::string::npos) { t1 = row.find_first_of(" \t"); t2 = t1 == std::string::npos ? t1 : row.find_first_of(" \t", row.find_first_not_of(" \t", t1)); }
            if (t1 != std::string::npos && t2 != std::string::npos) {
                char* endp = nullptr;
                const unsigned long long b = strtoull(row.c_str() + t1 + 1, &endp, 10), e = strtoull(row.c_str() + t2 + 1, nullptr, 10);
                if (endp != row.c_str() + t1 + 1) selection[row.substr(0, t1)].push_back({b, e});
            }
            row.clear();
        };
        while ((ch = fgetc(f)) != EOF) { if (ch == '\n') flush(); else if (ch != '\r') row.push_back((char)ch); }
        flush();
        fclose(f);
    }

    if (haveSelection) {   // a BED row whose reference names no indexed sequence selects nothing: say so
        for (auto& kv : selection) {
            bool known = false;
            for (auto& r : meta.ids) if (r.name == kv.first) { known = true; break; }
            if (!known) std::cerr << "WARNING: the selection names \"" << kv.first << "\", which is not a sequence of this index; its " << kv.second.size() << " interval(s) are ignored.\n";
        }
    }
    // sequences and files of the index (src/mappability.hpp:225-250)
    std::vector<uint64_t> seqLen; std::vector<uint32_t> seqFile; std::vector<std::string> fileNames; std::vector<uint64_t> seqsPerFile;
    for (auto& r : meta.ids) {
        if (fileNames.empty() || fileNames.back() != r.file) { fileNames.push_back(r.file); seqsPerFile.push_back(0); }
        seqLen.push_back(r.length); seqFile.push_back((uint32_t)fileNames.size() - 1); seqsPerFile.back()++;
    }
    // -D 0,1,2,...: one index replica per listed GPU; every fasta file's k-mer positions are split into contiguous shards,
    // one host thread per device computes its shard (positions are independent given the read-only index)
    std::vector<int> devices;
    { std::string d = a.get("device", "0"); size_t p0 = 0; while (p0 <= d.size()) { size_t c = d.find(',', p0); if (c == std::string::npos) c = d.size(); if (c > p0) devices.push_back(std::atoi(d.substr(p0, c - p0).c_str())); p0 = c + 1; } }
    if (devices.empty()) devices.push_back(0);
    std::vector<gm_index*> replicas(devices.size(), nullptr);
    std::vector<int> rcs(devices.size(), 0);
    const double tLoad = wall();
    {
        std::vector<std::thread> th;
        const uint32_t bb = (uint32_t)std::atoi(a.get("block-bytes", "0").c_str());
        const bool wideRows = (bb & GM_BLOCK_WIDE_ROWS) != 0 || bf.size() >= 0xFFFFFFFFull;   // 64-bit rows: two words per suffix array entry
        for (size_t d = 0; d < devices.size(); ++d)
            th.emplace_back([&, d] {
                rcs[d] = !sa.marks.empty()
                    ? gm_index_import_sampled(bf.data(), br.data(), sa.marks.data(), sa.samples.data(), wideRows ? 8u : 4u, sa.samples.size() / (wideRows ? 2 : 1), text.data(), seqLen.data(),
                                              (uint32_t)seqLen.size(), meta.sampling, bb, devices[d], &replicas[d])
                    : gm_index_import(bf.data(), br.data(), sa.full.empty() ? nullptr : sa.full.data(), sa.full.size() == 2 * bf.size() ? 8u : 4u, text.data(), seqLen.data(), (uint32_t)seqLen.size(),
                                      sa.full.empty() ? 0 : 1, bb, devices[d], &replicas[d]); });
        for (auto& t : th) t.join();
    }
    for (size_t d = 0; d < devices.size(); ++d) if (rcs[d]) { for (auto* r : replicas) gm_index_free(r); return fail_gm("cannot load the index onto the GPU", rcs[d]); }
    gm_index* ix = replicas[0];
    int rc = 0;
    if (verbose)   // load, compute and write are reported separately
        std::cout << "- Index files read in " << (std::round(readSeconds * 100.0) / 100.0) << " seconds, loaded onto " << devices.size() << " GPU(s) in "
                  << (std::round((wall() - tLoad) * 100.0) / 100.0) << " seconds\n" << std::flush;
    { std::vector<uint8_t>().swap(bf); std::vector<uint8_t>().swap(br); sa = gmh::SaFiles(); }

    const double start = wall();
    // An index of several fasta files, no selection, one GPU, a dense output format: the files are computed in ONE launch (gm_map_files: the loop
    // of src/mappability.hpp:289-365 fills and drains the device once per file otherwise -- five bacteria: 18 ms instead of 32) and handed to the
    // writers file by file below.
    std::vector<std::vector<uint8_t>> batched;
    if (fileNames.size() > 1 && !haveSelection && replicas.size() == 1 && (raw || txt)) {
        gm_map_params p; memset(&p, 0, sizeof p);
        p.K = K; p.E = E; p.overlap = xo; p.infix = 0; p.revcompl = revCompl; p.value_bits = fs ? 8 : 16; p.exclude_pseudo = ep;
        std::vector<uint32_t> ff, fn; std::vector<void*> outs; uint32_t s0 = 0;
        batched.resize(fileNames.size());
        for (size_t fi = 0; fi < fileNames.size(); ++fi) {
            uint64_t tl = 0; for (uint32_t s = 0; s < seqsPerFile[fi]; ++s) tl += seqLen[s0 + s];
            ff.push_back(s0); fn.push_back((uint32_t)seqsPerFile[fi]); s0 += (uint32_t)seqsPerFile[fi];
            batched[fi].assign((size_t)tl * (fs ? 1 : 2) + 16, 0); outs.push_back(batched[fi].data());
        }
        const int rcb = gm_map_files(ix, (uint32_t)fileNames.size(), ff.data(), fn.data(), &p, seqFile.data(), outs.data());
        if (rcb) { for (auto* r : replicas) gm_index_free(r); return fail_gm("computeMappability failed", rcb); }
    }
    uint64_t textBegin = 0; uint32_t firstSeq = 0;
    for (size_t fi = 0; fi < fileNames.size(); ++fi) {
        const uint32_t nSeq = (uint32_t)seqsPerFile[fi];
        gmh::SeqTable seqs; uint64_t textLen = 0;
        std::vector<uint64_t> intervals;
        for (uint32_t s = 0; s < nSeq; ++s) {
            const auto& row = meta.ids[firstSeq + s];
            auto it = selection.find(row.name);
            if (it != selection.end())
                for (auto& iv : it->second) {
                    if (iv.first >= row.length || iv.second > row.length) {
                        std::cerr << "Error in BED file! Coordinates exceed sequence length: Seq. \"" << row.name << "\" has a length of " << row.length
                                  << ", but half-closed interval [" << iv.first << ", " << iv.second << ") given.\n";
                        for (auto* r : replicas) gm_index_free(r);
                        return 1;
                    }
                    intervals.push_back(textLen + iv.first); intervals.push_back(textLen + iv.second);
                }
            seqs.names.push_back(row.name); seqs.lengths.push_back(row.length); textLen += row.length;
        }
        // no output at all for fasta files without an interval of interest (src/mappability.hpp:308-314)
        if (!(haveSelection && intervals.empty())) {
            gm_map_params p; memset(&p, 0, sizeof p);
            p.K = K; p.E = E; p.overlap = xo; p.infix = 0; p.revcompl = revCompl; p.value_bits = fs ? 8 : 16; p.exclude_pseudo = ep;
            const int width = fs ? 1 : 2;
            std::string stem = outputPath;
            if (!outputIncludesFilename) stem += fileNames[fi].substr(0, fileNames[fi].find_last_of('.')) + ".genmap";   // src/mappability.hpp:76-78
            bool ok = true; double t;
            const double tCompute = wall();
            bool computeReported = false;
            auto computed = [&]() { if (verbose && !computeReported) { computeReported = true; std::cout << "- " << fileNames[fi] << ": computed in " << (std::round((wall() - tCompute) * 1000.0) / 1000.0) << " seconds\n"; } };
            auto report = [&](const char* what) { if (verbose) std::cout << "- " << what << " written in " << (std::round((wall() - t) * 100.0) / 100.0) << " seconds\n"; };
            const uint64_t* ivp = intervals.empty() ? nullptr : intervals.data();
            bool runsOnly = !raw && !txt && !csv && replicas.size() == 1;
            gm_runs* R = nullptr;
            if (runsOnly) {
                // only run-length formats requested: the GPU hands back the runs, the frequency vector never crosses PCIe
                rc = gm_map_runs(ix, textBegin, textLen, firstSeq, nSeq, &p, ivp, intervals.size() / 2, seqFile.data(), &R);
                if (rc == GM_ERR_TOO_LONG) { runsOnly = false; rc = 0; }   // a fasta file of 2^32 - 1 positions or more: the dense vector and the host-side scan
            }
            if (runsOnly) {
                if (rc) { for (auto* r : replicas) gm_index_free(r); return fail_gm("computeMappability failed", rc); }
                computed();
                gmh::RunsInput ri; ri.n = R->n_runs; ri.start = R->start; ri.length = R->length; ri.value = R->value;
                if (wig) { t = wall(); ok = ok && gmh::save_wig_runs(ri, stem, seqs, mappability, err); report("WIG file"); }
                if (bg) { t = wall(); ok = ok && gmh::save_bedgraph_runs(ri, stem, seqs, true, mappability, err); report("bedgraph file"); }
                if (bed) { t = wall(); ok = ok && gmh::save_bedgraph_runs(ri, stem, seqs, false, mappability, err); report("BED file"); }
                gm_runs_free(R);
            } else {
            std::vector<uint8_t> c;
            if (!batched.empty()) c.swap(batched[fi]); else c.assign((size_t)textLen * width + 16, 0);
            if (!batched.empty()) computed();
            else if (raw || txt || wig || bg || bed) {
                if (replicas.size() == 1) {
                    rc = gm_map(ix, textBegin, textLen, firstSeq, nSeq, &p, ivp, intervals.size() / 2, seqFile.data(), c.data());
                } else {
                    // every device computes interleaved chunks of whole k-mer blocks (>= 64 per device: repeats and N deserts are
                    // spread over all of them, cf. the dynamic chunks of src/algo.hpp:422-434) and copies exactly its chunks
                    // into the one result vector -- nothing is merged on the CPU (gm_map_shard).  A selection is small: contiguous shares.
                    const uint64_t numKmers = textLen >= K ? textLen - K + 1 : 0;
                    const size_t nd = replicas.size();
                    const uint32_t tuned = ep ? gm_tuned_infix_length_locating(K, E) : gm_tuned_infix_length(K, E);
                    const uint64_t stepSize = K - (xo >= 0 ? infix : tuned) + 1, nBlocks = (numKmers + stepSize - 1) / stepSize;
                    const uint32_t chunkBlocks = (uint32_t)std::max<uint64_t>(1, (nBlocks + nd * 64 - 1) / (nd * 64));
                    const bool pinned = gm_host_pin(c.data(), c.size()) == GM_OK;
                    std::vector<std::thread> th;
                    for (size_t d = 0; d < nd; ++d)
                        th.emplace_back([&, d] {
                            gm_map_params q = p;
                            if (ivp) { q.flags |= GM_MAP_FLAG_RANGE; q.kmer_begin = numKmers * d / nd; q.kmer_end = numKmers * (d + 1) / nd; }
                            else { q.chunk_blocks = chunkBlocks; q.chunk_index = (uint32_t)d; q.chunk_stride = (uint32_t)nd; }
                            rcs[d] = gm_map_shard(replicas[d], textBegin, textLen, firstSeq, nSeq, &q, ivp, intervals.size() / 2, seqFile.data(), c.data());
                        });
                    for (auto& t : th) t.join();
                    if (pinned) gm_host_unpin(c.data());
                    for (size_t d = 0; d < nd && !rc; ++d) rc = rcs[d];
                }
                if (rc) { for (auto* r : replicas) gm_index_free(r); return fail_gm("computeMappability failed", rc); }
                computed();
            }
            if (raw) { t = wall(); ok = ok && gmh::save_raw(c.data(), textLen, width, stem, kind, err); report("RAW file"); }
            if (txt) { t = wall(); ok = ok && gmh::save_txt(c.data(), textLen, width, stem, seqs, mappability, err); report("TXT file"); }
            if (wig) { t = wall(); ok = ok && gmh::save_wig(c.data(), textLen, width, stem, seqs, mappability, err); report("WIG file"); }
            if (bg) { t = wall(); ok = ok && gmh::save_bedgraph(c.data(), textLen, width, stem, seqs, true, mappability, err); report("bedgraph file"); }
            if (bed) { t = wall(); ok = ok && gmh::save_bedgraph(c.data(), textLen, width, stem, seqs, false, mappability, err); report("BED file"); }
            if (csv && ok) {
                t = wall();
                // windows of k-mer positions, halved whenever one holds too many occurrences for a single gm_locate call
                const uint64_t numKmers = textLen >= K ? textLen - K + 1 : 0;
                // (a gm_locate call holds (4+8) B x 2W on the device and 16 B x W on the host for a window of W positions)
                uint64_t begin = 0, window = std::max<uint64_t>(std::min<uint64_t>(numKmers, 1ull << 26), 1); bool first = true;
                if (numKmers == 0) { gmh::CsvInput in; uint64_t z[1] = {0}; in.plusOff = in.minusOff = z; ok = gmh::save_csv(stem, in, seqs, K, revCompl, fileNames, seqsPerFile, false, err); }
                // every device of -D takes one contiguous range per round (its own halving windows inside); a round's
                // results are written in position order
                const size_t nd = replicas.size();
                if (nd > 1) window = std::max<uint64_t>(1, std::min<uint64_t>((numKmers + nd - 1) / nd, 1ull << 26));
                while (ok && begin < numKmers) {
                    std::vector<std::vector<gm_locations*>> got(nd);
                    std::vector<int> lrc(nd, 0);
                    std::vector<uint64_t> ends(nd, begin);
                    auto work = [&](size_t d, uint64_t rb, uint64_t re) {
                        uint64_t b2 = rb, w2 = std::max<uint64_t>(re - rb, 1), okRun = 0;
                        while (b2 < re) {
                            gm_map_params q = p;
                            q.kmer_begin = b2; q.kmer_end = std::min(re, b2 + w2);
                            gm_locations* L = nullptr;
                            int r2 = gm_locate(replicas[d], textBegin, textLen, firstSeq, nSeq, &q, ivp, intervals.size() / 2, &L);
                            if (r2 == GM_ERR_TOO_LONG && w2 > 1) { w2 = std::max<uint64_t>(1, w2 / 2); okRun = 0; continue; }
                            if (r2) { lrc[d] = r2; return; }
                            if (++okRun >= 4 && w2 < re - rb) { w2 *= 2; okRun = 0; }   // past the repeat that forced small windows: grow again
                            got[d].push_back(L);
                            // the window is rounded to whole k-mer blocks by the library: continue after what it covered
                            b2 = std::max<uint64_t>(q.kmer_end, L->n_positions ? L->pos_begin + L->n_positions : q.kmer_end);
                        }
                        ends[d] = b2;
                    };
                    std::vector<std::thread> th;
                    uint64_t rb = begin;
                    std::vector<std::pair<uint64_t, uint64_t>> rng(nd);
                    for (size_t d = 0; d < nd; ++d) {   // a k-mer block belongs to the range that holds its first k-mer: no overlap
                        const uint64_t re = std::min(numKmers, rb + window);
                        rng[d] = {rb, re}; rb = re;
                    }
                    if (nd == 1) work(0, rng[0].first, rng[0].second);
                    else { for (size_t d = 0; d < nd; ++d) if (rng[d].first < rng[d].second) th.emplace_back(work, d, rng[d].first, rng[d].second); for (auto& t2 : th) t2.join(); }
                    for (size_t d = 0; d < nd && !rc; ++d) rc = lrc[d];
                    for (size_t d = 0; d < nd; ++d)
                        for (gm_locations* L : got[d]) {
                            if (!rc && ok) {
                                gmh::CsvInput in; in.posBegin = L->pos_begin; in.nPositions = L->n_positions; in.plusOff = L->plus_off; in.minusOff = L->minus_off; in.plus = L->plus; in.minus = L->minus;
                                ok = gmh::save_csv(stem, in, seqs, K, revCompl, fileNames, seqsPerFile, !first, err);
                                first = false;
                            }
                            gm_locations_free(L);
                        }
                    if (rc) { for (auto* r : replicas) gm_index_free(r); return fail_gm("locate failed", rc); }
                    begin = rng[nd - 1].second;
                }
                report("CSV file");
            }
            }
            if (!ok) { std::cerr << "ERROR: " << err << "\n"; for (auto* r : replicas) gm_index_free(r); return 1; }
        }
        textBegin += textLen; firstSeq += nSeq;
    }
    if (verbose) std::cout << "Mappability computed in " << (std::round((wall() - start) * 100.0) / 100.0) << " seconds\n";
    for (auto* r : replicas) gm_index_free(r);
    return 0;
}

}  // namespace

// A crash of this program must say where it happened: round 4 saw ONE `genmap index` of ~350 die with SIGSEGV on the GPU box, with
// nothing to locate it by.  The handler writes the faulting thread's stack to stderr with async-signal-safe calls only (the symbol
// names come from the dynamic symbol table: the binary is linked -rdynamic), then lets the default action end the process with the
// original signal, so the exit status callers see does not change.
extern "C" void genmap_crash_handler(int sig, siginfo_t* info, void*)
{
    static const char head[] = "\ngenmap: fatal signal ";
    (void)!write(2, head, sizeof head - 1);
    char num[4] = {(char)('0' + sig / 10 % 10), (char)('0' + sig % 10), ' ', 0};
    (void)!write(2, num, 3);
    static const char at[] = "at address 0x";
    (void)!write(2, at, sizeof at - 1);
    char hex[17]; unsigned long long a = (unsigned long long)(uintptr_t)(info ? info->si_addr : nullptr);
    for (int i = 15; i >= 0; --i) { hex[i] = "0123456789abcdef"[a & 15u]; a >>= 4; }
    hex[16] = '\n';
    (void)!write(2, hex, 17);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, const char** argv)
{
    {
        void* warm[2]; (void)backtrace(warm, 2);   // (loads libgcc's unwinder now: the first call allocates, which a signal handler must not)
        struct sigaction sa; memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = genmap_crash_handler; sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_NODEFER;
        for (int sig : {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT}) sigaction(sig, &sa, nullptr);
    }
    // first non-flag token selects the sub-command (src/genmap.cpp:27-64)
    int cmd = 0;
    for (int i = 1; i < argc; ++i) if (argv[i][0] != '-') { cmd = i; break; }
    if (cmd == 0) {
        for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], "--version")) { std::cout << "GenMap version: " << kVersion << "\n"; return 0; }
        std::cerr << "GenMap (MI355X build) - Fast and Exact Computation of Genome Mappability\nUsage: genmap [index|map] [OPTIONS]\n";
        return 1;
    }
    std::vector<const char*> sub; sub.push_back(argv[0]);
    for (int i = 1; i < argc; ++i) if (i != cmd) sub.push_back(argv[i]);
    if (!strcmp(argv[cmd], "selftest-crash")) { volatile int* nowhere = nullptr; return *nowhere; }   // (tests: the crash handler reports where)
    if (!strcmp(argv[cmd], "index") || !strcmp(argv[cmd], "map")) {
        const int rc = !strcmp(argv[cmd], "index") ? index_main((int)sub.size(), sub.data()) : map_main((int)sub.size(), sub.data());
        // Every output file is closed by now (they are locals of the sub-commands) and every index handle is freed.  The process ends
        // WITHOUT running static destructors: the HIP runtime keeps worker threads of its own, and tearing its globals down under
        // them is the one place a `genmap index` of a 1 kbp fixture -- which starts no thread itself -- can die in a thread whose
        // stack ends in start_thread / clone: 1 process start in ~350 (round 4) and 1 in ~40 (round 5, `profiles/r05/crash_hunt/`)
        // of the GPU suite did, with SIGSEGV, after the index had been written; 3,600 looped runs outside the suite never did.
        // GENMAP_FULL_EXIT=1 takes the ordinary way out (static destructors, atexit handlers): profilers, sanitizers and coverage
        // tools write their reports there; tools/crash_hunt.sh loops over that path.
        std::cout.flush(); std::cerr.flush(); fflush(nullptr);
        { const char* fe = getenv("GENMAP_FULL_EXIT"); if (fe && fe[0] == '1') return rc; }
        _exit(rc);
    }
    std::cerr << "Invalid argument " << argv[cmd] << ". Use 'genmap index' or 'genmap map'.\n";
    return 1;
}

// gm_hostlib.h -- host side of the `genmap` program above the C ABI: FASTA ingestion, index directory,
// output writers.  Plain C++17, no GPU code.  Byte-compatibility targets (all relative to /root/reference):
//   FASTA rules            src/indexing.hpp:13-20,36-61,209-275,398-420
//   index.ids / index.info src/indexing.hpp:101-113,268-274 ; src/common.hpp:10-19 ; src/mappability.hpp:551-560
//   output files           src/output.hpp:10-288 ; src/mappability.hpp:69-155
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace gmh {

struct FastaRecord { std::string id; std::vector<uint8_t> codes; };

// one fasta file: records with empty sequences skipped, ids shortened at the first whitespace when still unique
bool read_fasta(const std::string& path, std::vector<FastaRecord>& out, std::string& err);

// all fasta files below `dir` (recursively), sorted by file name; duplicates are an error (returns false)
bool list_fasta_directory(const std::string& dir, std::vector<std::pair<std::string, std::string>>& pathAndName, std::string& err);

struct IdsRow { std::string file; uint64_t length; std::string name; };   // "fastaFile;length;seqName"

struct IndexMeta {
    uint32_t alphabetSize = 4;
    uint32_t seqNoBits = 16, seqPosBits = 32, bwtBits = 32;   // sa_dimensions_i1 / _i2 / bwt_dimensions
    uint32_t sampling = 1;
    bool directory = false;
    std::vector<IdsRow> ids;
};

// The index directory written by `genmap index` of this build (layout documented in DESIGN.md):
//   index.info  index.ids          text, same keys/rows as the reference's
//   index.txt4  index.bwt4  index.rev.bwt4   4-bit packed codes (text without sentinels; BWTs with code 5 = sentinel)
//   index.sa                       sampling_rate 1: the forward suffix array, uint32 little endian, one entry per row
//   index.sa.marks index.sa.samples   sampling_rate s > 1: one bit per row (uint32 words, bit row % 32), and SA[row] of the
//                                  marked rows in row order (rows whose in-sequence offset is a multiple of s)
struct SaFiles { std::vector<uint32_t> full, marks, samples; };
bool write_index_dir(const std::string& dir, const IndexMeta& meta, const std::vector<uint8_t>& text,
                     const std::vector<uint8_t>& bwtFwd, const std::vector<uint8_t>& bwtRev, const SaFiles& sa, std::string& err);
bool read_index_dir(const std::string& dir, IndexMeta& meta, std::vector<uint8_t>& text, std::vector<uint8_t>& bwtFwd,
                    std::vector<uint8_t>& bwtRev, SaFiles& sa, std::string& err);

// ---- writers (src/output.hpp) -------------------------------------------------------------------------------
struct SeqTable { std::vector<std::string> names; std::vector<uint64_t> lengths; };

enum class ValueKind { Mappability, Freq8, Freq16 };   // OutputType of src/mappability.hpp:16-22

// c: one value per text position of the fasta file (uint8 for Freq8, uint16 otherwise), width = bytes per value
bool save_raw(const void* c, uint64_t n, int width, const std::string& stem, ValueKind kind, std::string& err);
bool save_txt(const void* c, uint64_t n, int width, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err);
bool save_wig(const void* c, uint64_t n, int width, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err);
bool save_bedgraph(const void* c, uint64_t n, int width, const std::string& stem, const SeqTable& seqs, bool bedgraph, bool mappability, std::string& err);

// the same two writers fed from the GPU's run-length form (gm_map_runs): non-zero runs in text order, never across sequences
struct RunsInput { uint64_t n = 0; const uint64_t* start = nullptr; const uint64_t* length = nullptr; const uint16_t* value = nullptr; };
bool save_wig_runs(const RunsInput& runs, const std::string& stem, const SeqTable& seqs, bool mappability, std::string& err);
bool save_bedgraph_runs(const RunsInput& runs, const std::string& stem, const SeqTable& seqs, bool bedgraph, bool mappability, std::string& err);

// csv: occurrence lists per slice position as returned by gm_locate (packed seqNo << 32 | seqPos, global seqNo)
struct CsvInput {
    uint64_t posBegin = 0, nPositions = 0;
    const uint64_t *plusOff = nullptr, *minusOff = nullptr, *plus = nullptr, *minus = nullptr;
};
// fileNames / seqsPerFile: every fasta file of the index in order; firstSeq: global number of the slice's first sequence
bool save_csv(const std::string& stem, const CsvInput& in, const SeqTable& seqs, uint32_t K, bool revCompl,
              const std::vector<std::string>& fileNames, const std::vector<uint64_t>& seqsPerFile, bool append, std::string& err);

// SearchParams.overlap (common infix length) for (K, E, -xo); 0 = "-xo too large"  (src/mappability.hpp:519-543)
uint32_t default_infix_length(uint32_t K, uint32_t E, int32_t xo);

}  // namespace gmh

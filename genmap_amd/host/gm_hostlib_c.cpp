// gm_hostlib_c.cpp -- C entry points of the host library so that the writers and the FASTA/index-directory
// code can be exercised from the CPU tests (ctypes) without a GPU.
#include <cstring>
#include <string>
#include <vector>
#include "gm_hostlib.h"

static std::string g_err;

extern "C" {

const char* gmh_last_error() { return g_err.c_str(); }

// names: n_seq NUL-terminated strings back to back.  formats: bit 0 raw, 1 txt, 2 wig, 3 bedgraph, 4 bed.
// kind: 0 mappability, 1 freq8, 2 freq16.  Returns 0 on success.
int gmh_save_outputs(const void* c, uint64_t n, int width, const char* stem, int kind, int formats,
                     const char* names, const uint64_t* lengths, uint32_t n_seq)
{
    gmh::SeqTable seqs;
    const char* p = names;
    for (uint32_t s = 0; s < n_seq; ++s) { seqs.names.emplace_back(p); p += strlen(p) + 1; seqs.lengths.push_back(lengths[s]); }
    const gmh::ValueKind k = kind == 0 ? gmh::ValueKind::Mappability : kind == 1 ? gmh::ValueKind::Freq8 : gmh::ValueKind::Freq16;
    const bool mapp = kind == 0;
    bool ok = true;
    if (formats & 1) ok = ok && gmh::save_raw(c, n, width, stem, k, g_err);
    if (formats & 2) ok = ok && gmh::save_txt(c, n, width, stem, seqs, mapp, g_err);
    if (formats & 4) ok = ok && gmh::save_wig(c, n, width, stem, seqs, mapp, g_err);
    if (formats & 8) ok = ok && gmh::save_bedgraph(c, n, width, stem, seqs, true, mapp, g_err);
    if (formats & 16) ok = ok && gmh::save_bedgraph(c, n, width, stem, seqs, false, mapp, g_err);
    return ok ? 0 : 1;
}

// wig (bit 2) / bedgraph (bit 3) / bed (bit 4) from precomputed non-zero runs (the form gm_map_runs returns)
int gmh_save_outputs_runs(uint64_t n_runs, const uint64_t* start, const uint64_t* length, const uint16_t* value, const char* stem, int kind, int formats,
                          const char* names, const uint64_t* lengths, uint32_t n_seq)
{
    gmh::SeqTable seqs;
    const char* p = names;
    for (uint32_t s = 0; s < n_seq; ++s) { seqs.names.emplace_back(p); p += strlen(p) + 1; seqs.lengths.push_back(lengths[s]); }
    gmh::RunsInput r; r.n = n_runs; r.start = start; r.length = length; r.value = value;
    const bool mapp = kind == 0;
    bool ok = true;
    if (formats & 4) ok = ok && gmh::save_wig_runs(r, stem, seqs, mapp, g_err);
    if (formats & 8) ok = ok && gmh::save_bedgraph_runs(r, stem, seqs, true, mapp, g_err);
    if (formats & 16) ok = ok && gmh::save_bedgraph_runs(r, stem, seqs, false, mapp, g_err);
    return ok ? 0 : 1;
}

int gmh_save_csv(const char* stem, uint64_t pos_begin, uint64_t n_positions, const uint64_t* plus_off, const uint64_t* minus_off,
                 const uint64_t* plus, const uint64_t* minus, const char* seq_names, const uint64_t* lengths, uint32_t n_seq, uint32_t K,
                 int revcompl, const char* file_names, const uint64_t* seqs_per_file, uint32_t n_files, int append)
{
    gmh::SeqTable seqs;
    const char* p = seq_names;
    for (uint32_t s = 0; s < n_seq; ++s) { seqs.names.emplace_back(p); p += strlen(p) + 1; seqs.lengths.push_back(lengths[s]); }
    std::vector<std::string> files; std::vector<uint64_t> spf;
    p = file_names;
    for (uint32_t f = 0; f < n_files; ++f) { files.emplace_back(p); p += strlen(p) + 1; spf.push_back(seqs_per_file[f]); }
    gmh::CsvInput in; in.posBegin = pos_begin; in.nPositions = n_positions; in.plusOff = plus_off; in.minusOff = minus_off; in.plus = plus; in.minus = minus;
    return gmh::save_csv(stem, in, seqs, K, revcompl != 0, files, spf, append != 0, g_err) ? 0 : 1;
}

// FASTA file -> concatenated codes.  Two-call protocol: codes == NULL returns sizes only.
int gmh_read_fasta(const char* path, uint8_t* codes, uint64_t* lengths, char* names, uint64_t names_cap, uint64_t* n_seq, uint64_t* total, uint64_t* names_bytes)
{
    std::vector<gmh::FastaRecord> recs;
    if (!gmh::read_fasta(path, recs, g_err)) return 1;
    uint64_t tot = 0, nb = 0;
    for (auto& r : recs) { tot += r.codes.size(); nb += r.id.size() + 1; }
    *n_seq = recs.size(); *total = tot; *names_bytes = nb;
    if (!codes) return 0;
    if (nb > names_cap) { g_err = "names buffer too small"; return 1; }
    uint64_t o = 0; char* q = names;
    for (size_t i = 0; i < recs.size(); ++i) {
        memcpy(codes + o, recs[i].codes.data(), recs[i].codes.size()); o += recs[i].codes.size();
        lengths[i] = recs[i].codes.size();
        memcpy(q, recs[i].id.c_str(), recs[i].id.size() + 1); q += recs[i].id.size() + 1;
    }
    return 0;
}

uint32_t gmh_default_infix_length(uint32_t K, uint32_t E, int32_t xo) { return gmh::default_infix_length(K, E, xo); }

}  // extern "C"

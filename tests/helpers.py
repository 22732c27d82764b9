"""Test-side helpers: ctypes binding of the CPU oracle (oracle/libgmoracle.so), FASTA/BED parsing
for fixtures, and the table of the reference's 18 end-to-end cases
(/root/reference/tests/CMakeLists.txt:56-73, data copied to tests/golden/reference_cases).

Nothing here is product code; the product never imports tests/ or oracle/.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
CASES_DIR = GOLDEN / "reference_cases"

# case -> (directory mode, map flags) exactly as tests/CMakeLists.txt:56-73
CASES = {
    "1a": (False, dict(E=0, K=3, nc=True)),
    "1b": (False, dict(E=0, K=3)),
    "1c": (False, dict(E=0, K=3, nc=True)),
    "1d": (False, dict(E=0, K=3)),
    "1e": (False, dict(E=1, K=3, nc=True)),
    "1f": (False, dict(E=1, K=3)),
    "1g": (False, dict(E=1, K=3)),
    "2a": (False, dict(E=0, K=4, nc=True)),
    "2b": (False, dict(E=0, K=4)),
    "2c": (False, dict(E=0, K=4, nc=True)),
    "2d": (False, dict(E=0, K=4)),
    "2e": (False, dict(E=0, K=4)),
    "3a": (True, dict(E=0, K=4, nc=True)),
    "3b": (True, dict(E=0, K=4)),
    "3c": (True, dict(E=0, K=4, ep=True, nc=True)),
    "3d": (True, dict(E=0, K=4, ep=True)),
    "3e": (True, dict(E=0, K=4, ep=True)),
    "3f": (True, dict(E=0, K=4, ep=True)),
}


def xo_variants(case):
    """-xo reruns of /root/reference/tests/tests.sh:47-60."""
    v = [None]
    if case not in ("1e", "1f", "1g"):
        v.append(1)
    if case[0] != "1":
        v.append(2)
    return v


_CODE = np.full(256, 4, dtype=np.uint8)  # non-ACGT(U) -> N  (src/indexing.hpp:13-20)
for ch, c in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
    _CODE[ord(ch)] = c
    _CODE[ord(ch.lower())] = c


def read_fasta(path):
    """[(id, codes)] following readFasta (src/indexing.hpp:209-275): empty sequences skipped,
    ids cut at the first whitespace when that keeps them unique."""
    recs, name, chunks = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks)))
                name, chunks = line[1:].decode(), []
            elif name is not None:
                chunks.append(line.strip())
        if name is not None:
            recs.append((name, b"".join(chunks)))
    recs = [(n, s) for n, s in recs if len(s) > 0]
    short = [n.split()[0] if n.split() else "" for n, _ in recs]
    if len(set(short)) == len(short):
        recs = [(s, q) for s, (_, q) in zip(short, recs)]
    return [(n, _CODE[np.frombuffer(s, dtype=np.uint8)]) for n, s in recs]


class Genome:
    """All sequences of all fasta files of one index, in index order."""

    def __init__(self, files):
        # files: [(fasta file name, [(seq name, codes)])], sorted by file name for directories (src/indexing.hpp:407)
        self.files = files
        self.seq_names, self.seq_len, self.seq_file = [], [], []
        codes = []
        for fid, (_, recs) in enumerate(files):
            for n, c in recs:
                self.seq_names.append(n)
                self.seq_len.append(len(c))
                self.seq_file.append(fid)
                codes.append(c)
        self.codes = np.ascontiguousarray(np.concatenate(codes)) if codes else np.zeros(0, np.uint8)
        self.seq_len = np.asarray(self.seq_len, dtype=np.uint64)
        self.seq_file = np.asarray(self.seq_file, dtype=np.uint32)
        self.cum = np.concatenate([[0], np.cumsum(self.seq_len)]).astype(np.uint64)
        self.dna5 = bool((self.codes == 4).any())

    def file_slices(self):
        """[(file name, first_seq, n_seq, text_begin, text_len)]"""
        out, s = [], 0
        for name, recs in self.files:
            n = len(recs)
            out.append((name, s, n, int(self.cum[s]), int(self.cum[s + n] - self.cum[s])))
            s += n
        return out


def load_case(case):
    d = CASES_DIR / f"case_{case}"
    directory, flags = CASES[case]
    if directory:
        names = sorted(p.name for p in d.iterdir() if p.is_file() and p.suffix in (".fa", ".fasta", ".fna", ".fsa", ".fas", ".faa", ".fastq"))
    else:
        names = ["genome.fa"]
    g = Genome([(n, read_fasta(d / n)) for n in names])
    bed = d / "subset.bed"
    intervals = read_bed(bed) if bed.exists() else None
    return g, directory, flags, intervals


def read_bed(path):
    iv = {}
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) >= 3:
                iv.setdefault(t[0], []).append((int(t[1]), int(t[2])))
    return iv


def slice_intervals(g, first_seq, n_seq, intervals):
    """intervalsForSingleFasta of src/mappability.hpp:334-357: cumulative coordinates within the fasta file."""
    out = []
    base = int(g.cum[first_seq])
    for s in range(first_seq, first_seq + n_seq):
        for b, e in intervals.get(g.seq_names[s], []):
            ln = int(g.seq_len[s])
            if b >= ln or e > ln:
                raise ValueError("Error in BED file! Coordinates exceed sequence length")
            off = int(g.cum[s]) - base
            out.append((off + b, off + e))
    return out


# ---------------------------------------------------------------------------------------------
# oracle binding
# ---------------------------------------------------------------------------------------------
class _Params(C.Structure):
    _fields_ = [("K", C.c_uint32), ("E", C.c_uint32), ("overlap", C.c_int32), ("revcompl", C.c_int32),
                ("value_bits", C.c_int32), ("directory", C.c_int32), ("exclude_pseudo", C.c_int32),
                ("csv", C.c_int32), ("threads", C.c_int32), ("use_shortcut", C.c_int32), ("infix", C.c_int32)]


class _Locations(C.Structure):
    _fields_ = [("n_entries", C.c_uint64), ("key_seq", C.POINTER(C.c_uint32)), ("key_pos", C.POINTER(C.c_uint64)),
                ("plus_off", C.POINTER(C.c_uint64)), ("minus_off", C.POINTER(C.c_uint64)),
                ("plus_seq", C.POINTER(C.c_uint32)), ("plus_pos", C.POINTER(C.c_uint64)),
                ("minus_seq", C.POINTER(C.c_uint32)), ("minus_pos", C.POINTER(C.c_uint64))]


_libs = {}


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle")])


def build_oracle_native():
    """-march=native build for the machine this runs on (bench.py's cpu_baseline leg); returns its path or None"""
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", str(ROOT / "oracle"), "native"], timeout=300)
    except Exception:
        return None
    so = ROOT / "oracle" / "_native" / "libgmoracle.so"
    return so if so.exists() else None


def oracle_lib(so=None):
    if so is None:
        so = ROOT / "oracle" / "libgmoracle.so"
        src = ROOT / "oracle" / "gm_oracle.c"
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            build_oracle()
    if str(so) in _libs:
        return _libs[str(so)]
    lib = C.CDLL(str(so))
    vp, u8p, u64p, u32p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    lib.gmo_index_build.restype = vp
    lib.gmo_index_build.argtypes = [u8p, u64p, C.c_uint32, C.c_int]
    lib.gmo_index_from_bwt.restype = vp
    lib.gmo_index_from_bwt.argtypes = [u8p, u8p, u8p, u64p, C.c_uint32]
    lib.gmo_index_adopt_sa.restype = C.c_int
    lib.gmo_index_adopt_sa.argtypes = [vp, vp]
    lib.gmo_index_free.argtypes = [vp]
    lib.gmo_index_size.restype = C.c_uint64
    lib.gmo_index_size.argtypes = [vp]
    lib.gmo_index_bwt.restype = C.POINTER(C.c_uint8)
    lib.gmo_index_bwt.argtypes = [vp, C.c_int]
    lib.gmo_index_sa.restype = C.POINTER(C.c_uint32)
    lib.gmo_index_sa.argtypes = [vp]
    lib.gmo_compute_mappability.restype = C.c_int
    lib.gmo_compute_mappability.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_Params),
                                            u64p, C.c_uint64, u32p, vp, C.POINTER(C.c_int),
                                            C.POINTER(C.POINTER(_Locations))]
    lib.gmo_locations_free.argtypes = [C.POINTER(_Locations)]
    lib.gmo_trivial_backtracking.restype = C.c_int
    lib.gmo_trivial_backtracking.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
    lib.gmo_brute_force.restype = C.c_int
    lib.gmo_brute_force.argtypes = [u8p, u64p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
    lib.gmo_default_infix_length.restype = C.c_int
    lib.gmo_default_infix_length.argtypes = [C.c_uint32, C.c_uint32, C.c_int32]
    lib.gmo_last_counters.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gmo_set_line_symbols.argtypes = [C.c_uint32]
    _libs[str(so)] = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleIndex:
    def __init__(self, codes, seq_len, keep_sa=True, bwt=None, sa=None, lib=None):
        self.lib = lib if lib is not None else oracle_lib()
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.seq_len = np.ascontiguousarray(seq_len, dtype=np.uint64)
        self.cum = np.concatenate([[0], np.cumsum(self.seq_len)]).astype(np.uint64)
        if bwt is None:
            self.h = self.lib.gmo_index_build(_ptr(self.codes), _ptr(self.seq_len), len(self.seq_len), int(keep_sa))
        else:
            bf = np.ascontiguousarray(bwt[0], dtype=np.uint8)
            br = np.ascontiguousarray(bwt[1], dtype=np.uint8)
            self.h = self.lib.gmo_index_from_bwt(_ptr(bf), _ptr(br), _ptr(self.codes), _ptr(self.seq_len), len(self.seq_len))
        if not self.h:
            raise RuntimeError("oracle index build failed")
        if bwt is not None and sa is not None:   # adopted suffix array (checked by the caller, see check_sa_against_bwt)
            sa = np.ascontiguousarray(sa, dtype=np.uint32)
            if self.lib.gmo_index_adopt_sa(self.h, _ptr(sa)) != 0:
                raise RuntimeError("oracle could not adopt the suffix array")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.gmo_index_free(self.h)
            self.h = None

    @property
    def n(self):
        return int(self.lib.gmo_index_size(self.h))

    def bwt(self, rev):
        p = self.lib.gmo_index_bwt(self.h, int(rev))
        return np.ctypeslib.as_array(p, shape=(self.n,)).copy()

    def sa(self):
        p = self.lib.gmo_index_sa(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n,)).copy()

    def mappability(self, K, E, text_begin=0, text_len=None, first_seq=0, n_seq=None, overlap=None, revcompl=True,
                    value_bits=16, directory=False, exclude_pseudo=False, csv=False, threads=1, use_shortcut=False,
                    intervals=None, seq_file_id=None, want_flag=False, infix=0, out=None):
        if n_seq is None:
            n_seq = len(self.seq_len) - first_seq
        if text_len is None:
            text_len = int(self.cum[first_seq + n_seq] - self.cum[first_seq])
        p = _Params(K, E, -1 if overlap is None else overlap, int(revcompl), value_bits, int(directory),
                    int(exclude_pseudo), int(csv), threads, int(use_shortcut), int(infix))
        if out is None:
            out = np.zeros(text_len, dtype=np.uint8 if value_bits == 8 else np.uint16)
        else:   # (bench.py: a vector kept between calls; with gmo_set_skip_clear the caller keeps it zeroed)
            assert out.size == text_len and out.dtype == (np.uint8 if value_bits == 8 else np.uint16) and out.flags.c_contiguous
        iv = None
        if intervals:
            iv = np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
        sf = None if seq_file_id is None else np.ascontiguousarray(seq_file_id, dtype=np.uint32)
        flag = C.c_int(0)
        locs = C.POINTER(_Locations)()
        rc = self.lib.gmo_compute_mappability(self.h, text_begin, text_len, first_seq, n_seq, C.byref(p), _ptr(iv),
                                              0 if iv is None else len(iv) // 2, _ptr(sf), _ptr(out), C.byref(flag),
                                              C.byref(locs))
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}")
        L = None
        if csv:
            L = _unpack_locations(locs.contents)
            self.lib.gmo_locations_free(locs)
        if csv or want_flag:
            return out, bool(flag.value), L
        return out

    def counters(self):
        v, l = C.c_uint64(0), C.c_uint64(0)
        self.lib.gmo_last_counters(C.byref(v), C.byref(l))
        return int(v.value), int(l.value)

    def trivial(self, K, E, revcompl=True, value_bits=16):
        out = np.zeros(int(self.cum[-1]), dtype=np.uint8 if value_bits == 8 else np.uint16)
        self.lib.gmo_trivial_backtracking(self.h, K, E, int(revcompl), value_bits, _ptr(out))
        return out


def check_sa_against_bwt(codes, seq_len, bwt_fwd, sa):
    """A suffix array handed to the oracle together with an adopted forward BWT is pinned by two properties: it is a
    permutation of the sentinel-text positions whose preceding symbols are the BWT (sentinel rows wrap to the sequence's
    own end marker), and following the BWT's LF mapping decrements it (sa[LF(i)] == sa[i] - 1) -- together they
    determine sa from the BWT, whose own correctness the smaller builder-vs-oracle tests pin."""
    seq_len = np.asarray(seq_len, dtype=np.int64)
    n = int(seq_len.sum()) + len(seq_len)
    sa = np.asarray(sa).astype(np.int64)
    assert len(sa) == n and len(bwt_fwd) == n
    textS = np.full(n, 5, dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(seq_len + 1)])[:-1]
    ends = starts + seq_len                      # position of every sequence's sentinel
    pos = 0
    off = 0
    for s, ln in zip(starts, seq_len):
        textS[s:s + ln] = codes[off:off + ln]
        off += int(ln)
    assert np.array_equal(np.sort(sa), np.arange(n))
    prev = textS[sa - 1]                         # sa == 0 wraps to textS[-1], the last sentinel
    assert np.array_equal(prev, np.asarray(bwt_fwd))
    # LF: row i with symbol c = bwt[i] (a letter) maps to C[c] + rank_c(i)
    b = np.asarray(bwt_fwd)
    cnt = np.bincount(b, minlength=6)
    C = np.zeros(6, dtype=np.int64); C[0] = cnt[5]
    for c in range(1, 5):
        C[c] = C[c - 1] + cnt[c - 1]
    lf = np.full(n, -1, dtype=np.int64)
    for c in range(5):
        idx = np.flatnonzero(b == c)
        lf[idx] = C[c] + np.arange(len(idx))
    m = lf >= 0
    assert np.array_equal(sa[lf[m]], sa[m] - 1)


def check_sa_against_bwt_device(codes, seq_len, bwt_fwd, sa, device="cuda:0", chunk=1 << 27):
    """check_sa_against_bwt with torch on a device, in chunks of rows (indexes of billions of rows): the suffix array is a permutation of
    the sentinel-text positions, the symbols in front of its suffixes are the BWT, and sa[LF(i)] == sa[i] - 1 for every row with a letter"""
    import torch
    seq_len = np.asarray(seq_len, dtype=np.int64)
    n = int(seq_len.sum()) + len(seq_len)
    assert len(sa) == n and len(bwt_fwd) == n and np.asarray(sa).dtype == np.uint32
    textS = torch.full((n,), 5, dtype=torch.uint8, device=device)
    host = torch.from_numpy(np.ascontiguousarray(codes))
    off = s = 0
    for ln in seq_len.tolist():
        textS[s:s + ln] = host[off:off + ln].to(device)
        off += ln
        s += ln + 1
    b = torch.from_numpy(np.ascontiguousarray(bwt_fwd)).to(device)
    sa32 = torch.from_numpy(np.ascontiguousarray(sa).view(np.int32)).to(device)      # (values beyond 2^31 come back through the mask below)
    cnt = torch.zeros(6, dtype=torch.int64, device=device)
    for i in range(0, n, chunk):
        cnt += torch.bincount(b[i:i + chunk].long(), minlength=6)
    cnt = cnt.tolist()
    Cc = [cnt[5]]
    for c in range(1, 5):
        Cc.append(Cc[-1] + cnt[c - 1])
    seen = torch.zeros(n, dtype=torch.bool, device=device)
    base = [0] * 5
    for i in range(0, n, chunk):
        sac = sa32[i:i + chunk].long() & 0xFFFFFFFF
        assert int(sac.max()) < n
        seen[sac] = True
        bc = b[i:i + chunk]
        assert torch.equal(textS[(sac - 1) % n], bc), ("preceding symbols", i)       # row 0's suffix wraps to the last sentinel
        for c in range(5):
            m = bc == c
            k = int(m.sum())
            if k == 0:
                continue
            lf = Cc[c] + base[c] + torch.arange(k, device=device)
            assert torch.equal(sa32[lf].long() & 0xFFFFFFFF, sac[m] - 1), ("LF walk", i, c)
            base[c] += k
        del sac, bc
    assert bool(seen.all()), "not a permutation"


def _unpack_locations(L):
    n = int(L.n_entries)
    ent = []
    for e in range(n):
        key = (int(L.key_seq[e]), int(L.key_pos[e]))
        plus = [(int(L.plus_seq[q]), int(L.plus_pos[q])) for q in range(int(L.plus_off[e]), int(L.plus_off[e + 1]))]
        minus = [(int(L.minus_seq[q]), int(L.minus_pos[q])) for q in range(int(L.minus_off[e]), int(L.minus_off[e + 1]))]
        ent.append((key, plus, minus))
    return ent


def brute_force(codes, seq_len, K, E, revcompl=True, value_bits=16, text_begin=0, text_len=None):
    lib = oracle_lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    seq_len = np.ascontiguousarray(seq_len, dtype=np.uint64)
    if text_len is None:
        text_len = int(seq_len.sum()) - text_begin
    out = np.zeros(text_len, dtype=np.uint8 if value_bits == 8 else np.uint16)
    lib.gmo_brute_force(_ptr(codes), _ptr(seq_len), len(seq_len), text_begin, text_len, K, E, int(revcompl), value_bits, _ptr(out))
    return out


def default_infix_length(K, E, xo=None):
    return oracle_lib().gmo_default_infix_length(K, E, -1 if xo is None else xo)


def format_csv(g, entries, revcompl, csv_intervals=None):
    """Python restatement of saveCsv (src/output.hpp:189-288) used to pin the oracle's location lists
    against the golden csv files.  entries: [((seq,pos), plus, minus)] sorted by key;
    csv_intervals: sorted [(seq_local, begin, end)] or None."""
    # fastaFiles: (file name, cumulative number of sequences - 1)
    fasta, cnt = [], 0
    for name, recs in g.files:
        cnt += len(recs)
        fasta.append((name, cnt - 1))
    out = ['"k-mer"']
    out += [f';"+ strand {n}"' for n, _ in fasta]
    if revcompl:
        out += [f';"- strand {n}"' for n, _ in fasta]
    lines = ["".join(out)]
    for key, plus, minus in entries:
        if csv_intervals is not None:
            if not any(s == key[0] and b <= key[1] < e for s, b, e in csv_intervals):
                continue
        row = f"{key[0]},{key[1]}"
        for locs in ([plus, minus] if revcompl else [plus]):
            i, prev = 0, 0
            for _, last in fasta:
                row += ";"
                first = True
                while i < len(locs) and locs[i][0] <= last:
                    if not first:
                        row += "|"
                    row += f"{locs[i][0] - prev},{locs[i][1]}"
                    first = False
                    i += 1
                prev = last + 1
        lines.append(row)
    return "\n".join(lines) + "\n"


def random_codes(rng, n, dna5=False, n_frac=0.02):
    c = rng.integers(0, 4, size=n, dtype=np.uint8)
    if dna5:
        m = rng.random(n) < n_frac
        c[m] = 4
    return c

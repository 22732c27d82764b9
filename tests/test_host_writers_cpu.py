"""Host logic of the `genmap` program (no GPU): the C++ writers of genmap_amd/host reproduce the reference's
output files byte for byte when fed the golden frequency vectors; FASTA ingestion matches the fixtures' ids.
Golden data: tests/golden/reference_cases (copied data files of /root/reference/tests/test_cases)."""
import ctypes as C
import filecmp
import subprocess

import numpy as np
import pytest

import helpers as H


@pytest.fixture(scope="module")
def hostlib():
    subprocess.check_call(["make", "-s", "-C", str(H.ROOT / "genmap_amd" / "host"), "../lib/libgenmap_host.so"])
    lib = C.CDLL(str(H.ROOT / "genmap_amd" / "lib" / "libgenmap_host.so"))
    lib.gmh_last_error.restype = C.c_char_p
    lib.gmh_save_outputs.restype = C.c_int
    lib.gmh_save_outputs.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_uint32]
    lib.gmh_save_outputs_runs.restype = C.c_int
    lib.gmh_save_outputs_runs.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_uint32]
    lib.gmh_save_csv.restype = C.c_int
    lib.gmh_save_csv.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p,
                                 C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_void_p, C.c_uint32, C.c_int]
    lib.gmh_read_fasta.restype = C.c_int
    lib.gmh_read_fasta.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gmh_default_infix_length.restype = C.c_uint32
    lib.gmh_default_infix_length.argtypes = [C.c_uint32, C.c_uint32, C.c_int32]
    return lib


def _names(ns):
    return b"".join(n.encode() + b"\0" for n in ns)


FORMATS = {  # golden sub-directory -> (kind, format bit, source raw dir, dtype, produced extensions)
    "raw_map": (0, 1, "raw_freq16", np.uint16, [".map"]),
    "raw_freq8": (1, 1, "raw_freq8", np.uint8, [".freq8"]),
    "raw_freq16": (2, 1, "raw_freq16", np.uint16, [".freq16"]),
    "txt_map": (0, 2, "raw_freq16", np.uint16, [".txt"]),
    "txt_freq16": (2, 2, "raw_freq16", np.uint16, [".txt"]),
    "txt_freq8": (1, 2, "raw_freq8", np.uint8, [".txt"]),
    "wig_map": (0, 4, "raw_freq16", np.uint16, [".wig", ".chrom.sizes"]),
    "wig_freq16": (2, 4, "raw_freq16", np.uint16, [".wig", ".chrom.sizes"]),
    "bed_map": (0, 8, "raw_freq16", np.uint16, [".bedgraph"]),
    "bed_freq16": (2, 8, "raw_freq16", np.uint16, [".bedgraph"]),
}


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_writers_reproduce_reference_files(case, hostlib, tmp_path):
    d = H.CASES_DIR / f"case_{case}"
    g, _, _, _ = H.load_case(case)
    checked = 0
    for sub, (kind, bit, src, dt, exts) in FORMATS.items():
        if not (d / sub).is_dir():
            continue
        for name, first, nseq, tb, tl in g.file_slices():
            stem_name = name.rsplit(".", 1)[0] + ".genmap"
            raw = d / src / (stem_name + (".freq8" if dt == np.uint8 else ".freq16"))
            if not raw.exists():
                continue  # selection cases: no output for this fasta file
            c = np.fromfile(raw, dtype=dt)
            assert len(c) == tl
            lens = np.ascontiguousarray(g.seq_len[first:first + nseq], dtype=np.uint64)
            out = tmp_path / sub
            out.mkdir(exist_ok=True)
            rc = hostlib.gmh_save_outputs(H._ptr(c), len(c), c.itemsize, str(out / stem_name).encode(), kind, bit,
                                          _names(g.seq_names[first:first + nseq]), H._ptr(lens), nseq)
            assert rc == 0, hostlib.gmh_last_error()
            for ext in exts:
                assert filecmp.cmp(out / (stem_name + ext), d / sub / (stem_name + ext), shallow=False), (case, sub, name, ext)
                checked += 1
    assert checked >= 10


def _runs_of(c, lens):
    """what gm_map_runs returns: maximal non-zero runs that stay inside one sequence"""
    starts, lengths, values = [], [], []
    base = 0
    for ln in lens:
        k = 0
        while k < ln:
            e = k + 1
            while e < ln and c[base + e] == c[base + k]:
                e += 1
            if c[base + k] != 0:
                starts.append(base + k); lengths.append(e - k); values.append(int(c[base + k]))
            k = e
        base += ln
    return (np.asarray(starts + [0], np.uint64), np.asarray(lengths + [0], np.uint64), np.asarray(values + [0], np.uint16), len(starts))


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_run_length_writers_reproduce_reference_files(case, hostlib, tmp_path):
    d = H.CASES_DIR / f"case_{case}"
    g, _, _, _ = H.load_case(case)
    checked = 0
    for sub in ("wig_map", "wig_freq16", "bed_map", "bed_freq16"):
        kind, bit, src, dt, exts = FORMATS[sub]
        for name, first, nseq, tb, tl in g.file_slices():
            stem_name = name.rsplit(".", 1)[0] + ".genmap"
            raw = d / src / (stem_name + ".freq16")
            if not raw.exists():
                continue
            c = np.fromfile(raw, dtype=dt)
            lens = np.ascontiguousarray(g.seq_len[first:first + nseq], dtype=np.uint64)
            st, ln, va, n = _runs_of(c, [int(x) for x in lens])
            out = tmp_path / sub
            out.mkdir(exist_ok=True)
            rc = hostlib.gmh_save_outputs_runs(n, H._ptr(st), H._ptr(ln), H._ptr(va), str(out / stem_name).encode(), kind, bit,
                                               _names(g.seq_names[first:first + nseq]), H._ptr(lens), nseq)
            assert rc == 0, hostlib.gmh_last_error()
            for ext in exts:
                assert filecmp.cmp(out / (stem_name + ext), d / sub / (stem_name + ext), shallow=False), (case, sub, name, ext)
                checked += 1
    assert checked >= 4


def _runs_vectorised(c, lens):
    c = np.asarray(c); n = len(c)
    head = np.ones(n, bool); head[1:] = c[1:] != c[:-1]
    head[np.cumsum(lens)[:-1]] = True
    st = np.flatnonzero(head); ln = np.diff(np.append(st, n)); va = c[st]
    keep = va != 0
    z = np.zeros(1, np.uint64)
    return (np.concatenate([st[keep].astype(np.uint64), z]), np.concatenate([ln[keep].astype(np.uint64), z]),
            np.concatenate([va[keep].astype(np.uint16), z.astype(np.uint16)]), int(keep.sum()))


@pytest.mark.parametrize("kind", [0, 2])   # mappability (1/v as %g), 16-bit frequencies
def test_run_length_writers_on_all_cores_equal_the_serial_ones(kind, hostlib, tmp_path):
    """run lists long enough for the threaded formatter (several threads, unequal shares, sequence boundaries inside a share,
    a span repeated across a share boundary) against the dense single-threaded writers of the same vector"""
    rng = np.random.default_rng(5 + kind)
    lens = np.array([250_001, 7, 190_000, 1, 359_991], dtype=np.uint64)
    n = int(lens.sum())
    c = np.ascontiguousarray(np.repeat(rng.integers(0, 4, size=n), rng.integers(1, 4, size=n))[:n].astype(np.uint16))
    assert len(c) == n
    c[1000:3000] = 7; c[300_000:300_050] = 65535
    names = ["chrA", "s 2", "chrB", "x", "tail"]
    st, ln, va, nr = _runs_vectorised(c, [int(x) for x in lens])
    assert nr > (1 << 17)
    for bit, exts in ((4, [".wig", ".chrom.sizes"]), (8, [".bedgraph"]), (16, [".bed"])):
        a, b = tmp_path / f"dense{bit}", tmp_path / f"runs{bit}"
        a.mkdir(); b.mkdir()
        assert hostlib.gmh_save_outputs(H._ptr(c), n, 2, str(a / "o").encode(), kind, bit, _names(names), H._ptr(lens), len(lens)) == 0
        assert hostlib.gmh_save_outputs_runs(nr, H._ptr(st), H._ptr(ln), H._ptr(va), str(b / "o").encode(), kind, bit, _names(names), H._ptr(lens), len(lens)) == 0
        for ext in exts:
            assert filecmp.cmp(a / ("o" + ext), b / ("o" + ext), shallow=False), (kind, bit, ext)
    # the dense txt / raw writers, also threaded: against a restatement in Python
    t = tmp_path / "txt"; t.mkdir()
    assert hostlib.gmh_save_outputs(H._ptr(c), n, 2, str(t / "o").encode(), kind, 3, _names(names), H._ptr(lens), len(lens)) == 0
    inv = np.zeros(65536, np.float32); inv[1:] = np.float32(1) / np.arange(1, 65536, dtype=np.float32)
    text = np.array([("%g" % inv[v]) if kind == 0 else str(v) for v in range(65536)], dtype=object)
    exp, pos = [], 0
    for nm, ln in zip(names, lens):
        exp.append(">" + nm + "\n" + " ".join(text[c[pos:pos + int(ln)]]) + "\n"); pos += int(ln)
    assert (t / "o.txt").read_text() == "".join(exp)
    if kind == 0:
        assert np.array_equal(np.fromfile(t / "o.map", dtype=np.float32), inv[c])
    else:
        assert np.array_equal(np.fromfile(t / "o.freq16", dtype=np.uint16), c)


def test_wig_run_writer_with_zero_runs_at_share_boundaries(hostlib, tmp_path):
    """A run list may hold zero runs (selections leave them inside a sequence).  The serial writer skips them without touching
    last_occ (src/output.hpp:88-113); the threaded one has to seed every share from the nearest NON-ZERO run before it.
    Pattern: (1 x1)(0 x2)(2 x2)(0 x1)...: every zero run is as long as the run after it and the non-zero run before it is not,
    so a share that starts behind a zero run and seeds from it would drop a `variableStep` header."""
    reps = 1 << 16
    c = np.ascontiguousarray(np.tile(np.array([1, 0, 0, 2, 2, 0], np.uint16), reps))
    n = len(c)
    lens = np.array([n // 2 + 3, n - n // 2 - 3], dtype=np.uint64)   # a sequence boundary inside a run
    head = np.ones(n, bool); head[1:] = c[1:] != c[:-1]; head[int(lens[0])] = True
    st = np.flatnonzero(head); ln = np.diff(np.append(st, n)); va = c[st]
    assert len(st) >= (1 << 18) and (va == 0).sum() >= (1 << 17)
    z = np.zeros(1, np.uint64)
    st, ln, va = np.concatenate([st.astype(np.uint64), z]), np.concatenate([ln.astype(np.uint64), z]), np.concatenate([va.astype(np.uint16), z.astype(np.uint16)])
    names = ["one", "two"]
    for kind in (0, 2):
        a, b = tmp_path / f"dense{kind}", tmp_path / f"runs{kind}"
        a.mkdir(); b.mkdir()
        assert hostlib.gmh_save_outputs(H._ptr(c), n, 2, str(a / "o").encode(), kind, 4 | 8 | 16, _names(names), H._ptr(lens), 2) == 0
        assert hostlib.gmh_save_outputs_runs(len(st) - 1, H._ptr(st), H._ptr(ln), H._ptr(va), str(b / "o").encode(), kind, 4 | 8 | 16, _names(names), H._ptr(lens), 2) == 0
        for ext in (".wig", ".bedgraph", ".bed"):
            assert filecmp.cmp(a / ("o" + ext), b / ("o" + ext), shallow=False), (kind, ext)


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_csv_writer_reproduces_reference_files(case, hostlib, tmp_path):
    """oracle location lists -> gm_locate's CSR layout -> C++ csv writer == golden csv"""
    d = H.CASES_DIR / f"case_{case}"
    g, directory, fl, bed = H.load_case(case)
    ix = H.OracleIndex(g.codes, g.seq_len, keep_sa=True)
    rc_strand = not fl.get("nc", False)
    file_names = [n for n, _ in g.files]
    spf = np.asarray([len(r) for _, r in g.files], dtype=np.uint64)
    for name, first, nseq, tb, tl in g.file_slices():
        iv = None
        if bed is not None:
            iv = H.slice_intervals(g, first, nseq, bed)
            if not iv:
                continue
        _, _, locs = ix.mappability(fl["K"], fl["E"], text_begin=tb, text_len=tl, first_seq=first, n_seq=nseq, revcompl=rc_strand,
                                    directory=True, csv=True, intervals=iv, seq_file_id=g.seq_file)
        # CSR over slice positions
        po = np.zeros(tl + 1, dtype=np.uint64); mo = np.zeros(tl + 1, dtype=np.uint64)
        plus, minus = [], []
        by_pos = {}
        for (s, p), pl, mi in locs:
            by_pos[int(g.cum[first + s] - g.cum[first]) + p] = (pl, mi)
        for j in range(tl):
            pl, mi = by_pos.get(j, ([], []))
            plus += [(a << 32) | b for a, b in pl]; minus += [(a << 32) | b for a, b in mi]
            po[j + 1] = len(plus); mo[j + 1] = len(minus)
        plus = np.asarray(plus + [0], dtype=np.uint64); minus = np.asarray(minus + [0], dtype=np.uint64)
        lens = np.ascontiguousarray(g.seq_len[first:first + nseq], dtype=np.uint64)
        stem = tmp_path / (name.rsplit(".", 1)[0] + ".genmap")
        rc = hostlib.gmh_save_csv(str(stem).encode(), 0, tl, H._ptr(po), H._ptr(mo), H._ptr(plus), H._ptr(minus),
                                  _names(g.seq_names[first:first + nseq]), H._ptr(lens), nseq, fl["K"], int(rc_strand),
                                  _names(file_names), H._ptr(spf), len(file_names), 0)
        assert rc == 0, hostlib.gmh_last_error()
        assert (tmp_path / (stem.name + ".csv")).read_bytes() == (d / "csv" / (stem.name + ".csv")).read_bytes(), (case, name)


def test_csv_writer_on_all_cores_equals_a_restatement(hostlib, tmp_path):
    """windows long enough for the threaded formatter (shares that start inside a sequence, positions without hits, k-mers that
    would span two sequences, windows appended one after the other) against a restatement of src/output.hpp:189-288"""
    rng = np.random.default_rng(11)
    lens = np.array([30_011, 17, 25_000, 9, 14_963], dtype=np.uint64)
    seqs_per_file = np.array([2, 1, 2], dtype=np.uint64)
    file_names = ["a.fa", "b.fasta", "c.fa"]
    seq_names = ["s0", "s one", "s2", "t", "u"]
    K, n = 21, int(lens.sum())
    cum = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    last_of_file = np.cumsum(seqs_per_file) - 1

    def lists(count):
        out = []
        for c in count:
            sq = np.sort(rng.integers(0, len(lens), size=c))
            out.append([(int(a), int(rng.integers(0, 1 << 20))) for a in sq])
        return out
    cnt_p = rng.integers(0, 4, size=n) * (rng.random(n) < 0.7); cnt_m = rng.integers(0, 3, size=n) * (rng.random(n) < 0.5)
    plus, minus = lists(cnt_p), lists(cnt_m)

    def expect(j0, j1, header):
        rows = []
        if header:
            rows.append('"k-mer"' + "".join(f';"+ strand {f}"' for f in file_names) + "".join(f';"- strand {f}"' for f in file_names) + "\n")
        for j in range(j0, j1):
            if not plus[j] and not minus[j]:
                continue
            s = int(np.searchsorted(cum, j, side="right") - 1)
            off = j - cum[s]
            if off > int(lens[s]) - K:
                continue
            row = f"{s},{off}"
            for lst in (plus[j], minus[j]):
                prev = 0
                for fi, last in enumerate(last_of_file):
                    row += ";" + "|".join(f"{a - prev},{b}" for a, b in lst if prev <= a <= last)
                    prev = int(last) + 1
            rows.append(row + "\n")
        return "".join(rows)

    stem = tmp_path / "x.genmap"
    exp = ""
    for w, (j0, j1) in enumerate(((0, 40_000), (40_000, 40_003), (40_003, n))):
        po = np.zeros(j1 - j0 + 1, np.uint64); mo = np.zeros(j1 - j0 + 1, np.uint64)
        po[1:] = np.cumsum([len(plus[j]) for j in range(j0, j1)]); mo[1:] = np.cumsum([len(minus[j]) for j in range(j0, j1)])
        pl = np.asarray([(a << 32) | b for j in range(j0, j1) for a, b in plus[j]] + [0], dtype=np.uint64)
        mi = np.asarray([(a << 32) | b for j in range(j0, j1) for a, b in minus[j]] + [0], dtype=np.uint64)
        assert hostlib.gmh_save_csv(str(stem).encode(), j0, j1 - j0, H._ptr(po), H._ptr(mo), H._ptr(pl), H._ptr(mi), _names(seq_names), H._ptr(lens), len(lens), K, 1,
                                    _names(file_names), H._ptr(seqs_per_file), len(file_names), int(w > 0)) == 0, hostlib.gmh_last_error()
        exp += expect(j0, j1, w == 0)
    assert (tmp_path / "x.genmap.csv").read_text() == exp


def test_fasta_reader_matches_fixture_parsing(hostlib):
    for case in ("1c", "2c", "3a"):
        d = H.CASES_DIR / f"case_{case}"
        for fa in sorted(d.glob("*.fa")):
            n, tot, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
            assert hostlib.gmh_read_fasta(str(fa).encode(), None, None, None, 0, C.byref(n), C.byref(tot), C.byref(nb)) == 0
            codes = np.zeros(tot.value, np.uint8); lens = np.zeros(n.value, np.uint64); names = C.create_string_buffer(nb.value + 1)
            assert hostlib.gmh_read_fasta(str(fa).encode(), H._ptr(codes), H._ptr(lens), names, nb.value + 1, C.byref(n), C.byref(tot), C.byref(nb)) == 0
            ref = H.read_fasta(fa)
            assert names.raw[:nb.value].split(b"\0")[:-1] == [r[0].encode() for r in ref]
            assert np.array_equal(codes, np.concatenate([r[1] for r in ref]))


def test_host_infix_rule(hostlib):
    for K in (3, 4, 8, 24, 30, 100, 128):
        for E in range(5):
            for xo in (None, 0, 1, 3):
                b = H.default_infix_length(K, E, xo)
                assert hostlib.gmh_default_infix_length(K, E, -1 if xo is None else xo) == (0 if b < 0 else b)

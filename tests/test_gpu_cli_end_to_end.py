"""The reference's end-to-end harness (/root/reference/tests/tests.sh + tests/CMakeLists.txt:27-73) against the
`genmap` program of this build: `genmap index` (GPU suffix sort) + `genmap map` with the flags of every case,
every output format and every -xo rerun; all produced files must equal tests/golden/reference_cases."""
import filecmp
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

GENMAP = H.ROOT / "genmap_amd" / "bin" / "genmap"

FORMAT_FLAGS = {  # tests/CMakeLists.txt:29-52
    "raw_map": ["-r"], "raw_freq8": ["-r", "-fs"], "raw_freq16": ["-r", "-fl"], "txt_map": ["-t"], "txt_freq16": ["-t", "-fl"],
    "txt_freq8": ["-t", "-fs"], "wig_map": ["-w"], "wig_freq16": ["-w", "-fl"], "bed_map": ["-bg"], "bed_freq16": ["-bg", "-fl"], "csv": ["-d"],
}


def _run_checked(cmd):
    """check_call that shows what the process said.  No retry: in round 4 one `genmap index` of ~350 process starts of a suite run died
    with SIGSEGV on the GPU box and a silent rerun hid it.  A death by signal now fails the test at once, and the program's crash
    handler (genmap_main.cpp: genmap_crash_handler) has written the faulting thread's stack to stderr, which the assertion shows;
    tools/crash_hunt.sh loops the suspected commands thousands of times (plain and under AddressSanitizer) to catch it in the act."""
    r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:   # the whole of what the process said (pytest -q cuts the assertion's tuple short): printed, and kept for the round's evidence
        print(f"FAILED {cmd}: exit {r.returncode}\n{r.stderr}", flush=True)
        try:
            out = H.ROOT / "gpurun_out"; out.mkdir(exist_ok=True)
            with open(out / "cli_failures.txt", "a") as f:
                f.write(f"== {cmd}: exit {r.returncode}\n{r.stderr}\n")
        except OSError:
            pass
    assert r.returncode == 0, (cmd, r.returncode, r.stderr[-4000:])


def _same_tree(a, b):
    cmp = filecmp.dircmp(a, b)
    assert not cmp.left_only and not cmp.right_only, (cmp.left_only, cmp.right_only)
    for f in cmp.common_files:
        assert filecmp.cmp(a / f, b / f, shallow=False), f


@pytest.mark.parametrize("sampling", [None, 10, 3])
@pytest.mark.parametrize("case", sorted(H.CASES))
def test_cli_reproduces_reference_outputs(case, sampling, tmp_path):
    """sampling None: the default index (full suffix array).  -S 10 (the reference's default, src/indexing.hpp:311-315) and -S 3:
    the sampled array on disk and in HBM, locate by LF walk; run on the outputs that locate (csv, --exclude-pseudo) plus one
    frequency format."""
    assert GENMAP.exists(), "genmap binary not built (python -c 'import __graft_entry__ as g; g.build()')"
    d = H.CASES_DIR / f"case_{case}"
    directory, fl = H.CASES[case]
    idx = tmp_path / "index"
    if directory:
        src = tmp_path / "fastas"
        src.mkdir()
        for f in d.glob("*.fa"):
            shutil.copy(f, src / f.name)
    smp = [] if sampling is None else ["-S", str(sampling)]
    if directory:
        _run_checked([str(GENMAP), "index", "-FD", str(src), "-I", str(idx), "-A", "skew"] + smp)
    else:
        _run_checked([str(GENMAP), "index", "-F", str(d / "genome.fa"), "-I", str(idx), "-A", "divsufsort"] + smp)
    if sampling is not None:
        assert f"sampling_rate:{sampling}\n" in (idx / "index.info").read_text()
        assert (idx / "index.sa.samples").exists() and not (idx / "index.sa").exists()
    flags = ["-E", str(fl["E"]), "-K", str(fl["K"])] + (["-nc"] if fl.get("nc") else []) + (["-ep"] if fl.get("ep") else [])
    if (d / "subset.bed").exists():
        flags += ["-S", str(d / "subset.bed")]
    runs = []
    for sub, ff in FORMAT_FLAGS.items():
        if not (d / sub).is_dir():
            continue
        if sampling is not None and sub not in ("csv", "raw_freq16", "txt_map") and not fl.get("ep"):
            continue
        for xo in H.xo_variants(case):
            out = tmp_path / f"out_{sub}_{xo}"
            out.mkdir()
            runs.append((sub, out, [str(GENMAP), "map", "-I", str(idx), "-O", str(out)] + flags + ff + (["-xo", str(xo)] if xo is not None else [])))
    # the map runs of one case are independent processes reading one index directory: six at a time (a run is mostly process start
    # and HIP initialisation -- one after the other they were half of the GPU suite's wall time)
    with ThreadPoolExecutor(max_workers=6) as pool:
        rcs = list(pool.map(lambda r: subprocess.run(r[2], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True), runs))
    for (sub, out, cmd), r in zip(runs, rcs):
        assert r.returncode == 0, (cmd, r.returncode, r.stderr[-2000:])
        _same_tree(out, d / sub)


def test_cli_error_paths(tmp_path):
    d = H.CASES_DIR / "case_1a"
    idx = tmp_path / "index"
    subprocess.check_call([str(GENMAP), "index", "-F", str(d / "genome.fa"), "-I", str(idx)], stdout=subprocess.DEVNULL)
    out = tmp_path / "o"; out.mkdir()
    base = [str(GENMAP), "map", "-I", str(idx), "-O", str(out), "-K", "3", "-E", "0"]
    r = subprocess.run(base, capture_output=True, text=True)                         # no output format
    assert r.returncode != 0 and "Please choose at least one output format" in r.stderr
    r = subprocess.run(base + ["-r", "-fs", "-fl"], capture_output=True, text=True)  # -fs and -fl
    assert r.returncode != 0 and "Cannot use both" in r.stderr
    r = subprocess.run(base[:-1] + ["5", "-r"], capture_output=True, text=True)      # E > 4
    assert r.returncode != 0 and "E > 4 not yet supported." in r.stderr
    r = subprocess.run(base + ["-r", "-xo", "3"], capture_output=True, text=True)    # overlap too large
    assert r.returncode != 0 and "overlap cannot be larger than" in r.stderr
    r = subprocess.run([str(GENMAP), "index", "-F", str(d / "genome.fa"), "-I", str(idx)], capture_output=True, text=True)
    assert r.returncode != 0 and "already exists" in r.stderr
    # -O as a file name for single-fasta indices (src/mappability.hpp:562-619)
    subprocess.check_call(base[:4] + ["-O", str(tmp_path / "named"), "-K", "3", "-E", "0", "-nc", "-r", "-fl"], stdout=subprocess.DEVNULL)
    assert filecmp.cmp(tmp_path / "named.freq16", d / "raw_freq16" / "genome.genmap.freq16", shallow=False)


@pytest.mark.parametrize("case", ["1f", "2d", "3b"])
def test_cli_multi_device_shards(case, tmp_path):
    """-D a,b: one replica and one host thread per listed device, shards merged on the host.  The GPU box has one
    device, so it is listed twice; the data path is the one eight devices would use."""
    d = H.CASES_DIR / f"case_{case}"
    directory, fl = H.CASES[case]
    idx = tmp_path / "index"
    if directory:
        src = tmp_path / "fastas"; src.mkdir()
        for f in d.glob("*.fa"):
            shutil.copy(f, src / f.name)
        subprocess.check_call([str(GENMAP), "index", "-FD", str(src), "-I", str(idx)], stdout=subprocess.DEVNULL)
    else:
        subprocess.check_call([str(GENMAP), "index", "-F", str(d / "genome.fa"), "-I", str(idx)], stdout=subprocess.DEVNULL)
    flags = ["-E", str(fl["E"]), "-K", str(fl["K"])] + (["-nc"] if fl.get("nc") else [])
    for sub in ("raw_freq16", "wig_freq16", "txt_map", "csv"):
        out = tmp_path / f"out_{sub}"; out.mkdir()
        subprocess.check_call([str(GENMAP), "map", "-I", str(idx), "-O", str(out), "-D", "0,0,0"] + flags + FORMAT_FLAGS[sub], stdout=subprocess.DEVNULL)
        _same_tree(out, d / sub)


@pytest.mark.parametrize("case", ["1f", "3b"])
def test_cli_sampled_suffix_array_with_64_bit_rows(case, tmp_path):
    """the reference's default -S 10 on an index with 64-bit rows (forced here with --block-bytes 65600 = 64 | GM_BLOCK_WIDE_ROWS,
    what 2^32 - 1 rows or more get by themselves): index.sa.samples holds 8-byte entries; csv and a frequency format"""
    d = H.CASES_DIR / f"case_{case}"
    directory, fl = H.CASES[case]
    idx = tmp_path / "index"
    wide = ["--block-bytes", "65600"]
    if directory:
        src = tmp_path / "fastas"; src.mkdir()
        for f in d.glob("*.fa"):
            shutil.copy(f, src / f.name)
        subprocess.check_call([str(GENMAP), "index", "-FD", str(src), "-I", str(idx), "-S", "10"] + wide, stdout=subprocess.DEVNULL)
    else:
        subprocess.check_call([str(GENMAP), "index", "-F", str(d / "genome.fa"), "-I", str(idx), "-S", "10"] + wide, stdout=subprocess.DEVNULL)
    assert (idx / "index.sa.samples").exists() and (idx / "index.sa.samples").stat().st_size % 8 == 0
    flags = ["-E", str(fl["E"]), "-K", str(fl["K"])] + (["-nc"] if fl.get("nc") else []) + (["-ep"] if fl.get("ep") else [])
    for sub in ("csv", "raw_freq16"):
        if not (d / sub).is_dir():
            continue
        out = tmp_path / f"out_{sub}"; out.mkdir()
        subprocess.check_call([str(GENMAP), "map", "-I", str(idx), "-O", str(out)] + flags + FORMAT_FLAGS[sub] + wide, stdout=subprocess.DEVNULL)
        _same_tree(out, d / sub)


def test_cli_two_device_threads_k100_e1_at_0p77_gbp(tmp_path):
    """The C++ host path of a multi-GPU run at a size where it matters (a quarter of the metric's text, 0.77 Gbp in 24 sequences):
    `genmap index` writes the directory, `genmap map -K 100 -E 1 -r -fs -D 0,0` = config C4's (k, e) with one replica and one host thread
    per listed device (the box has one: listed twice), interleaved chunks copied straight into the one pinned result vector -- the raw
    file must equal the library's single-call result on an index built in this process."""
    import numpy as np
    import genmap_amd as g
    from genmap_amd import synth
    codes, lens, _ = synth.workload("grch38", 0.25)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    fa = tmp_path / "genome.fa"
    off = 0
    with open(fa, "wb") as f:
        for k, ln in enumerate(lens):
            f.write(b">seq%d synthetic\n" % (k + 1))
            seq = lut[codes[off:off + ln]]
            off += ln
            n = len(seq) // 60 * 60
            if n:
                rows = np.empty((n // 60, 61), dtype=np.uint8)
                rows[:, :60] = seq[:n].reshape(-1, 60)
                rows[:, 60] = 10
                f.write(rows.tobytes())
            if n < len(seq):
                f.write(seq[n:].tobytes() + b"\n")
    idx, out = tmp_path / "index", tmp_path / "out"
    out.mkdir()
    _run_checked([str(GENMAP), "index", "-F", str(fa), "-I", str(idx)])
    _run_checked([str(GENMAP), "map", "-I", str(idx), "-O", str(out), "-K", "100", "-E", "1", "-r", "-fs", "-D", "0,0"])
    got = np.fromfile(out / "genome.genmap.freq8", dtype=np.uint8)
    shutil.rmtree(idx)
    ix = g.Index.build(codes, lens, sampling=1)
    try:
        assert np.array_equal(got, ix.map(100, 1, value_bits=8))
    finally:
        ix.close()


def test_cli_long_kmers_k300(tmp_path):
    """`genmap map -K 300` (the reference takes any -K, src/mappability.hpp:425-426): the program goes through the long k-mer kernel
    (gm_longk.h); raw 16-bit frequencies with the library's block shape and with an explicit -xo 290 (blocks of 291 k-mers in the reference:
    clamped to 255 here, the result does not depend on it), against the oracle."""
    import numpy as np
    rng = np.random.default_rng(300)
    lens = [30000, 500, 9000]
    codes = rng.integers(0, 4, size=sum(lens), dtype=np.uint8)
    codes[20000:22000] = codes[3000:5000]            # a long exact copy
    codes[31000:32500] = codes[3200:4700]; codes[31700] ^= 1   # and one a substitution away, in another sequence
    codes[8000:8040] = 4
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    fa = tmp_path / "genome.fa"
    off = 0
    with open(fa, "wb") as f:
        for k, ln in enumerate(lens):
            f.write(b">s%d\n" % k)
            f.write(lut[codes[off:off + ln]].tobytes() + b"\n")
            off += ln
    idx = tmp_path / "index"
    _run_checked([str(GENMAP), "index", "-F", str(fa), "-I", str(idx)])
    ora = H.OracleIndex(codes, lens, keep_sa=False)
    for E, extra in ((0, []), (1, []), (1, ["-xo", "290"])):
        exp = ora.mappability(300, E, value_bits=16, threads=4)
        assert exp.max() >= 2
        out = tmp_path / f"out_{E}_{len(extra)}"; out.mkdir()
        _run_checked([str(GENMAP), "map", "-I", str(idx), "-O", str(out), "-K", "300", "-E", str(E), "-r", "-fl"] + extra)
        got = np.fromfile(out / "genome.genmap.freq16", dtype=np.uint16)
        assert np.array_equal(got, exp), (E, extra)

"""The tools that turn rocprofv3 output into the tables under profiles/ (tools/passes.py) and the host figures bench.py reports
(physical cores): plain host logic, checked on synthetic input."""
import importlib.util
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


WALKER = "void gm::search_kernel_w4<1, gm::CountEnv<1, 2>, true>(gm::SearchArgs)"
ONE_LOOP = "void gm::search_kernel_w4<1, gm::CountEnv<1, 1>, true>(gm::SearchArgs)"
CORR = "void gm::search_kernel_w4<1, gm::ScatterEnv<1>, true>(gm::SearchArgs)"
EXPAND = "gm::expand_kernel(gm::SearchArgs)"
FIN = "void gm::finalize_kernel<unsigned char>(unsigned int const*, unsigned char*, unsigned long)"


def test_passes_groups_the_dispatches_of_a_call_up_to_its_finalize_kernel():
    p = _load("passes", ROOT / "tools" / "passes.py")
    t = [0]

    def row(name, ms):
        t[0] += 1000
        r = {"Kernel_Name": name, "Start_Timestamp": str(t[0]), "End_Timestamp": str(t[0] + int(ms * 1e6))}
        t[0] += int(ms * 1e6)
        return r
    rows = []
    # pass 0: a split search of three slices (the last walker finds no packets) and its correction pass
    rows += [row(CORR, 2.0), row(EXPAND, 10.0), row(WALKER, 30.0), row(EXPAND, 11.0), row(WALKER, 31.0), row(EXPAND, 0.05), row(WALKER, 0.1), row(FIN, 1.0)]
    # pass 1: a one-loop kernel
    rows += [row(ONE_LOOP, 50.0), row(FIN, 1.0)]
    # pass 2: nothing but the finalize kernel (an empty share)
    rows += [row(FIN, 1.0)]
    ps = p.passes(list(reversed(rows)), lambda r: int(r["Start_Timestamp"]), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)   # (sorted inside)
    assert len(ps) == 3
    assert [round(x, 3) for x in ps[0]["phase_a"]] == [10.0, 11.0, 0.05] and [round(x, 3) for x in ps[0]["walker"]] == [30.0, 31.0, 0.1]
    assert [round(x, 3) for x in ps[0]["correction"]] == [2.0] and not ps[0]["one_loop"]
    assert [round(x, 3) for x in ps[1]["one_loop"]] == [50.0] and not ps[1]["walker"] and not ps[1]["phase_a"]
    assert not any(ps[2].values())


def test_passes_sums_a_counter_per_pass_and_leaves_the_correction_pass_out():
    p = _load("passes", ROOT / "tools" / "passes.py")
    rows, d = [], [0]

    def row(name, counter, value):
        d[0] += 1
        return {"Kernel_Name": name, "Dispatch_Id": str(d[0]), "Counter_Name": counter, "Counter_Value": str(value)}
    for rep in range(2):
        rows += [row(CORR, "FETCH_SIZE", 7), row(EXPAND, "FETCH_SIZE", 100), row(WALKER, "FETCH_SIZE", 300), row(EXPAND, "FETCH_SIZE", 1), row(WALKER, "FETCH_SIZE", 2), row(FIN, "FETCH_SIZE", 5)]
    ps = p.passes(rows, lambda r: int(r["Dispatch_Id"]), lambda r: float(r["Counter_Value"]))
    assert len(ps) == 2
    for q in ps:
        assert sum(q["phase_a"]) == 101 and sum(q["walker"]) == 302 and sum(q["correction"]) == 7 and not q["one_loop"]


def test_bench_reports_physical_cores_not_hardware_threads():
    sys.path.insert(0, str(ROOT))
    bench = _load("bench_for_test", ROOT / "bench.py")
    c = bench.physical_cores()
    assert 1 <= c <= (os.cpu_count() or 1)
    ids = set()
    phys = core = None
    for line in open("/proc/cpuinfo"):
        if line.startswith("physical id"):
            phys = line.split(":")[1].strip()
        elif line.startswith("core id"):
            core = line.split(":")[1].strip()
        elif not line.strip():
            if phys is not None and core is not None:
                ids.add((phys, core))
            phys = core = None
    if ids:
        assert c == len(ids)


def _kmer_keys(codes, K):
    """2-bit packed K-mers (K <= 31) of an N-free stretch model: windows with an N are dropped"""
    import numpy as np
    c = codes.astype(np.uint64)
    n = len(c) - K + 1
    key = np.zeros(n, dtype=np.uint64)
    bad = np.zeros(n, dtype=bool)
    for i in range(K):
        key = (key << np.uint64(2)) | (c[i:i + n] & np.uint64(3))
        bad |= codes[i:i + n] > 3
    return key[~bad]


def test_hard_workload_shares_its_families_between_sequences_and_strands():
    """synth.workload("grch38h"): the same 24 lengths as S3, but the repeat families are common to ALL sequences and half of
    the copies are reverse-complemented -- S3 seeds every sequence by itself and plants forward copies only.  Checked where it shows: 24-mers that
    two different sequences have in common, on either strand."""
    import numpy as np
    sys.path.insert(0, str(ROOT))
    from genmap_amd import synth
    K = 24
    easy, lens_e, _ = synth.workload("grch38", 0.001)
    hard, lens_h, desc = synth.workload("grch38h", 0.001)
    assert lens_e == lens_h and len(lens_h) == 24 and "S3h" in desc
    again, _, _ = synth.workload("grch38h", 0.001)
    assert np.array_equal(hard, again)                                   # deterministic
    assert (hard <= 4).all() and (hard == 4).any()

    def shared(text, lens, rc):
        cum = np.concatenate([[0], np.cumsum(lens)])
        a = text[cum[0]:cum[1]]
        b = text[cum[1]:cum[2]]
        if rc:
            b = np.where(b[::-1] < 4, 3 - b[::-1], 4).astype(np.uint8)
        return len(np.intersect1d(_kmer_keys(a, K), _kmer_keys(b, K)))
    fwd_e, rc_e = shared(easy, lens_e, False), shared(easy, lens_e, True)
    fwd_h, rc_h = shared(hard, lens_h, False), shared(hard, lens_h, True)
    assert fwd_h > 20 * max(fwd_e, 1) and rc_h > 20 * max(rc_e, 1), (fwd_e, rc_e, fwd_h, rc_h)
    assert rc_h > fwd_h // 10 and fwd_h > rc_h // 10                     # both orientations, neither a rarity

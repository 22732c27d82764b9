"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/genmap_amd.h
declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    txt = (ROOT / "include" / "genmap_amd.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import genmap_amd as g
    lib = g.load_library()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(g.capi.EXPORTS)


def test_no_cpu_fallback_without_device():
    import genmap_amd as g
    if g.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(g.GenmapError) as e:
        g.Index.build(np.zeros(100, np.uint8), [100])
    assert e.value.status == -1


def test_default_infix_matches_oracle_rule():
    import genmap_amd as g
    import helpers as H
    for K in (3, 4, 8, 24, 30, 36, 50, 100, 101, 128, 150, 255):
        for E in range(5):
            for xo in (None, 0, 1, 2, 5):
                a = g.default_infix_length(K, E, xo)
                b = H.default_infix_length(K, E, xo)
                assert a == (0 if b < 0 else b), (K, E, xo, a, b)


def test_product_does_not_reference_oracle():
    """The shipped sources never include, link or load anything under oracle/ or tests/."""
    for p in list((ROOT / "genmap_amd").rglob("*")) + [ROOT / "include" / "genmap_amd.h"]:
        if p.is_file() and p.suffix in (".py", ".h", ".hip", ".cpp", ".hpp") or p.name == "Makefile":
            if not p.is_file():
                continue
            t = p.read_text(errors="ignore")
            assert "gm_oracle" not in t and "libgmoracle" not in t and "libgmemu" not in t, p


def test_tuned_infix_is_always_schedulable():
    """the GPU-tuned block shape keeps at least one symbol per OSS block and never exceeds K (any K, E the ABI accepts)"""
    import genmap_amd as g
    nb = [1, 2, 4, 5, 6]
    for K in range(1, 256):
        for E in range(5):
            t = g.tuned_infix_length(K, E)
            if K < nb[E]:
                continue  # the call is rejected with GM_ERR_BAD_OVERLAP (infix shorter than the number of blocks)
            assert nb[E] <= t <= K, (K, E, t)
    assert g.tuned_infix_length(30, 5) == 0
    # k-mers longer than 255 (gm_longk.h): blocks of 48 .. 255 k-mers, never more (a block list holds the count in 8 bits)
    for K in (256, 300, 1000, 1021, 5000, 32768):
        for E in range(5):
            t = g.tuned_infix_length(K, E)
            assert nb[E] <= t <= K and 48 <= K - t + 1 <= 255, (K, E, t)
            assert g.tuned_infix_length(K, E, locating=True) == t
    assert g.tuned_infix_length(32769, 0) == 0


def test_tuned_block_shape_at_one_error_follows_the_measured_rule():
    """e = 1 (profiles/r02/sweep_grch38_steps*.txt): from K = 44 the first, exact OSS part keeps at least 17 characters
    (infix >= 35); up to K = 112 the window K + n - 1 fits five 32-symbol LDS chunks; blocks never shrink as K grows past
    the cliffs except where the window bound bites; BASELINE shapes are pinned"""
    import genmap_amd as g
    n = lambda K: K - g.tuned_infix_length(K, 1) + 1
    assert (n(24), n(30), n(50), n(64), n(100), n(150), n(250)) == (8, 8, 16, 16, 24, 48, 48)   # K <= 43: re-measured with jump patterns (r03)
    assert g.tuned_infix_length(24, 1, locating=True) == 24 and g.tuned_infix_length(30, 2, locating=True) == 30 and g.tuned_infix_length(100, 1, locating=True) == 97
    assert g.tuned_infix_length(30, 0, locating=True) == 30 and g.tuned_infix_length(100, 0, locating=True) == 97 and g.tuned_infix_length(3, 2, locating=True) >= 3
    assert g.tuned_infix_length(30, 0) == 17 and g.tuned_infix_length(30, 2) == 25 and g.tuned_infix_length(24, 1) == 17
    for K in range(44, 256):
        assert g.tuned_infix_length(K, 1) >= 35, K
        if K <= 112:
            assert K + n(K) - 1 <= 127, K
        assert 5 <= n(K) <= 48


def test_every_test_runs_under_a_time_limit(tmp_path):
    """tests/conftest.py bounds every test (SIGALRM + a faulthandler watchdog): a hung kernel fails ITS test within minutes instead of
    burning the whole `pytest -x` step.  Checked on a throw-away test that sleeps past a 1-second limit."""
    import shutil
    import subprocess
    import sys
    shutil.copy(Path(__file__).resolve().parent / "conftest.py", tmp_path / "conftest.py")
    (tmp_path / "test_sleepy.py").write_text(
        "import time, pytest\n"
        "@pytest.mark.time_limit(1)\n"
        "def test_sleeps():\n"
        "    time.sleep(30)\n"
        "def test_fine():\n"
        "    assert True\n")
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert time.time() - t0 < 25, "the limit did not interrupt the sleeping test"
    assert r.returncode != 0 and "exceeded its time limit of 1 s" in r.stdout + r.stderr, r.stdout[-2000:]
    assert "1 failed, 1 passed" in r.stdout, r.stdout[-2000:]


def test_genmap_program_reports_where_it_crashed():
    """The `genmap` program writes the faulting thread's stack to stderr before a fatal signal ends it (genmap_main.cpp:
    genmap_crash_handler), and still ends with that signal: a crash can be located, and nothing hides it (no retry in the CLI tests)."""
    import subprocess
    exe = Path(__file__).resolve().parent.parent / "genmap_amd" / "bin" / "genmap"
    assert exe.exists(), "genmap binary not built (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([str(exe), "selftest-crash"], capture_output=True, text=True)
    assert r.returncode == -11, r.returncode                                   # died from SIGSEGV itself
    assert "genmap: fatal signal 11" in r.stderr and "main" in r.stderr, r.stderr
    src = (Path(__file__).resolve().parent / "test_gpu_cli_end_to_end.py").read_text()
    assert "retried" not in src and "warnings.warn" not in src                 # the round-4 retry is gone

"""CPU check of the DEVICE engine's logic: tests/emu compiles the very headers the HIP kernel is built
from (gm_engine.h, gm_rank.h, gm_oss.h, gm_host.h) and runs them lane by lane on the host.  Compared
bit-exactly with the oracle.  This is not a product path (the product fails without a GPU); it exists so
that logic errors are found here and GPU minutes are spent on GPU questions."""
import ctypes as C
import subprocess

import numpy as np
import pytest

import helpers as H

_emu = None


def emu():
    global _emu
    if _emu is None:
        subprocess.check_call(["make", "-s", "-C", str(H.ROOT / "tests" / "emu")])
        _emu = C.CDLL(str(H.ROOT / "tests" / "emu" / "libgmemu.so"))
        _emu.gm_emu_map.restype = C.c_int
        _emu.gm_emu_map.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                    C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                    C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return _emu


def emu_map(ix, wpp, K, E, first_seq=0, n_seq=None, xo=None, infix=0, revcompl=True, value_bits=16, intervals=None, verify_t=0):
    if n_seq is None:
        n_seq = len(ix.seq_len) - first_seq
    tb = int(ix.cum[first_seq])
    tl = int(ix.cum[first_seq + n_seq]) - tb
    bf, br = ix.bwt(0), ix.bwt(1)
    text = np.ascontiguousarray(ix.codes[tb:tb + tl])
    cum = np.ascontiguousarray(ix.cum[first_seq:first_seq + n_seq + 1] - ix.cum[first_seq]).astype(np.uint64)
    out = np.zeros(tl, dtype=np.uint8 if value_bits == 8 else np.uint16)
    iv = None
    if intervals:
        iv = np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
    stats = np.zeros(4, dtype=np.uint64)
    sa = ix.sa() if verify_t else None
    allcodes = np.ascontiguousarray(ix.codes)
    allcum = np.ascontiguousarray(ix.cum, dtype=np.uint64)
    rc = emu().gm_emu_map(wpp, H._ptr(bf), H._ptr(br), ix.n, len(ix.seq_len), H._ptr(text), tl, H._ptr(cum), n_seq, K, E,
                          -1 if xo is None else xo, infix, int(revcompl), value_bits, H._ptr(iv),
                          0 if iv is None else len(iv) // 2, H._ptr(out), H._ptr(stats), H._ptr(sa), verify_t, H._ptr(allcodes), H._ptr(allcum))
    assert rc == 0, rc
    return out, stats


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_engine_logic_on_reference_fixtures(case):
    d = H.CASES_DIR / f"case_{case}"
    g, directory, fl, bed = H.load_case(case)
    if fl.get("ep"):
        pytest.skip("--exclude-pseudo goes through the locate path")
    ix = H.OracleIndex(g.codes, g.seq_len, keep_sa=False)
    for xo in H.xo_variants(case):
        for wpp in (1, 3, 9):
            for name, first, nseq, tb, tl in g.file_slices():
                iv = None
                if bed is not None:
                    iv = H.slice_intervals(g, first, nseq, bed)
                    if not iv:
                        continue
                out, _ = emu_map(ix, wpp, fl["K"], fl["E"], first, nseq, xo=xo, revcompl=not fl.get("nc", False), intervals=iv)
                exp = np.fromfile(d / "raw_freq16" / (name.rsplit(".", 1)[0] + ".genmap.freq16"), dtype=np.uint16)
                assert np.array_equal(out, exp), (case, xo, wpp, name, out.tolist(), exp.tolist())


@pytest.mark.parametrize("dna5", [False, True])
@pytest.mark.parametrize("E", [0, 1, 2, 3, 4])
def test_engine_logic_gtest_matrix(E, dna5):
    rng = np.random.default_rng(2000 + 10 * E + dna5)
    nseq, ln = 3, (1000 if E < 3 else 300)
    codes = rng.integers(0, 5 if dna5 else 4, size=nseq * ln, dtype=np.uint8)
    ix = H.OracleIndex(codes, [ln] * nseq, keep_sa=False)
    minK = E + 1 + (E >= 2)
    nblocks = [1, 2, 4, 5, 6][E]
    for K in range(minK, 9 if E < 4 else 8):
        rc = bool(rng.integers(0, 2))
        triv = ix.trivial(K, E, revcompl=rc, value_bits=8)
        for infix in range(max(minK, nblocks), K + 1):
            out, st = emu_map(ix, 3, K, E, infix=infix, revcompl=rc, value_bits=8)
            assert np.array_equal(out, triv), (E, dna5, K, infix)
            assert st[0] <= st[1], ("stack bound violated", st)


@pytest.mark.parametrize("K,E", [(30, 0), (30, 1), (30, 2), (100, 1), (24, 1), (50, 3), (36, 4), (128, 0), (150, 2), (250, 1), (255, 0)])
def test_engine_logic_baseline_settings(K, E):
    rng = np.random.default_rng(K * 10 + E)
    lens = [1500, 700, K - 1, 900, 3]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rng.integers(0, 4, size=200, dtype=np.uint8)
    for s in (50, 400, 1600, 2300, 2900):
        cp = fam.copy()
        mut = rng.random(200) < 0.03
        cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        codes[s:s + 200] = cp
    codes[700:760] = 4
    codes[1234] = 4
    codes[1000:1100] = 0  # poly-A
    ix = H.OracleIndex(codes, lens, keep_sa=False)
    for bits in (8, 16):
        exp = ix.mappability(K, E, value_bits=bits, threads=4)
        for wpp in (1, 3, 9):
            out, st = emu_map(ix, wpp, K, E, value_bits=bits)
            assert np.array_equal(out, exp), (K, E, bits, wpp)
            assert st[0] <= st[1], ("stack bound violated", st)


@pytest.mark.parametrize("T", [1, 3, 1000000])
@pytest.mark.parametrize("case", sorted(H.CASES))
def test_verification_shortcut_on_reference_fixtures(case, T):
    d = H.CASES_DIR / f"case_{case}"
    g, directory, fl, bed = H.load_case(case)
    if fl.get("ep"):
        pytest.skip("--exclude-pseudo goes through the locate path")
    ix = H.OracleIndex(g.codes, g.seq_len, keep_sa=True)
    for xo in H.xo_variants(case):
        for name, first, nseq, tb, tl in g.file_slices():
            iv = None
            if bed is not None:
                iv = H.slice_intervals(g, first, nseq, bed)
                if not iv:
                    continue
            out, st = emu_map(ix, 1, fl["K"], fl["E"], first, nseq, xo=xo, revcompl=not fl.get("nc", False), intervals=iv, verify_t=T)
            exp = np.fromfile(d / "raw_freq16" / (name.rsplit(".", 1)[0] + ".genmap.freq16"), dtype=np.uint16)
            assert np.array_equal(out, exp), (case, xo, T, name, out.tolist(), exp.tolist())


@pytest.mark.parametrize("dna5", [False, True])
@pytest.mark.parametrize("E", [0, 1, 2, 3, 4])
def test_verification_shortcut_gtest_matrix(E, dna5):
    rng = np.random.default_rng(4000 + 10 * E + dna5)
    nseq, ln = 3, (400 if E < 3 else 250)
    codes = rng.integers(0, 5 if dna5 else 4, size=nseq * ln, dtype=np.uint8)
    ix = H.OracleIndex(codes, [ln] * nseq, keep_sa=True)
    minK = E + 1 + (E >= 2)
    nblocks = [1, 2, 4, 5, 6][E]
    for K in range(minK, 9 if E < 4 else 8):
        rc = bool(rng.integers(0, 2))
        triv = ix.trivial(K, E, revcompl=rc, value_bits=8)
        for infix in range(max(minK, nblocks), K + 1):
            for T in (1, 2, 5, 1 << 30):
                out, st = emu_map(ix, 1, K, E, infix=infix, revcompl=rc, value_bits=8, verify_t=T)
                assert np.array_equal(out, triv), (E, dna5, K, infix, T)


@pytest.mark.parametrize("K,E", [(30, 0), (30, 1), (30, 2), (100, 1), (24, 1), (50, 3), (36, 4), (128, 0), (200, 1), (255, 2)])
def test_verification_shortcut_baseline_settings(K, E):
    rng = np.random.default_rng(K * 10 + E + 7)
    lens = [1500, 700, K - 1, 900, 3, K, K + 1]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rng.integers(0, 4, size=200, dtype=np.uint8)
    for s in (50, 400, 1600, 2300, 2900):
        cp = fam.copy()
        mut = rng.random(200) < 0.03
        cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        codes[s:s + 200] = cp
    codes[700:760] = 4
    codes[1234] = 4
    codes[1000:1100] = 0
    codes[n - 40:n - 10] = codes[10:40]   # a repeat touching the very end of the text
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    exp = ix.mappability(K, E, value_bits=16, threads=4)
    for T in (1, 4, 1 << 30):
        out, st = emu_map(ix, 1, K, E, value_bits=16, verify_t=T)
        assert np.array_equal(out, exp), (K, E, T)
        assert st[3] > 0


# ---- jump patterns + N-less main pass + correction pass (gm_oss.h: oss_jump_patterns, gm_engine.h: Env::NLESS) -------------------
def emu_map2(ix, wpp, K, E, first_seq=0, n_seq=None, xo=None, infix=0, revcompl=True, value_bits=16, intervals=None, verify_t=0, jump=15, nless=True):
    e = emu()
    e.gm_emu_map2.restype = C.c_int
    e.gm_emu_map2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                              C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_int,
                              C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    if n_seq is None:
        n_seq = len(ix.seq_len) - first_seq
    tb = int(ix.cum[first_seq])
    tl = int(ix.cum[first_seq + n_seq]) - tb
    bf, br = ix.bwt(0), ix.bwt(1)
    allcodes = np.ascontiguousarray(ix.codes, dtype=np.uint8)
    allcum = np.ascontiguousarray(ix.cum, dtype=np.uint64)
    text = allcodes[tb:tb + tl]                                     # a view: the slice's offset inside the whole text is its address
    cum = np.ascontiguousarray(ix.cum[first_seq:first_seq + n_seq + 1] - ix.cum[first_seq]).astype(np.uint64)
    out = np.zeros(tl, dtype=np.uint8 if value_bits == 8 else np.uint16)
    iv = np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1)) if intervals else None
    stats = np.zeros(7, dtype=np.uint64)
    sa = ix.sa()
    rc = e.gm_emu_map2(wpp, H._ptr(bf), H._ptr(br), ix.n, len(ix.seq_len), C.c_void_p(allcodes.ctypes.data + tb), tl, H._ptr(cum), n_seq, K, E,
                       -1 if xo is None else xo, infix, int(revcompl), value_bits, H._ptr(iv), 0 if iv is None else len(iv) // 2,
                       H._ptr(out), H._ptr(stats), H._ptr(sa), verify_t, H._ptr(allcodes), H._ptr(allcum), jump, int(nless))
    assert rc == 0, rc
    return out, stats


@pytest.fixture(params=[0, 1, 2], ids=["one_loop", "split", "split_two_passes"])
def expand_mode(request):
    """0: a root's jump patterns are consumed by the lane that walks their subtrees (the kernels of rounds 3-5); 1: phase A / phase B of
    round 6 (gm_expand.h): one work item per (root, item), node packets in three lists, every packet walked from its own needle window,
    the neighbour filters of one- and two-row table entries on; 2: the same in two passes -- the patterns without a substitution of every
    root first (a work item per root), then everything else (gm_expand.h: expand_strip_exact)"""
    e = emu()
    e.gm_emu_set_expand(request.param)
    yield request.param
    e.gm_emu_set_expand(0)


@pytest.fixture(params=[0, 2], ids=["one_loop", "split_two_passes"])
def expand_mode2(request):
    """the one-loop kernel's order and the split search in two passes (what the library runs by default): the long matrices; `expand_mode` adds the one-pass split"""
    e = emu()
    e.gm_emu_set_expand(request.param)
    yield request.param
    e.gm_emu_set_expand(0)


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_jump_patterns_and_n_correction_on_reference_fixtures(case, expand_mode2):
    d = H.CASES_DIR / f"case_{case}"
    g, directory, fl, bed = H.load_case(case)
    if fl.get("ep"):
        pytest.skip("--exclude-pseudo goes through the locate path")
    ix = H.OracleIndex(g.codes, g.seq_len, keep_sa=True)
    for xo in H.xo_variants(case):
        for name, first, nseq, tb, tl in g.file_slices():
            iv = None
            if bed is not None:
                iv = H.slice_intervals(g, first, nseq, bed)
                if not iv:
                    continue
            exp = np.fromfile(d / "raw_freq16" / (name.rsplit(".", 1)[0] + ".genmap.freq16"), dtype=np.uint16)
            for T, jump in ((0, 15), (1, 15), (1, 0), (4, 2)):
                out, st = emu_map2(ix, 1, fl["K"], fl["E"], first, nseq, xo=xo, revcompl=not fl.get("nc", False), intervals=iv, verify_t=T, jump=jump)
                assert np.array_equal(out, exp), (case, xo, T, jump, name, out.tolist(), exp.tolist())


@pytest.mark.parametrize("dna5", [False, True])
@pytest.mark.parametrize("E", [1, 2, 3, 4])
def test_jump_patterns_and_n_correction_gtest_matrix(E, dna5, expand_mode2):
    """random Dna5 text is 20 % N: nearly every window goes through the correction pass; Dna4: the patterns alone"""
    rng = np.random.default_rng(6000 + 10 * E + dna5)
    nseq, ln = 3, (400 if E < 3 else 250)
    codes = rng.integers(0, 5 if dna5 else 4, size=nseq * ln, dtype=np.uint8)
    ix = H.OracleIndex(codes, [ln] * nseq, keep_sa=True)
    minK = E + 1 + (E >= 2)
    nblocks = [1, 2, 4, 5, 6][E]
    used = 0
    for K in range(minK, 9 if E < 4 else 8):
        rc = bool(rng.integers(0, 2))
        triv = ix.trivial(K, E, revcompl=rc, value_bits=8)
        for infix in range(max(minK, nblocks), K + 1):
            for T, jump in ((0, 15), (2, 3), (1 << 30, 15)):
                out, st = emu_map2(ix, 1, K, E, infix=infix, revcompl=rc, value_bits=8, verify_t=T, jump=jump)
                assert np.array_equal(out, triv), (E, dna5, K, infix, T, jump)
                used += int(st[4])
    assert used > 0


@pytest.mark.parametrize("K,E", [(30, 1), (30, 2), (100, 1), (24, 1), (50, 3), (36, 4), (150, 2), (250, 1)])
def test_jump_patterns_and_n_correction_baseline_settings(K, E, expand_mode):
    rng = np.random.default_rng(K * 10 + E + 11)
    lens = [1500, 700, K - 1, 900, 3, K, K + 1]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rng.integers(0, 4, size=200, dtype=np.uint8)
    for s in (50, 400, 1600, 2300, 2900):
        cp = fam.copy()
        mut = rng.random(200) < 0.03
        cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        codes[s:s + 200] = cp
    codes[700:760] = 4          # a long run
    codes[1234] = 4             # isolated letters, one inside a repeat copy
    codes[1650] = 4
    codes[2310:2312] = 4        # a run of two
    codes[1499] = 4; codes[1500:1502] = 4   # a run across a sequence boundary
    codes[1000:1100] = 0
    codes[n - 40:n - 10] = codes[10:40]
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    exp = ix.mappability(K, E, value_bits=16, threads=4)
    for T, jump in ((0, 15), (1, 15), (4, 7), (1 << 30, 15), (1, 16)):   # 16: the jump length of indexes beyond 2^30 rows (offsets up to 15)
        out, st = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump)
        assert np.array_equal(out, exp), (K, E, T, jump, np.flatnonzero(out != exp)[:10], out[out != exp][:10], exp[out != exp][:10])
        assert st[4] > 0 and st[5] > 0
    # the first FASTA "file" alone (a slice): occurrences outside the slice do not count for it, needles from everywhere do
    exp2 = ix.mappability(K, E, first_seq=0, n_seq=2, value_bits=16, threads=4) if hasattr(ix, "mappability") else None
    out2, _ = emu_map2(ix, 1, K, E, first_seq=0, n_seq=2, value_bits=16, verify_t=1, jump=15)
    assert np.array_equal(out2, exp2), (K, E, "slice")


@pytest.mark.parametrize("K,E", [(30, 1), (30, 2), (24, 1), (24, 2), (50, 3), (36, 4), (100, 1)])
def test_groups_of_jump_patterns_read_through_the_existence_bitmap(K, E, expand_mode):
    """gm_oss.h: patterns that differ in their last three characters only form a group answered by one word of the bitmap "which J-mers
    occur" (word_to_rotations, rotations_to_low6, rotations_errors, oss_group_patterns): grouped == plain patterns == oracle, for the jump
    lengths the device uses (15, 16) and short ones, on a text with repeats and N"""
    if expand_mode == 1 and E >= 3:
        pytest.skip("the long settings run the one-loop order and the split in two passes (the library's default); the one-pass split runs the others")
    rng = np.random.default_rng(K * 10 + E + 77)
    lens = [1800, 600, K + 2, 1000]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rng.integers(0, 4, size=150, dtype=np.uint8)
    for s in (30, 500, 1900, 2500, 3100):
        cp = fam.copy()
        mut = rng.random(150) < 0.04
        cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        codes[s:s + 150] = cp
    codes[800:830] = 4
    codes[2000] = 4
    codes[1200:1290] = 1
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    exp = ix.mappability(K, E, value_bits=16, threads=4)
    e = emu()
    try:
        for T, jump in (((0, 16), (1, 15), (4, 9), (1, 5)) if E < 4 else ((0, 16), (4, 9))):   # (four errors: a minute per mode with all four -- the jump lengths of the device and a short one)
            e.gm_emu_set_jump_groups(0)
            plain, st0 = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump)
            e.gm_emu_set_jump_groups(1)
            out, st1 = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump)
            assert np.array_equal(plain, exp) and np.array_equal(out, exp), (K, E, T, jump, np.flatnonzero(out != exp)[:10])
            assert 0 < st1[4] <= st0[4]          # only J-mers that occur are looked up
            e.gm_emu_set_jump_groups(2)          # round 5: groups at any three adjacent characters (the searches that start on the right)
            out2, st2 = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump)
            assert np.array_equal(out2, exp), (K, E, T, jump, "all layouts", np.flatnonzero(out2 != exp)[:10])
            assert 0 < st2[4] <= st1[4]
            # every root above went through the lane's pattern-fetch state machine (gm_oss.h: jump_decide, the code of the kernel's part B)
            # with an iteration bound: none may hang, none may ask for a word at another address than the kernel would; and a plain loop
            # over the items looks up exactly the same patterns
            e.gm_emu_hangs.restype = C.c_uint64
            assert e.gm_emu_hangs(1) == 0
            e.gm_emu_set_state_machine(0)
            out3, st3 = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump)
            e.gm_emu_set_state_machine(1)
            assert np.array_equal(out3, exp) and st3[4] == st2[4], (K, E, T, jump, st3[4], st2[4])
    finally:
        e.gm_emu_set_jump_groups(0)
        e.gm_emu_set_state_machine(1)


@pytest.mark.parametrize("K,E", [(30, 1), (30, 2), (24, 2), (36, 3), (100, 1)])
def test_phase_a_node_packets_and_neighbour_filters(K, E):
    """gm_expand.h end to end on the CPU: expand_root (J-mer index, neighbours and the two letters behind the J-mer taken from the 4-bit text at
    any alignment, both strands), expand_item / expand_word / expand_next (plain patterns and groups of every layout), expand_filter (one- and
    two-row entries, rows-only nodes), the three lists -- equal to the oracle with the filters on, one-row only,
    and off; the filters must end nodes and every list must be used."""
    e = emu()
    e.gm_emu_packets.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(K * 7 + E)
    lens = [2500, 37, K + 5, 1900, K - 1, 600]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rng.integers(0, 4, size=260, dtype=np.uint8)
    for s0 in (100, 700, 2700, 3300, 4200, 4600):
        cp = fam.copy()
        mut = rng.random(260) < 0.03
        cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        codes[s0:s0 + 260] = cp
    codes[1500:1560] = 4
    for p0 in (150, 2801, 2802, 4300):
        codes[p0] = 4
    codes[1700:1800] = 3
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    exp = ix.mappability(K, E, value_bits=16, threads=4)
    pk = np.zeros(4, dtype=np.uint64)
    ended = {}
    try:
        for groups, xmode in ((0, 1), (2, 1), (2, 2), (0, 2)):
            e.gm_emu_set_expand(xmode)
            e.gm_emu_set_jump_groups(groups)
            for nbf in ((1, 2, 0) if xmode == 1 else (1,)):
                e.gm_emu_set_nb_filter(nbf)
                for T, jump in ((1, 16), (4, 9), (0, 12)):
                    e.gm_emu_packets(H._ptr(pk), 1)
                    out, st = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump)
                    assert np.array_equal(out, exp), (K, E, groups, nbf, T, jump, np.flatnonzero(out != exp)[:10])
                    e.gm_emu_packets(H._ptr(pk), 1)
                    assert pk[0] > 0 and (jump < 16 or K >= 64 or (pk[1] > 0 and (E < 2 or pk[2] > 0))), pk
                    ended[(groups, nbf, T, jump, xmode)] = int(pk[3])
                    assert st[6] > 0                       # self hits (the walker's: a redone chunk of phase A must not add twice)
        assert ended[(0, 1, 1, 16, 1)] > 0 and ended[(0, 0, 1, 16, 1)] == 0 and ended[(0, 1, 1, 16, 1)] >= ended[(0, 2, 1, 16, 1)] > 0 and ended[(0, 1, 1, 16, 2)] == ended[(0, 1, 1, 16, 1)], ended
        # slices and a selection go through the same code
        e.gm_emu_set_nb_filter(1)
        out2, _ = emu_map2(ix, 1, K, E, first_seq=0, n_seq=2, value_bits=16, verify_t=1, jump=15)
        assert np.array_equal(out2, ix.mappability(K, E, first_seq=0, n_seq=2, value_bits=16, threads=4))
        iv = [(40, 900), (2400, 2600), (4100, 4500)]
        out3, _ = emu_map2(ix, 1, K, E, value_bits=16, verify_t=1, jump=15, intervals=iv)
        assert np.array_equal(out3, ix.mappability(K, E, value_bits=16, intervals=iv, threads=4))
        e.gm_emu_hangs.restype = C.c_uint64
        assert e.gm_emu_hangs(1) == 0      # expand_nth (the device's way to a rotation) agreed with the item's own iterator everywhere, every word address was right
    finally:
        e.gm_emu_set_expand(0); e.gm_emu_set_jump_groups(0); e.gm_emu_set_nb_filter(1)


def test_n_window_intervals_list_exactly_the_windows_that_can_match():
    """gm_host.h: n_window_intervals against brute force on random texts with runs of N of every kind (isolated letters, short and
    long runs, runs across sequence boundaries, at both ends of the text): every window with 1..E letters N inside one sequence is
    listed, no window without N and no window crossing a sequence boundary ever is, intervals are sorted and disjoint"""
    e = emu()
    e.gm_emu_n_window_intervals.restype = C.c_uint64
    e.gm_emu_n_window_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(9)
    for trial in range(300):
        lens = [int(x) for x in rng.integers(1, 80, size=int(rng.integers(1, 6)))]
        n = sum(lens)
        codes = rng.integers(0, 4, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 7))):
            s = int(rng.integers(0, n)); codes[s:s + int(rng.integers(1, 12))] = 4
        if trial % 7 == 0:
            codes[:3] = 4; codes[-2:] = 4
        K, E = int(rng.integers(1, 14)), int(rng.integers(1, 5))
        cum = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        out = np.zeros(2 * (n + 8), np.uint64)
        cnt = int(e.gm_emu_n_window_intervals(H._ptr(codes), H._ptr(cum), len(lens), K, E, H._ptr(out), n + 8))
        listed = np.zeros(n + 1, bool)
        prev = 0
        for k in range(cnt):
            b, en = int(out[2 * k]), int(out[2 * k + 1])
            assert prev <= b < en <= n, (trial, k, b, en)
            prev = en
            listed[b:en] = True
        isn = (codes == 4).astype(np.int64)
        pre = np.concatenate([[0], np.cumsum(isn)])
        for t in range(n):
            sq = int(np.searchsorted(cum, t, side="right") - 1)
            inside = t + K <= int(cum[sq + 1])
            c = int(pre[min(n, t + K)] - pre[t])
            if listed[t]:
                assert inside and c >= 1, (trial, t, K, E, lens)
            elif inside and 1 <= c <= E:
                assert False, ("window with 1..E letters N not listed", trial, t, K, E, lens, codes[t:t + K].tolist())


@pytest.mark.parametrize("E,weights", [(1, 0x31), (2, 0x7755), (2, 0x1F31), (3, 0x12345), (4, 0x654321)])
def test_uneven_oss_block_lengths_give_the_same_counts(E, weights):
    """the scheme covers every error distribution exactly once for ANY positive block lengths (l/u bound errors per block, not
    per position): the library's e = 2 default (blocks of 5,5,7,7 for an infix of 24) and lopsided splits, with and without jumps"""
    rng = np.random.default_rng(8000 + E)
    lens = [700, 31, 500]
    codes = rng.integers(0, 4, size=sum(lens), dtype=np.uint8)
    codes[100:160] = codes[900:960]; codes[300] = 4; codes[650:653] = 4
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    K = 30 if E <= 2 else 36
    exp = ix.mappability(K, E, value_bits=16, threads=4)
    emu().gm_emu_set_oss_weights(weights)
    try:
        for T, jump, nless in ((0, 0, False), (1, 15, True), (4, 6, True)):
            out, _ = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=jump, nless=nless)
            assert np.array_equal(out, exp), (E, hex(weights), T, jump)
    finally:
        emu().gm_emu_set_oss_weights(0)


@pytest.mark.parametrize("E", [1, 2, 3, 4])
def test_items_of_a_search_expand_to_its_patterns(E):
    """gm_oss.h without an index: rotation words, groups in both layouts and both kinds against the pattern descriptors, for the block shapes
    the library schedules and the jump lengths the device uses (and short ones); the 3 Gbp case K=30 e=2 J=16 must form its 14 + 4 groups"""
    e = emu()
    e.gm_emu_check_items.restype = C.c_int
    e.gm_emu_check_items.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
    st = np.zeros(3, dtype=np.uint64)
    for K in (24, 30, 36, 50, 100, 150):
        for J in (16, 15, 12, 9, 6, 5, 4):
            rc = e.gm_emu_check_items(K, E, J, 0x8845 if E == 2 else 0, 3, K * 100 + J, H._ptr(st))
            assert rc in (0, -1, -2), (K, E, J, rc)
            if (K, E, J) == (30, 2, 16):
                assert rc == 0 and tuple(int(v) for v in st) == (261, 80, 18), st
    # round 5: every layout may hold kind-0 groups -- the one-substitution patterns of the searches that start on the right of the infix
    e.gm_emu_check_items2.restype = C.c_int
    e.gm_emu_check_items2.argtypes = [C.c_uint32] * 6 + [C.c_void_p, C.c_uint32]
    for K in (24, 30, 36, 50, 100, 150):
        for J in (16, 15, 12, 9, 6, 5, 4):
            rc = e.gm_emu_check_items2(K, E, J, 0x8845 if E == 2 else 0, 3, K * 100 + J + 1, H._ptr(st), 0xFF)
            assert rc in (0, -1, -2), (K, E, J, rc)
            if (K, E, J) == (30, 2, 16):
                assert rc == 0 and int(st[0]) == 261 and int(st[1]) < 80 and int(st[2]) > 18, st
                print("K=30 e=2 J=16 with every layout: patterns, items, groups =", [int(v) for v in st])


# ---- fast verification: the whole window compared in one go (gm_engine.h: fv_masks, scan_side over masks) ---------------------------
def test_fast_verification_masks_match_their_definition():
    """fv_masks (nibble funnels, reversal + complement of the reverse-strand needle, record alignment, flag compression) against a
    symbol-by-symbol loop, and the mask form of scan_side against a plain scan: 200 000 random windows, every alignment and anchor."""
    e = emu()
    e.gm_emu_check_fv_masks.restype = C.c_int
    e.gm_emu_check_fv_masks.argtypes = [C.c_uint32, C.c_uint32]
    for seed in (1, 2, 3, 4):
        assert e.gm_emu_check_fv_masks(50000, seed) == 0, seed


@pytest.mark.parametrize("K,E,T", [(30, 2, 1), (30, 1, 4), (30, 2, 16), (24, 1, 2), (32, 1, 3), (20, 3, 2), (12, 1, 1), (30, 0, 1)])
def test_fast_verification_gives_the_same_counts(K, E, T):
    """Narrow nodes settled from the masks (K <= 32: one round of reads per item on the device) against the oracle and against the
    scanning verification, on a text with repeat families, N runs and short sequences; plain walk and jump patterns + N-less pass."""
    e = emu()
    e.gm_emu_fast_items.restype = C.c_uint64
    rng = np.random.default_rng(K * 100 + E * 10 + T)
    lens = [1500, 700, K - 1, 900, 3, K, K + 1, 33]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rng.integers(0, 4, size=200, dtype=np.uint8)
    for s in (50, 400, 1600, 2300, 2900):
        cp = fam.copy()
        mut = rng.random(200) < 0.04
        cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        codes[s:s + 200] = cp
    codes[700:760] = 4
    for p in (1234, 1610, 2333, 2334, 60):
        codes[p] = 4
    codes[1000:1100] = 0
    codes[n - 40:n - 10] = codes[10:40]
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    exp = ix.mappability(K, E, value_bits=16, threads=4)
    try:
        for fast in (1, 0):
            e.gm_emu_set_fast_verify(fast)
            e.gm_emu_fast_items(1)
            out, st = emu_map(ix, 1, K, E, value_bits=16, verify_t=T)
            assert np.array_equal(out, exp), (K, E, T, fast, "plain walk")
            if E >= 1:
                out, st = emu_map2(ix, 1, K, E, value_bits=16, verify_t=T, jump=9)
                assert np.array_equal(out, exp), (K, E, T, fast, "jumps + N-less + correction")
            assert (e.gm_emu_fast_items(0) > 0) == bool(fast), (K, E, T, fast)
    finally:
        e.gm_emu_set_fast_verify(1)


# ---- k-mers longer than 255: the per-node code of the long k-mer kernel (genmap_amd/csrc/gm_longk_step.h) ---------------------------------
@pytest.mark.parametrize("K,E", [(256, 0), (300, 1), (300, 2), (513, 3), (1000, 1)])
def test_long_kmer_node_logic(K, E):
    """gm_longk_step.h (long_root_node, long_node: split, plan, children in the kernel's order, verify_fields with OssRecordL) through a plain
    LIFO on the host: plain walk down to the leaves, single rows / up to four rows settled against the text; both counter widths, one strand,
    a selection, an explicit infix -- all equal to the oracle; the lane stack stays within stack_bound."""
    rng = np.random.default_rng(K * 10 + E + 5)
    lens = [5000, K + 40, K - 1, 2600, 3]
    n = sum(lens)
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = codes[100:1500].copy()                       # long exact and near-exact copies: long k-mers with several occurrences
    for i, s0 in enumerate((2600, n - 2550)):
        cp = fam.copy()
        if i: cp[rng.integers(0, 1400, size=3)] ^= 1
        codes[s0:s0 + 1400] = cp
    codes[4000:4030] = 4
    codes[700] = 4
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    for bits in (8, 16):
        exp = ix.mappability(K, E, value_bits=bits, threads=4)
        for T in (0, 1, 4):
            out, st = emu_map(ix, 1, K, E, value_bits=bits, verify_t=T)
            assert np.array_equal(out, exp), (K, E, bits, T)
            assert st[0] <= st[1] and (T == 0 or st[3] > 0), (K, E, T, st.tolist())
    assert exp.max() >= 2
    out, _ = emu_map(ix, 1, K, E, value_bits=16, verify_t=1, infix=K - 5)
    assert np.array_equal(out, exp)
    out, _ = emu_map(ix, 1, K, E, value_bits=16, verify_t=1, revcompl=False)
    assert np.array_equal(out, ix.mappability(K, E, value_bits=16, revcompl=False, threads=4))
    iv = [(50, 900), (2500, 3100), (4900, 5000)]
    out, _ = emu_map(ix, 1, K, E, value_bits=16, verify_t=1, intervals=iv)
    assert np.array_equal(out, ix.mappability(K, E, value_bits=16, intervals=iv, threads=4))

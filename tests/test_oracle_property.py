"""Seeded restatement of the reference's randomized gtest (/root/reference/tests/tests.cpp:133-260):
computeMappability<E> == trivial backtracking for E=0..4, Dna4 and Dna5, 3 x 1000 bp, every infix
length, with a portable PRNG -- plus an index-free brute-force check of the definition itself."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("dna5", [False, True])
@pytest.mark.parametrize("E", [0, 1, 2, 3, 4])
def test_genmap_equals_trivial_backtracking(E, dna5):
    rng = np.random.default_rng(1000 + 10 * E + dna5)
    nseq, ln = 3, (1000 if E < 4 else 400)
    hi = 5 if dna5 else 4
    codes = rng.integers(0, hi, size=nseq * ln, dtype=np.uint8)  # uniform over the alphabet (tests.cpp:21-28)
    ix = H.OracleIndex(codes, [ln] * nseq, keep_sa=True)
    minK = E + 1 + (E >= 2)
    maxK = 8 if E < 4 else 7
    for K in range(minK, maxK + 1):
        rc = bool(rng.integers(0, 2))
        triv = ix.trivial(K, E, revcompl=rc, value_bits=8)
        for infix in range(minK, K + 1):
            for shortcut in (False, True):
                got = ix.mappability(K, E, revcompl=rc, value_bits=8, infix=infix, use_shortcut=shortcut, threads=2)
                assert np.array_equal(got, triv), (E, dna5, K, infix, shortcut)


@pytest.mark.parametrize("E", [0, 1, 2, 3])
def test_trivial_backtracking_equals_index_free_definition(E):
    rng = np.random.default_rng(77 + E)
    lens = [60, 5, 90, 31]
    codes = rng.integers(0, 5, size=sum(lens), dtype=np.uint8)
    codes[20:29] = 4  # an N run
    # plant repeats so counts exceed 1
    codes[100:130] = codes[0:30]
    ix = H.OracleIndex(codes, lens, keep_sa=False)
    for K in (E + 2, 6, 9):
        for rc in (False, True):
            triv = ix.trivial(K, E, revcompl=rc, value_bits=16)
            brute = H.brute_force(codes, lens, K, E, revcompl=rc, value_bits=16)
            assert np.array_equal(triv, brute), (E, K, rc)


@pytest.mark.parametrize("K,E", [(30, 0), (30, 1), (30, 2), (100, 1), (24, 1), (50, 3), (36, 4)])
def test_baseline_settings_against_definition(K, E):
    """The BASELINE (K,E) settings with the reference's default infix lengths on a small repeat-bearing
    Dna5 text: oracle == index-free definition, 8- and 16-bit."""
    rng = np.random.default_rng(K * 10 + E)
    lens = [1500, 700, K - 1, 900]
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
    ix = H.OracleIndex(codes, lens, keep_sa=True)
    for bits in (8, 16):
        brute = H.brute_force(codes, lens, K, E, revcompl=True, value_bits=bits)
        for shortcut in (False, True):
            got = ix.mappability(K, E, revcompl=True, value_bits=bits, use_shortcut=shortcut, threads=4)
            assert np.array_equal(got, brute), (K, E, bits, shortcut)


def test_default_infix_lengths():
    # SURVEY §8a [probe]: k30e0 -> 9, k30e1 -> 24, k30e2 -> 26, k100e1 -> 31, k24e1 -> 19 (src/mappability.hpp:519-543)
    assert H.default_infix_length(30, 0) == 9
    assert H.default_infix_length(30, 1) == 24
    assert H.default_infix_length(30, 2) == 26
    assert H.default_infix_length(100, 1) == 31
    assert H.default_infix_length(24, 1) == 19
    assert H.default_infix_length(3, 1) == 3
    assert H.default_infix_length(4, 0, 1) == 3
    assert H.default_infix_length(3, 1, 1) == -1  # "overlap cannot be larger than min(K - 1, K - E - 2)"

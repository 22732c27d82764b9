"""Pins the CPU oracle against the reference's own 18 end-to-end fixture cases
(/root/reference/tests/test_cases, copied as data to tests/golden/reference_cases) for every -xo
rerun of /root/reference/tests/tests.sh:47-60, with and without the duplicate-copy shortcut."""
import numpy as np
import pytest

import helpers as H


def _run_case(case, xo, shortcut, value_bits, csv=False):
    g, directory, fl, bed = H.load_case(case)
    ix = H.OracleIndex(g.codes, g.seq_len, keep_sa=True)
    results = {}
    for name, first, nseq, tb, tl in g.file_slices():
        iv = None
        if bed is not None:
            iv = H.slice_intervals(g, first, nseq, bed)
            if not iv:  # src/mappability.hpp:308-314: no interval in this fasta file -> no output at all
                continue
        out, csk, locs = ix.mappability(fl["K"], fl["E"], text_begin=tb, text_len=tl, first_seq=first, n_seq=nseq,
                                        overlap=xo, revcompl=not fl.get("nc", False), value_bits=value_bits,
                                        directory=directory, exclude_pseudo=fl.get("ep", False), csv=csv,
                                        use_shortcut=shortcut, intervals=iv, seq_file_id=g.seq_file, want_flag=True)
        if iv and csk:  # selection reset of src/mappability.hpp:83-99
            keep = np.zeros(len(out), bool)
            for b, e in iv:
                keep[b:e] = True
            out[~keep] = 0
        results[name] = (out, locs, iv, first, nseq)
    return g, results


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_oracle_freq_matches_reference_fixtures(case):
    d = H.CASES_DIR / f"case_{case}"
    for xo in H.xo_variants(case):
        for shortcut in (False, True):
            for bits, sub, ext in ((16, "raw_freq16", "freq16"), (8, "raw_freq8", "freq8")):
                g, res = _run_case(case, xo, shortcut, bits)
                expected_files = sorted(p.name for p in (d / sub).iterdir())
                got_files = sorted(n.rsplit(".", 1)[0] + ".genmap." + ext for n in res)
                assert got_files == expected_files, (case, xo, shortcut)
                for name, (out, _, _, _, _) in res.items():
                    exp = np.fromfile(d / sub / (name.rsplit(".", 1)[0] + ".genmap." + ext),
                                      dtype=np.uint16 if bits == 16 else np.uint8)
                    assert np.array_equal(out, exp), (case, xo, shortcut, bits, name, out.tolist(), exp.tolist())


@pytest.mark.parametrize("case", sorted(H.CASES))
def test_oracle_csv_matches_reference_fixtures(case):
    d = H.CASES_DIR / f"case_{case}"
    _, _, fl, bed = H.load_case(case)
    for xo in H.xo_variants(case):
        for shortcut in (False, True):
            g, res = _run_case(case, xo, shortcut, 16, csv=True)
            for name, (_, locs, iv, first, nseq) in res.items():
                civ = None
                if bed is not None:
                    civ = sorted((s - first, b, e) for s in range(first, first + nseq)
                                 for b, e in bed.get(g.seq_names[s], []))
                txt = H.format_csv(g, locs, not fl.get("nc", False), civ)
                exp = (d / "csv" / (name.rsplit(".", 1)[0] + ".genmap.csv")).read_text()
                assert txt == exp, (case, xo, shortcut, name)

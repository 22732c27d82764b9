"""Synthetic genomes for bench.py and the large parity tests (no network: no E. coli / GRCh38 FASTA here).

Deterministic and portable: every base is a pure function of (seed, position) through splitmix64, and the
repeat / N-block / tandem structure is drawn from the same hash, so the same call gives the same bytes on
any box (S1 "ecoli-like", S2 "chr1-like", S3 "grch38-like", S5 "5 bacteria").
Workload support only -- not part of the product path.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _hash_range(seed, stream, start, n):
    with np.errstate(over="ignore"):
        base = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)
        return splitmix64(np.arange(start, start + n, dtype=np.uint64) + base)


def _scalar_hash(seed, stream, i):
    return int(_hash_range(seed, stream, i, 1)[0])


def random_bases(seed, stream, start, n):
    return (_hash_range(seed, stream, start, n) >> np.uint64(62)).astype(np.uint8)


def _plant_family(text, seed, fam_id, unit_len, copies, div_lo, div_hi):
    n = len(text)
    if n <= unit_len + 1 or copies <= 0:
        return
    cons = random_bases(seed, 1000 + fam_id, 0, unit_len)
    pos = (_hash_range(seed, 2000 + fam_id, 0, copies) % np.uint64(n - unit_len)).astype(np.int64)
    divs = div_lo + (div_hi - div_lo) * ((_hash_range(seed, 3000 + fam_id, 0, copies) >> np.uint64(11)).astype(np.float64) / float(1 << 53))
    # process in batches to bound memory
    B = max(1, (1 << 24) // unit_len)
    for s in range(0, copies, B):
        p = pos[s:s + B]
        k = len(p)
        h = _hash_range(seed, 4000 + fam_id, s * unit_len, k * unit_len).reshape(k, unit_len)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        mut = u < divs[s:s + k, None]
        sub = ((h & np.uint64(3)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)  # 1..3: always a different base
        block = np.where(mut, (cons[None, :] + sub) & 3, cons[None, :]).astype(np.uint8)
        idx = p[:, None] + np.arange(unit_len, dtype=np.int64)[None, :]
        text[idx.reshape(-1)] = block.reshape(-1)


def _plant_tandem(text, seed, frac):
    n = len(text)
    target = int(n * frac)
    done, j = 0, 0
    while done < target and j < 100000:
        h = _scalar_hash(seed, 5000, j)
        motif_len = 1 + (h & 7)              # 1..8
        run = 200 + ((h >> 8) % 3000)
        run = min(run, n // 4)
        if run < motif_len + 1:
            break
        start = (h >> 24) % (n - run)
        motif = random_bases(seed, 6000 + j, 0, motif_len)
        text[start:start + run] = np.resize(motif, run)
        done += run
        j += 1


def make_sequence(length, seed, dna5=True, n_frac=0.07, short_copies_per_mbp=400, long_copies_per_mbp=20, tandem_frac=0.02):
    """One chromosome-like sequence of `length` codes (A0 C1 G2 T3 N4)."""
    text = np.empty(length, dtype=np.uint8)
    CH = 1 << 24
    for s in range(0, length, CH):
        m = min(CH, length - s)
        text[s:s + m] = random_bases(seed, 1, s, m)
    mbp = length / 1e6
    _plant_family(text, seed, 1, 300, int(short_copies_per_mbp * mbp), 0.10, 0.15)   # SINE-like
    _plant_family(text, seed, 2, 6000, int(long_copies_per_mbp * mbp), 0.03, 0.07)   # LINE-like
    _plant_family(text, seed, 3, 1300, int(8 * mbp), 0.0, 0.01)                       # recent segmental copies
    _plant_tandem(text, seed, tandem_frac)
    if dna5 and length >= 1000:
        end = max(1, min(10000, length // 200))
        text[:end] = 4
        text[length - end:] = 4
        big = int(length * n_frac) - 2 * end
        if big > 0:
            start = int(length * 0.49)
            text[start:start + big] = 4
        # a few isolated IUPAC-like positions
        k = max(1, int(mbp))
        iso = (_hash_range(seed, 7000, 0, k) % np.uint64(length)).astype(np.int64)
        text[iso] = 4
    return text


def _plant_shared_family(text, cons_seed, seed, fam_id, unit_len, copies, div_lo, div_hi):
    """like _plant_family, but the consensus comes from `cons_seed` (one family for the whole genome: copies on every sequence)
    and every other copy is planted reverse-complemented"""
    n = len(text)
    if n <= unit_len + 1 or copies <= 0:
        return
    cons = random_bases(cons_seed, 1000 + fam_id, 0, unit_len)
    pos = (_hash_range(seed, 2000 + fam_id, 0, copies) % np.uint64(n - unit_len)).astype(np.int64)
    hh = _hash_range(seed, 3000 + fam_id, 0, copies)
    divs = div_lo + (div_hi - div_lo) * ((hh >> np.uint64(11)).astype(np.float64) / float(1 << 53))
    flip = (hh & np.uint64(1)).astype(bool)
    B = max(1, (1 << 24) // unit_len)
    for s in range(0, copies, B):
        p = pos[s:s + B]
        k = len(p)
        h = _hash_range(seed, 4000 + fam_id, s * unit_len, k * unit_len).reshape(k, unit_len)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        mut = u < divs[s:s + k, None]
        sub = ((h & np.uint64(3)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)
        block = np.where(mut, (cons[None, :] + sub) & 3, cons[None, :]).astype(np.uint8)
        f = flip[s:s + k]
        block[f] = (3 - block[f])[:, ::-1]
        idx = p[:, None] + np.arange(unit_len, dtype=np.int64)[None, :]
        text[idx.reshape(-1)] = block.reshape(-1)


def make_sequence_hard(length, seed, satellite=0, dna5=True, n_frac=0.07):
    """A chromosome of the HARD variant of S3 ("grch38h"): what S3 lacks against a real genome.  The repeat families are shared by ALL
    sequences (one consensus per family for the whole text), half of their copies lie on the reverse strand, a young SINE-like subfamily
    (1-3 % from its consensus, ~3 % of the text) puts thousands of copies within two substitutions of one another at K = 30, and
    `satellite` > 0 plants an array of that many bases of a 171-bp monomer (1-2 % between monomers), as at a centromere."""
    text = np.empty(length, dtype=np.uint8)
    CH = 1 << 24
    for s in range(0, length, CH):
        m = min(CH, length - s)
        text[s:s + m] = random_bases(seed, 1, s, m)
    mbp = length / 1e6
    G = 424242   # the genome-wide consensus seed
    _plant_shared_family(text, G, seed, 1, 300, int(400 * mbp), 0.10, 0.15)        # old SINE-like, as S3 but genome-wide and on both strands
    _plant_shared_family(text, G, seed, 2, 6000, int(20 * mbp), 0.03, 0.07)        # LINE-like
    _plant_shared_family(text, G, seed, 4, 300, int(0.03 * length / 300), 0.01, 0.03)   # young SINE-like subfamily: ~3 % of the text
    _plant_family(text, seed, 3, 1300, int(8 * mbp), 0.0, 0.01)                    # recent segmental copies (per sequence)
    _plant_tandem(text, seed, 0.02)
    if satellite > 0 and length > 4 * satellite:
        mono = random_bases(G, 9001, 0, 171)
        start = int(length * 0.40)
        arr = np.resize(mono, satellite).copy()
        h = _hash_range(seed, 9002, 0, satellite)
        mut = ((h >> np.uint64(11)).astype(np.float64) / float(1 << 53)) < 0.015
        arr[mut] = (arr[mut] + ((h[mut] & np.uint64(3)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)) & 3
        text[start:start + satellite] = arr
    if dna5 and length >= 1000:
        end = max(1, min(10000, length // 200))
        text[:end] = 4
        text[length - end:] = 4
        big = int(length * n_frac) - 2 * end
        if big > 0:
            start = int(length * 0.49)
            text[start:start + big] = 4
        k = max(1, int(mbp))
        iso = (_hash_range(seed, 7000, 0, k) % np.uint64(length)).astype(np.int64)
        text[iso] = 4
    return text


# GRCh38 primary assembly chromosome lengths (chr1..22, X, Y)
GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                  133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
                  58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def workload(name, scale=1.0):
    """Returns (codes, seq_len list, description).  scale < 1 shrinks every sequence (tests)."""
    if name == "ecoli":      # S1 -> BASELINE configs[0]
        ln = max(1000, int(4641652 * scale))
        t = make_sequence(ln, seed=1, dna5=False, short_copies_per_mbp=4, long_copies_per_mbp=10, tandem_frac=0.002)
        return t, [ln], f"S1 ecoli-like {ln} bp Dna4"
    if name == "chr1":       # S2 -> BASELINE configs[1]
        ln = max(1000, int(248956422 * scale))
        return make_sequence(ln, seed=2), [ln], f"S2 chr1-like {ln} bp Dna5"
    if name == "grch38":     # S3 -> BASELINE configs[2], configs[3]
        lens = [max(1000, int(x * scale)) for x in GRCH38_LENGTHS]
        parts = [make_sequence(ln, seed=300 + i) for i, ln in enumerate(lens)]
        return np.concatenate(parts), lens, f"S3 grch38-like {sum(lens)} bp in {len(lens)} sequences Dna5"
    if name == "grch38h":    # the hard variant of S3 (round 6): genome-wide families on both strands, a young subfamily, two satellite arrays
        lens = [max(1000, int(x * scale)) for x in GRCH38_LENGTHS]
        parts = [make_sequence_hard(ln, seed=300 + i, satellite=int(scale * (2_000_000 if i == 0 else 1_200_000 if i == 8 else 0))) for i, ln in enumerate(lens)]
        return np.concatenate(parts), lens, f"S3h grch38-like (genome-wide families on both strands, young subfamily, satellites) {sum(lens)} bp in {len(lens)} sequences Dna5"
    raise ValueError(name)


def bacteria5(scale=1.0):
    """S5: five related 'genomes' (files) of 1-3 sequences each, derived from one ancestor.  Returns
    [(file name, [(seq name, codes)])]."""
    anc_len = max(2000, int(4_000_000 * scale))
    anc = make_sequence(anc_len, seed=5, dna5=False, short_copies_per_mbp=6, long_copies_per_mbp=8, tandem_frac=0.003)
    files = []
    for g in range(5):
        div = 0.01 + 0.01 * g
        h = _hash_range(5, 8000 + g, 0, anc_len)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        sub = ((h & np.uint64(3)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)
        t = np.where(u < div, (anc + sub) & 3, anc).astype(np.uint8)
        island = random_bases(5, 9000 + g, 0, max(200, anc_len // 20))
        t = np.concatenate([t, island])
        if g % 2 == 1:
            t[len(t) // 3: len(t) // 3 + 50] = 4
        nseq = 1 + g % 3
        cuts = [0] + [len(t) * (i + 1) // nseq for i in range(nseq)]
        recs = [(f"g{g}_seq{i}", np.ascontiguousarray(t[cuts[i]:cuts[i + 1]])) for i in range(nseq)]
        files.append((f"genome{g}.fa", recs))
    return files

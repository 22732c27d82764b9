#!/usr/bin/env python3
"""64-bit rows with the suffix array resident: jump patterns, the N-less pass with its correction pass and verification against the text
on an index of more than 2^32 rows (round 5).  The text is TWO copies of a 12-sequence ~2.16 Gbp synthetic genome (tools/wide_rows_smoke.py):
every count is min(MAX, 2 x the count on the single copy), which the 32-bit path of the same library supplies on intervals.
Timed: K=30 e=0 on the whole text (table of all 16-mers, 16-byte entries), e=1 on the WHOLE text with and without jump patterns,
e=2 on 1 %."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import genmap_amd as g
from genmap_amd import synth

quick = "--quick" in sys.argv   # (the GPU suite: no jump = 0 whole-text pass, no e = 2 sample timing)
args = [a for a in sys.argv[1:] if a != "--quick"]
scale = float(args[0]) if args else 0.7
lens_half = [max(1000, int(x * scale)) for x in synth.GRCH38_LENGTHS[:24]]
t0 = time.time()
half = np.concatenate([synth.make_sequence(ln, seed=900 + i) for i, ln in enumerate(lens_half)])
print(f"half text: {len(half)} bp in {len(lens_half)} sequences, generated in {time.time() - t0:.0f} s", flush=True)
K = 30
nh = len(half)
cum = np.concatenate([[0], np.cumsum(lens_half)])
iv = [(0, 20000), (int(lens_half[0] * 0.49) - 5000, int(lens_half[0] * 0.49) + 5000), (int(cum[1]) - 5000, int(cum[1]) + 5000),
      (int(cum[12]) + 1000000, int(cum[12]) + 1020000), (nh - 20000 - K, nh - K + 1)]
t0 = time.time()
ixh = g.Index.build(half, lens_half, sampling=1)
print(f"half index ({ixh.info()['n_rows']} rows, 32-bit rows) in {time.time() - t0:.0f} s", flush=True)
want = {(E, bits): ixh.map(K, E, value_bits=bits, intervals=iv) for E, bits in ((0, 8), (0, 16), (1, 16), (2, 8))}
dev = torch.zeros(2 * nh + 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for E, frac in ((0, 1.0), (1, 1.0)):
    for _ in range(2):
        ixh.map_device(dev.data_ptr(), K, E, value_bits=8, stream=st)
    print(f"32-bit rows, half text, K=30 e={E} whole text: search kernel {ixh.kernel_times(1)[0]:.1f} ms = {(nh - K + 1) / ixh.kernel_times(1)[0] / 1e6:.3f} G k-mers/s", flush=True)
ixh.close()
del dev; torch.cuda.empty_cache()   # (the build of 4.32 G rows takes six 35 GB arrays beside the suffix array itself)
full = np.concatenate([half, half]); lens = lens_half + lens_half
del half
t0 = time.time()
ix = g.Index.build(full, lens, sampling=1)
dev = torch.zeros(2 * nh + 16, dtype=torch.uint8, device="cuda:0")
info = ix.info()
print(f"wide index: {info['n_rows']} rows, row_bits {info['row_bits']}, {info['device_bytes'] / 2**30:.1f} GiB, built in {time.time() - t0:.0f} s", flush=True)
assert info["row_bits"] == 64 and info["n_rows"] >= 2**32 - 1
iv2 = iv + [(a + nh, b + nh) for a, b in iv[:-1]] + [(nh - 20000, nh + 20000), (2 * nh - 20000 - K, 2 * nh - K + 1)]
ok = True
for (E, bits), jump in (((0, 8), -1), ((0, 16), -1), ((1, 16), -1), ((1, 16), 0), ((2, 8), -1), ((2, 8), 0)):
    ix.set_tuning(jump=jump)
    t0 = time.time()
    got = ix.map(K, E, value_bits=bits, intervals=iv2)
    dt = time.time() - t0
    mx = 255 if bits == 8 else 65535
    good = True
    for a, b in iv:
        exp = np.minimum(2 * want[(E, bits)][a:b].astype(np.int64), mx)
        for off in (0, nh):
            if off + b > len(got):
                continue
            same = np.array_equal(got[off + a:off + b].astype(np.int64), exp)
            good &= bool(same)
            if not same:
                bad = np.flatnonzero(got[off + a:off + b].astype(np.int64) != exp)
                print(f"MISMATCH K={K} E={E} bits={bits} jump={jump} interval ({a},{b}) copy offset {off}: {len(bad)} positions, first {bad[:5]}", flush=True)
    ok &= good
    tq = ix.last_stats()["detail"].get("table_q", 0)
    print(f"K={K} E={E} bits={bits} jump={'default' if jump < 0 else jump}: {'ok' if good else 'FAILED'} ({dt:.1f} s for the interval call; table q {tq & 255}, jump J {tq >> 8})", flush=True)
nk = 2 * nh - K + 1
ref = {}
for E, frac, settings in (((0, 1.0, (-1,)), (1, 1.0, (-1,))) if quick else ((0, 1.0, (-1,)), (1, 1.0, (-1, 0)), (2, 0.01, (-1, 0)))):
    span = int(nk * frac) // 48 * 48
    kb = ((nk - span) // 2) // 48 * 48
    rng = None if frac >= 1.0 else (kb, kb + span)
    for jump in settings:
        ix.set_tuning(jump=jump)
        dev.zero_()
        for _ in range(2):
            ix.map_device(dev.data_ptr(), K, E, value_bits=8, kmer_range=rng, stream=st)
        ms = ix.kernel_times(1)[0]
        chk = int(dev[:2 * nh].to(torch.int64).sum().item())
        ref.setdefault(E, chk)
        ok &= chk == ref[E]
        det = ix.last_stats()["detail"]
        print(f"64-bit rows K=30 e={E} on {'the whole text' if rng is None else '%.0f %% of the text' % (100 * frac)}, jump={'default' if jump < 0 else jump}: search kernel {ms:.1f} ms = "
              f"{(nk if rng is None else span) / ms / 1e6:.3f} G k-mers/s (correction pass {det.get('correction_us', 0) / 1e3:.1f} ms), checksum {'ok' if chk == ref[E] else 'DIFFERS'}", flush=True)
if True:   # the doubling property on the whole e = 1 vector of the last full pass (jump = 0): both copies agree, nothing odd but saturation
    ix.set_tuning(jump=-1)
    ix.map_device(dev.data_ptr(), K, 1, value_bits=8, stream=st)
    torch.cuda.synchronize()
    a, b = dev[:nh - K + 1], dev[nh:2 * nh - K + 1]
    same = bool(torch.equal(a, b)); even = bool((((dev[:2 * nh] % 2) == 0) | (dev[:2 * nh] == 255)).all().item())
    ok &= same and even
    print(f"e=1 whole vector: copies agree {same}, every count even or saturated {even}", flush=True)
print("WIDE_FIRST_CLASS_OK" if ok else "WIDE_FIRST_CLASS_FAILED", flush=True)
ix.close()
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""An index of more than 2^32 rows on one MI355X (64-bit rows: the reference's 64-bit BWT variants, src/indexing.hpp:158-169).

The text is TWO copies of the same 12-sequence, ~2.16 Gbp synthetic genome (24 sequences, > 4.3 G rows), so the expected result
needs no second implementation: every k-mer occurs exactly twice as often as in the single copy, i.e.
c_wide[j] == min(MAX, 2 * c_half[j]) with c_half from the 32-bit path of the same library on the half text (itself pinned by
the parity suite).  Checked on intervals at the N-block edges, a sequence boundary, the copy boundary and the end of the text,
for K=30 e=0 (both counter widths) and e=1."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import genmap_amd as g
from genmap_amd import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
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
ixh = g.Index.build(half, lens_half, sampling=0)
assert ixh.info()["row_bits"] == 32
print(f"half index ({ixh.info()['n_rows']} rows, 32-bit rows) in {time.time() - t0:.0f} s", flush=True)
want = {}
for E, bits in ((0, 8), (0, 16), (1, 16)):
    want[(E, bits)] = ixh.map(K, E, value_bits=bits, intervals=iv)
ixh.close()
full = np.concatenate([half, half])
lens = lens_half + lens_half
del half
t0 = time.time()
ix = g.Index.build(full, lens, sampling=0)
info = ix.info()
print(f"wide index: {info['n_rows']} rows, row_bits {info['row_bits']}, {info['device_bytes'] / 2**30:.1f} GiB, built in {time.time() - t0:.0f} s", flush=True)
assert info["row_bits"] == 64 and info["n_rows"] >= 2**32 - 1
iv2 = iv + [(a + nh, b + nh) for a, b in iv[:-1]] + [(nh - 20000, nh + 20000), (2 * nh - 20000 - K, 2 * nh - K + 1)]
ok = True
for E, bits in ((0, 8), (0, 16), (1, 16)):
    t0 = time.time()
    got = ix.map(K, E, value_bits=bits, intervals=iv2)
    dt = time.time() - t0
    mx = 255 if bits == 8 else 65535
    for a, b in iv:
        exp = np.minimum(2 * want[(E, bits)][a:b].astype(np.int64), mx)
        for off in (0, nh):
            if off + b > len(got):
                continue
            same = np.array_equal(got[off + a:off + b].astype(np.int64), exp)
            ok &= bool(same)
            if not same:
                bad = np.flatnonzero(got[off + a:off + b].astype(np.int64) != exp)
                print(f"MISMATCH K={K} E={E} bits={bits} interval ({a},{b}) copy offset {off}: {len(bad)} positions, first {bad[:5]}", flush=True)
    print(f"K={K} E={E} bits={bits}: {'ok' if ok else 'FAILED'} ({dt:.1f} s for the interval call)", flush=True)
t0 = time.time()
out = ix.map(K, 0, value_bits=8)
dt = time.time() - t0
print(f"whole text K=30 e=0 through gm_map (host vector, PCIe included): {len(full) - K + 1} k-mers in {dt:.2f} s; search kernel {ix.last_stats()['search_ms']:.1f} ms", flush=True)
# doubling property on the whole vector: the two copies agree position by position and nothing is odd except saturation
a, b = out[:nh - K + 1], out[nh:2 * nh - K + 1]
ok &= bool(np.array_equal(a, b)) and bool(((out % 2 == 0) | (out == 255)).all())
if len(sys.argv) > 2:   # e.g. "14,15,16": forced lengths of the q-mer table (32-byte entries with 64-bit rows; the default stops at 14)
    import torch
    dev = torch.zeros(len(full) + 16, dtype=torch.uint8, device="cuda:0")
    for q in map(int, sys.argv[2].split(",")):
        ix.set_tuning(qtable=q)
        for _ in range(2):
            ix.map_device(dev.data_ptr(), K, 0, value_bits=8, stream=torch.cuda.current_stream().cuda_stream)
        same = bool(np.array_equal(dev[:len(out)].cpu().numpy(), out))
        ok &= same
        print(f"qtable={q}: search kernel {ix.kernel_times(1)[0]:.1f} ms, table_q {ix.last_stats()['detail']['table_q'] & 255}, same result: {same}", flush=True)
# cooperative reads of the 64-byte blocks (groups of four lanes, the default since r04) against one lane per block, and an e = 1 figure
import torch
dev = torch.zeros(len(full) + 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for coop in (1, 0):
    ix.set_tuning(qtable=-1, coop=coop)
    for _ in range(2):
        ix.map_device(dev.data_ptr(), K, 0, value_bits=8, stream=st)
    same = bool(np.array_equal(dev[:len(out)].cpu().numpy(), out))
    ok &= same
    print(f"K=30 e=0 whole text, coop={coop}: search kernel {ix.kernel_times(1)[0]:.1f} ms = {(len(full) - K + 1) / ix.kernel_times(1)[0] / 1e6:.2f} G k-mers/s, same result: {same}", flush=True)
    nk = len(full) - K + 1
    rng = (nk // 2 // 8 * 8, nk // 2 // 8 * 8 + nk // 100 // 8 * 8)
    ref = None
    for _ in range(2):
        ix.map_device(dev.data_ptr(), K, 1, value_bits=8, kmer_range=rng, stream=st)
    e1 = dev[rng[0]:rng[1]].cpu().numpy()
    if coop == 1: e1_first = e1
    else: ok &= bool(np.array_equal(e1, e1_first))
    print(f"K=30 e=1 on 1 % of the text, coop={coop}: search kernel {ix.kernel_times(1)[0]:.1f} ms = {(rng[1] - rng[0]) / ix.kernel_times(1)[0] / 1e6:.3f} G k-mers/s", flush=True)
ix.set_tuning(coop=-1)
print("WIDE_ROWS_OK" if ok else "WIDE_ROWS_FAILED", flush=True)
ix.close()
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""bench.py -- (k,e)-mappability throughput on MI355X, on BASELINE.json's own configuration.

  python bench.py [--gpus N --steps K --warmup W] [--workload grch38|chr1|ecoli] [--scale f] [--fasta F.fa] [--K 30 --E 0]

Default workload: S3 "grch38-like" (24 sequences with the GRCh38 chromosome lengths, 3,088,269,832 bp, Dna5) -- the
"3.1 Gbp index" the metric is quoted on (BASELINE.json configs[2]/[3]); it fits one GPU.  --fasta replaces the synthetic
text by a real FASTA file.  Protocol of /root/reference/benchmarks/bench.sh:24-43: both strands, -fs (8-bit
frequencies), raw output; the map computation alone is timed, index resident (load / write are reported separately by
`genmap map -v`).

A "step" is one complete computeMappability pass (src/algo.hpp:405-483) over the text with the index already in HBM:
clear of the accumulators, the search kernel, finalize/resetLimits, and -- for N > 1 -- the gather of the ranks' chunks of
the frequency vector to rank 0.  Each rank holds a full index replica and computes interleaved chunks of whole k-mer
blocks (strong scaling: the genome is fixed).  The headline line is K=30, e=2 -- BASELINE.json's one-GPU configuration on the
3.1 Gbp index (C3); the same JSON line carries sub-records for (30,0), (30,1) and (100,1) = config C4, measured in the same
run AT EVERY N, each with its
own roofline (numerator counted by the instrumented twin library after the timed part, shard by shard for N > 1;
denominator = HIP-event time of the search kernel over the timed steps, the slowest rank's for N > 1) and CPU baseline.

  python bench.py --workload bacteria5 [--gpus 2]      config C5: five FASTA files, K=24 e=1, --exclude-pseudo frequencies
                                                       (+ csv location lists), contiguous shares per file and rank
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


_T0 = time.time()
TRACE = False


def mark(*a):
    """stage marker of EVERY rank with a time stamp (--trace): where a multi-rank run stands when it stops making progress"""
    if TRACE:
        print(f"[bench r{os.environ.get('RANK', '0')} +{time.time() - _T0:7.1f}s]", *a, file=sys.stderr, flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """physical cores of the host (sockets x cores: distinct (physical id, core id) pairs of /proc/cpuinfo); os.cpu_count() counts hardware threads"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


class CpuBaseline:
    """The oracle (a port of the reference algorithm, kind="port") on this box's host cores, on a bounded sample of the SAME positions the
    GPU rate covers: seeded random intervals of 1,000 positions spread over every sequence of the text (repeat families, unique
    sequence and N runs in the text's own proportions -- round 4 sampled a prefix behind the leading N block, which held none of the
    repeats).  The all-thread figure takes as many of the intervals as make the timed run last about 10 s; the one-thread figure takes a
    prefix of the same list.  The output vector is allocated and cleared outside the timed calls.  Built for THIS host with -O3
    -march=native (POPCNT, as the reference ships it: README.rst:60-62), 64-byte rank blocks (one cache line per query), rank arrays built
    and first touched by the threads that use them."""

    INTERVAL = 1000
    TARGET_S = 8.0     # length of the all-thread timed run (the one-thread run: half of it); --cpu-seconds

    def __init__(self, codes, lens, bwt, threads, seq_file_id=None, sa=None):
        sys.path.insert(0, str(ROOT / "tests"))
        import helpers as H
        t0 = time.time()
        so = H.build_oracle_native()
        self.build = "-O3 -march=native" if so else "-O3 -mpopcnt (portable build: the native build failed on this host)"
        self.ora = H.OracleIndex(codes, lens, keep_sa=False, bwt=bwt, sa=sa, lib=H.oracle_lib(so))
        self.fid = seq_file_id
        self.n, self.threads, self.model = len(codes), threads, cpu_model()
        self.lens = [int(x) for x in lens]
        self.cum = np.concatenate([[0], np.cumsum(np.asarray(self.lens, dtype=np.int64))])
        self.out = {8: np.zeros(self.n, np.uint8), 16: np.zeros(self.n, np.uint16)}
        self.skip_clear = hasattr(self.ora.lib, "gmo_set_skip_clear")   # (a process-wide switch of the oracle: set around the timed calls only)
        log(f"cpu_baseline: oracle ({self.build}) adopted the GPU-built BWTs in {time.time() - t0:.1f} s on {self.model}, {threads} threads")

    def intervals(self, K, count, seed=20260929):
        """`count` seeded intervals of INTERVAL positions, dealt to the sequences in proportion to their lengths, in a random ORDER (a prefix
        of the list is itself spread over the whole text)"""
        rng = np.random.default_rng(seed)
        total = int(self.cum[-1])
        iv = []
        for _ in range(count):
            p = int(rng.integers(0, max(1, total - self.INTERVAL - K)))
            q = int(np.searchsorted(self.cum, p, side="right")) - 1
            b = min(p + self.INTERVAL, int(self.cum[q + 1]))
            if b - p >= 64:
                iv.append((p, b))
        return iv

    def _timed(self, K, E, threads, target, first, **kw):
        """grow the prefix of the interval list until the timed run takes >= target seconds"""
        bits = kw.pop("value_bits", 8)
        pool = self.intervals(K, 400_000)
        m, dt, npos = max(1, first), 0.0, 0
        while True:
            iv = sorted(pool[:m])
            merged = []
            for a, b in iv:
                if merged and a < merged[-1][1]:
                    merged[-1] = (merged[-1][0], max(b, merged[-1][1]))
                else:
                    merged.append((a, b))
            npos = sum(b - a for a, b in merged)
            try:
                if self.skip_clear:
                    self.ora.lib.gmo_set_skip_clear(1)
                t0 = time.time()
                self.ora.mappability(K, E, value_bits=bits, threads=threads, intervals=merged, out=self.out[bits], **kw)
                dt = time.time() - t0
            finally:
                if self.skip_clear:
                    self.ora.lib.gmo_set_skip_clear(0)
            if self.skip_clear:
                for a, b in merged:
                    self.out[bits][a:b + K] = 0
            if dt >= target or m >= len(pool):
                break
            m = int(min(len(pool), max(m * 2, m * (target * 1.3) / max(dt, 1e-3))))
        return m, npos, dt

    def run_c5(self, K, E, slices):
        """config C5: the --exclude-pseudo pass (16-bit) of every FASTA file of the index on seeded random intervals of each file; the same
        intervals for the all-thread figure and (a prefix per file) the one-thread figure"""
        def timed(threads, target, per_file):
            while True:
                t_sum, npos = 0.0, 0
                for q, (fs, ns, tl) in enumerate(slices):
                    rng = np.random.default_rng(1000 + q)
                    st = np.sort(rng.integers(0, max(1, tl - self.INTERVAL - K), per_file))
                    iv = []
                    for a in st:
                        a = int(a) if not iv or int(a) >= iv[-1][1] else iv[-1][1]
                        b = min(a + self.INTERVAL, tl - K + 1)
                        if b > a:
                            iv.append((a, b))
                    npos += sum(b - a for a, b in iv)
                    t0 = time.time()
                    self.ora.mappability(K, E, first_seq=fs, n_seq=ns, text_begin=int(self.cum[fs]), text_len=tl, value_bits=16, exclude_pseudo=True, directory=True,
                                         threads=threads, intervals=iv, seq_file_id=self.fid)
                    t_sum += time.time() - t0
                if t_sum >= target or per_file * self.INTERVAL >= max(tl for _, _, tl in slices):
                    return per_file, npos, t_sum
                per_file = int(max(per_file * 2, per_file * (target * 1.3) / max(t_sum, 1e-3)))
        m, npos, dt = timed(self.threads, self.TARGET_S, 50)
        m1, npos1, dt1 = timed(1, self.TARGET_S / 2, max(1, m // max(8, self.threads // 2)))
        return {"value": npos / dt, "unit": "k-mers/s", "cores": physical_cores(), "threads": self.threads, "kind": "port", "build": self.build, "cpu_model": self.model,
                "threads_1": {"value": npos1 / dt1, "unit": "k-mers/s", "sample": f"{m1} intervals per file: {npos1} positions, {dt1:.1f} s"},
                "sample": f"{npos} k-mer positions in {m} seeded random intervals of {self.INTERVAL} per FASTA file ({len(slices)} files) of the same index, K={K} E={E}, --exclude-pseudo, both strands, {dt:.1f} s"}

    def run(self, K, E, first_guess, **kw):
        m, npos, dt = self._timed(K, E, self.threads, self.TARGET_S, max(1, first_guess // self.INTERVAL), **kw)
        m1, npos1, dt1 = self._timed(K, E, 1, self.TARGET_S / 2, max(1, m // max(8, self.threads // 2)), **kw)
        return {"value": npos / dt, "unit": "k-mers/s", "cores": physical_cores(), "threads": self.threads, "kind": "port", "build": self.build, "cpu_model": self.model,
                "threads_1": {"value": npos1 / dt1, "unit": "k-mers/s", "sample": f"the first {m1} of the same intervals: {npos1} positions, {dt1:.1f} s"},
                "sample": f"{npos} k-mer positions in {m} seeded random intervals of {self.INTERVAL} spread over all {len(self.lens)} sequences of the same index, K={K} E={E}, both strands, {dt:.1f} s (output vector cleared outside the timed call)"}


def read_fasta(path):
    """FASTA -> (codes, lengths) through the host library's reader (the `genmap index` rules, src/indexing.hpp:209-275)"""
    import ctypes as C
    lib = C.CDLL(str(ROOT / "genmap_amd" / "lib" / "libgenmap_host.so"))
    lib.gmh_read_fasta.restype = C.c_int
    nseq, total, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    if lib.gmh_read_fasta(str(path).encode(), None, None, None, C.c_uint64(0), C.byref(nseq), C.byref(total), C.byref(nb)):
        raise SystemExit(f"cannot read {path}")
    codes = np.empty(total.value, np.uint8)
    lens = np.empty(nseq.value, np.uint64)
    names = C.create_string_buffer(nb.value + 1)
    if lib.gmh_read_fasta(str(path).encode(), codes.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), names, C.c_uint64(nb.value + 1),
                          C.byref(nseq), C.byref(total), C.byref(nb)):
        raise SystemExit(f"cannot read {path}")
    return codes, [int(x) for x in lens]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="grch38", help="grch38 | chr1 | ecoli (one FASTA, frequency pass) | bacteria5 (config C5: five FASTA files, --exclude-pseudo, csv)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--fasta", default=None, help="real FASTA file instead of the synthetic workload")
    ap.add_argument("--K", type=int, default=0, help="default 30 (24 for bacteria5)")
    ap.add_argument("--E", type=int, default=-1, help="default 2 = BASELINE.json's one-GPU 3.1 Gbp configuration C3 (1 for bacteria5)")
    ap.add_argument("--sub", default="30,0:20;30,1:3;100,1:3", help="sub-records 'K,E:steps;...' measured after the headline (one warm-up step each); '' = none")
    ap.add_argument("--block-bytes", type=int, default=0)
    ap.add_argument("--infix", type=int, default=0, help="common-infix length (SearchParams.overlap); 0 = library default")
    ap.add_argument("--sampling", type=int, default=1, help="1: keep the suffix array resident (narrow nodes are verified against the text); 0: rank queries only; "
                    "2..64: sampled suffix array (the reference's -S; bacteria5 only needs it for locate)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the single-GPU rehearsal of the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--comm", default="p2p", choices=["p2p", "collective"], help="N > 1: chunks pushed into the root's vector with peer DMA copies "
                    "overlapping the search kernel (falls back to the collective when IPC is not available or far slower than predicted), or one gather collective per step")
    ap.add_argument("--watchdog", type=float, default=900.0, help="N > 1: seconds one measured record may take before the run gives up")
    ap.add_argument("--verify", action="store_true", help="N > 1: after the timed steps rank 0 recomputes the whole vector alone and compares it with the gathered one")
    ap.add_argument("--no-host-rate", action="store_true", help="skip value_host (its gm_map call launches the search kernel in four pieces: keeps a rocprofv3 kernel trace of the timed launches clean)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
    ap.add_argument("--c5-per-file", action="store_true", help="bacteria5 at N = 1: one launch per FASTA file (round 5) instead of one for all of them")
    ap.add_argument("--no-csv", action="store_true", help="bacteria5: skip the csv location lists (gm_locate), time the --exclude-pseudo frequencies only")
    ap.add_argument("--trace", type=float, default=0.0, help="print stage markers from every rank and, after this many seconds, every thread's Python stack (diagnosis of a stalled multi-rank run)")
    ap.add_argument("--allow-mixed-features", action="store_true", help="N > 1: time the run even if the ranks' index replicas differ (records, table length)")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip the verified (100,1) step through both transports that precedes every timing")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="length of the all-thread run of cpu_baseline (the one-thread run takes half of it)")
    ap.add_argument("--extra-configs", default="chr1,bacteria5", help="default workload at N = 1 only: BASELINE configs measured on an index of their own by a child process each "
                    "(chr1: C2, K=30 e=0 on the chr1-like text; bacteria5: C5) and reported as sub-records; '' = none")
    ap.add_argument("--no-traffic", action="store_true", help="skip roofline.traffic (two extra processes under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE after the timed part, N = 1 only)")
    ap.add_argument("--tune", default="", help="experiments: scheduling knobs 'name=value,...' set on the index before anything is timed (gm_index_set_tuning; results never depend on them); recorded in config.tune")
    ap.add_argument("--protocol", default="", choices=["", "reference"], help="reference: the reference's own benchmark protocol (benchmarks/bench.sh:35-43: (5,0), (6,0), (101,0..4)) "
                    "on the same index, one pass each -- tools/protocol_reference.py; E = 4 takes minutes, never part of the default line")
    args = ap.parse_args()
    if args.protocol == "reference":
        import runpy
        sys.argv = [str(ROOT / "tools" / "protocol_reference.py"), "--workload", args.workload, "--scale", str(args.scale)]
        runpy.run_path(sys.argv[0], run_name="__main__")
        return
    if args.trace > 0:
        global TRACE
        TRACE = True
        import faulthandler
        faulthandler.dump_traceback_later(args.trace, repeat=True, file=sys.stderr)
    c5 = args.workload == "bacteria5" and not args.fasta
    CpuBaseline.TARGET_S = max(0.5, args.cpu_seconds)
    args.K = args.K or (24 if c5 else 30)
    args.E = args.E if args.E >= 0 else (1 if c5 else 2)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import genmap_amd as g
    from genmap_amd import synth
    if not torch.cuda.is_available() or g.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    cdev = dev if args.backend == "nccl" else "cpu"   # where the small control tensors of the collectives live
    rccl_ranks = dist.get_world_size() if (world > 1 and args.backend == "nccl") else 0

    t0 = time.time()
    files5 = fid = None
    if args.fasta:
        codes, lens = read_fasta(args.fasta)
        desc, data = f"{os.path.basename(args.fasta)} {len(codes)} bp in {len(lens)} sequences", "real FASTA"
    elif c5:
        files5 = synth.bacteria5(args.scale)
        codes = np.concatenate([c for _, recs in files5 for _, c in recs])
        lens = [len(c) for _, recs in files5 for _, c in recs]
        fid = np.array([f for f, (_, recs) in enumerate(files5) for _ in recs], dtype=np.uint32)
        desc, data = f"S5 five-bacteria-like {len(codes)} bp in {len(files5)} FASTA files / {len(lens)} sequences Dna5", "synthetic"
    else:
        codes, lens, desc = synth.workload(args.workload, args.scale)
        data = "synthetic"
    n = int(len(codes))
    log(f"workload {desc}: ready in {time.time() - t0:.1f} s")
    t0 = time.time()
    if args.same_device and world > 1:   # rehearsal on one device: the builder's temporaries do not fit twice, build in turn
        ix = None
        for r in range(world):
            if r == rank:
                ix = g.Index.build(codes, lens, sampling=args.sampling, block_bytes=args.block_bytes, device=local_rank)
                mark("index built")
            dist.barrier()
    else:
        ix = g.Index.build(codes, lens, sampling=args.sampling, block_bytes=args.block_bytes, device=local_rank)
    t_build = time.time() - t0
    if args.tune:
        ix.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",") if kv})
    info = ix.info()
    log(f"index built on the GPU in {t_build:.1f} s: {info['n_rows']} rows, {info['block_bytes']}-B blocks, {info['device_bytes'] / 2**30:.2f} GiB")

    from genmap_amd.distributed import PeerGather, ShardPlan, gather_chunks, gather_frequency, gather_locations, max_shard_len, shard_ranges
    stream = torch.cuda.current_stream().cuda_stream
    out = None

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank_of(values):
        if world == 1:
            return None
        mine = torch.tensor(values, dtype=torch.float64, device=cdev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        return [[float(v) for v in x.tolist()] for x in allv]

    class Watchdog:
        """a multi-rank run that stops making progress (a peer that died, a transfer that never completes) must not sit in a
        barrier until the launcher's own limit: say so and leave"""

        def __enter__(self):
            self.t = None
            if world > 1:
                import threading

                def _fire():
                    print(f"[bench] rank {rank}: no result after {args.watchdog} s in the multi-rank run -- giving up", file=sys.stderr, flush=True)
                    os._exit(4)
                self.t = threading.Timer(args.watchdog, _fire)
                self.t.daemon = True
                self.t.start()
            return self

        def __exit__(self, *exc):
            if self.t:
                self.t.cancel()
            return False

    features = {"per_rank": None}

    def check_features(K, E):
        """N > 1: every rank reports what its index replica got (verification records, suffix array, the q-mer table / jump length of this
        call -- the 69 GB table of all 16-mers shrinks by itself when a device is short of memory); ranks with different features compute the
        same result at different speeds, which would look like a scaling problem: refuse to time such a mix"""
        if world == 1:
            return
        tq = ix.last_stats()["detail"].get("table_q", 0)
        mine = {"rank": rank, "verify_records": info["verify_records"], "sampling": info["sampling"], "q": tq & 255, "J": tq >> 8, "device_gib": round(ix.info()["device_bytes"] / 2**30, 2)}
        allf = [None] * world
        dist.all_gather_object(allf, mine)
        features["per_rank"] = allf
        log(f"index features per rank for K={K} E={E}: {allf}")
        key = lambda f: (f["verify_records"], f["sampling"], f["q"], f["J"])
        if len({key(f) for f in allf}) > 1 and not args.allow_mixed_features:
            raise SystemExit(f"ranks got different index features {allf}: not timing a mix (--allow-mixed-features overrides)")

    def measure(K, E, steps, warmup, verify=None, comm_mode=None):
        """W warm-up steps, then exactly `steps` timed steps between barrier + synchronize; returns the record."""
        nonlocal out
        verify = args.verify if verify is None else verify
        comm_mode = comm_mode or args.comm
        infix = args.infix or g.tuned_infix_length(K, E)
        num_kmers = n - K + 1
        plan = ShardPlan(num_kmers, K - infix + 1, world)
        if out is None or out.numel() < plan.padded_len(n):
            out = torch.zeros(plan.padded_len(n), dtype=torch.uint8, device=dev)        # -fs: 8-bit frequencies
        comm = {"wait_s": 0.0, "mode": "none", "note": None}
        pg = None
        mark(f"measure K={K} E={E}: plan {plan.nchunks} chunks of {plan.chunk_len} positions")
        if world > 1 and comm_mode == "p2p":
            pg = PeerGather(plan, n, 1, rank, local_rank, dist, mark=mark)
            mark(f"PeerGather ready: ok={pg.ok} launches={pg.launches}")
            if not pg.ok:
                log("peer copies not available between these ranks: falling back to the gather collective")
                comm["note"] = "peer copies not available (IPC open / probe failed): gather collective"
                pg.close(); pg = None

        def one_step():
            if pg is None:
                ix.map_device(out.data_ptr(), K, E, infix=args.infix, value_bits=8, chunks=plan.chunk_arg(rank), stream=stream)
                if world > 1:
                    t1 = time.perf_counter()
                    gather_chunks(out, plan, rank, dist, stage_on_host=(args.backend != "nccl"))
                    comm["wait_s"] += time.perf_counter() - t1
                return
            pieces = plan.sub_ranges(pg.launches)
            share = (pieces[0][0], pieces[-1][1])         # the rank's whole share: ONE clear and ONE correction pass for its launches (GM_MAP_FLAG_PIECE)
            for sub in pieces:                            # the chunks of one launch travel while the next launch computes
                ix.map_device(pg.local_ptr, K, E, infix=args.infix, value_bits=8, kmer_range=sub, chunks=plan.chunk_arg(rank), stream=stream, piece_of=share)
                ev = torch.cuda.Event()
                ev.record()
                pg.push(sub, ev)
            t1 = time.perf_counter()
            torch.cuda.current_stream().synchronize()
            pg.finish()
            comm["wait_s"] += time.perf_counter() - t1

        if pg is not None:
            # Two untimed steps first (a cold one, then a timed probe): if the peer copies take far longer than the search plus a transfer at a fifth of one xGMI
            # link would (a fabric that routes through host memory, a driver that serialises the copies behind the persistent
            # kernel), every rank switches to the gather collective and the line says so.
            one_step()   # cold: builds the q-mer tables and the workspaces of this (K, E)
            sync()
            t1 = time.perf_counter()
            one_step()
            sync()
            wall = time.perf_counter() - t1
            mine = float(np.sum(ix.kernel_times(pg.launches))) * 1e-3
            predicted = mine + (pg.nbytes / world) / 10e9
            slow = max_over_ranks(1.0 if wall > 10.0 * predicted + 0.05 else 0.0)
            mark(f"p2p probe step: {wall * 1e3:.1f} ms, predicted {predicted * 1e3:.1f} ms")
            if slow > 0:
                comm["note"] = f"first p2p step took {wall * 1e3:.0f} ms against {predicted * 1e3:.0f} ms predicted: switched to the gather collective"
                log(comm["note"])
                pg.close(); pg = None
        if world > 1:
            comm["mode"] = "peer DMA copies overlapping compute" if pg else "gather collective"
        comm["wait_s"] = 0.0
        for i in range(warmup):
            one_step()
            mark(f"warm-up step {i} issued")
        sync()
        mark("warm-up done")
        if warmup > 0 or pg is not None:
            check_features(K, E)
        warm_wait = comm["wait_s"]
        t0 = time.perf_counter()
        for i in range(steps):
            one_step()
            mark(f"step {i} issued")
        sync()
        mark("timed steps done")
        dt = time.perf_counter() - t0
        launches = pg.launches if pg else 1
        kms = ix.kernel_times(min(steps * launches, 64 // launches * launches))   # HIP events around the search kernel of each timed launch, on the launch stream
        kms = [float(np.sum(kms[i * launches:(i + 1) * launches])) for i in range(len(kms) // launches)]
        verified = None
        if verify and world > 1:
            sync()
            if rank == 0:
                if pg:
                    got = torch.empty(pg.nbytes, dtype=torch.uint8, device=dev)
                    pg.assemble(got.data_ptr())
                    got = got[:n]
                else:
                    got = out[:n]
                ref = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
                ix.map_device(ref.data_ptr(), K, E, infix=args.infix, value_bits=8, stream=stream)
                torch.cuda.synchronize()
                same = bool(torch.equal(got, ref[:n]))
                log(f"verify K={K} E={E}: gathered vector {'==' if same else '!='} single-rank vector ({comm['mode']})")
                if not same and verify != "report":
                    raise SystemExit("gathered result differs from the single-rank result")
                del got, ref
                verified = same
            sync()
            verified = bool(max_over_ranks(1.0 if verified else 0.0)) if verified is not None or rank != 0 else None
        if pg:
            pg.close()
        dt = max_over_ranks(dt)
        my_ms = float(np.mean(kms))
        corr_ms = ix.last_stats()["detail"].get("correction_us", 0) / 1e3 if E >= 1 else 0.0   # the step's one correction pass (beside the main search)
        pr = per_rank_of([my_ms, (comm["wait_s"] - warm_wait) / max(1, steps) * 1e3, corr_ms])
        return {"K": K, "E": E, "infix": infix, "num_kmers": num_kmers, "steps": steps, "warmup": warmup, "dt": dt,
                "kernel_ms": max(r[0] for r in pr) if pr else my_ms, "kernel_ms_min": float(np.min(kms)), "plan": plan,
                "comm_mode": comm["mode"], "comm_note": comm["note"], "per_rank": pr, "verified": verified, "correction_ms": corr_ms}

    def host_rate(K, E):
        """PCIe-inclusive rates of the drop-in call gm_map (host result vector), never `value`: into ordinary (pageable) memory,
        where the runtime stages the copy, and into a page-locked vector (gm_host_pin), where the pieces travel by DMA while
        the next piece is searched"""
        buf = np.zeros(n, dtype=np.uint8)
        t0 = time.perf_counter()
        ix.map(K, E, infix=args.infix, value_bits=8, out=buf)
        pageable = (n - K + 1) / (time.perf_counter() - t0)
        pinned = None
        try:
            g.host_pin(buf)
            ix.map(K, E, infix=args.infix, value_bits=8, out=buf)   # first touch of the mapping
            t0 = time.perf_counter()
            ix.map(K, E, infix=args.infix, value_bits=8, out=buf)
            pinned = (n - K + 1) / (time.perf_counter() - t0)
            g.host_unpin(buf)
        except Exception as e:
            log("pinned host rate failed:", e)
        return pageable, pinned

    # ---- config C5: one pass = every FASTA file's --exclude-pseudo frequencies (+ csv location lists), contiguous shares ----
    def measure_c5(K, E, steps, warmup):
        slices, first = [], 0
        for _, recs in files5:
            tl = sum(len(c) for _, c in recs)
            slices.append((first, len(recs), tl)); first += len(recs)
        # N = 1: the five files in ONE launch of the persistent kernel (what gm_map_files does: a value depends on the k-mer and on the whole index, not on
        # the file it is computed with; positions across a file boundary cross a sequence boundary and are zeroed by resetLimits) -- the five launches
        # of 6.4 ms filled and drained 256 CUs for 4 Mbp each (profiles/r05).  N > 1 keeps a share per file and rank.
        files = list(slices)
        if world == 1 and not args.c5_per_file:
            slices = [(0, first, sum(tl for _, _, tl in files))]
        infix = args.infix or g.tuned_infix_length(K, E, locating=True)   # --exclude-pseudo and csv locate: the library's block shape for those calls
        step_size = K - infix + 1
        total_kmers = sum(tl - K + 1 for _, _, tl in files)
        bufs = []
        for _, _, tl in slices:
            rg = shard_ranges(tl - K + 1, step_size, world)
            bufs.append((rg, torch.zeros(tl + max_shard_len(rg) + 16, dtype=torch.uint16, device=dev)))
        t_csv = {"s": 0.0, "occ": 0}

        def one_step(csv):
            for (fs, ns, tl), (rg, buf) in zip(slices, bufs):
                ix.map_device(buf.data_ptr(), K, E, first_seq=fs, n_seq=ns, infix=args.infix, value_bits=16, exclude_pseudo=True, seq_file_id=fid,
                              kmer_range=rg[rank] if world > 1 else None, stream=stream)
                if world > 1:
                    torch.cuda.current_stream().synchronize()
                    gather_frequency(buf, rg, rank, world, dist, stage_on_host=(args.backend != "nccl"))
            if csv:
                t1 = time.perf_counter()
                for (fs, ns, tl), (rg, buf) in zip(slices, bufs):
                    loc = ix.locate(K, E, first_seq=fs, n_seq=ns, infix=args.infix, kmer_range=rg[rank] if world > 1 else None)
                    merged = gather_locations(loc, rank, world, dist, device=dev if args.backend == "nccl" else None) if world > 1 else loc
                    if rank == 0:
                        t_csv["occ"] += len(merged[2]) + len(merged[4])
                t_csv["s"] += time.perf_counter() - t1

        for _ in range(warmup):
            one_step(False)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(False)
        sync()
        dt = max_over_ranks(time.perf_counter() - t0)
        kms = ix.kernel_times(min(64, steps * len(slices)))
        my_ms = float(np.sum(kms)) / max(1, len(kms) // len(slices))
        pr = per_rank_of([my_ms, 0.0])
        rec = {"K": K, "E": E, "infix": infix, "num_kmers": total_kmers, "steps": steps, "warmup": warmup, "dt": dt, "kernel_ms": max(r[0] for r in pr) if pr else my_ms,
               "per_rank": pr, "csv": None, "slices": slices, "ranges": [rg for rg, _ in bufs]}
        if not args.no_csv:
            sync()
            t0 = time.perf_counter()
            one_step(True)
            sync()
            dtc = max_over_ranks(time.perf_counter() - t0)
            rec["csv"] = {"value": total_kmers / dtc, "unit": "k-mers/s", "ms_per_pass": dtc * 1e3, "locate_ms": max_over_ranks(t_csv["s"]) * 1e3, "occurrences": t_csv["occ"],
                          "note": "one pass incl. gm_locate of every file: location lists sorted on the device and delivered to HOST memory (PCIe-inclusive), gathered to rank 0"}
        return rec

    if c5:
        with Watchdog():
            head = measure_c5(args.K, args.E, args.steps, args.warmup)
        subs, value_host, value_host_pinned, host_cfg = [], None, None, None
    else:
        preflight = None
        if world > 1 and not args.no_preflight:
            # Before anything is timed: ONE step of config C4 (K=100, e=1) through each transport, the gathered vector compared with rank 0's
            # own single-rank computation.  A transport that fails (or cannot be set up) is not timed; if both fail the run stops here.
            preflight = {}
            for mode in ("p2p", "collective"):
                with Watchdog():
                    r = measure(100, 1, 1, 0, verify="report", comm_mode=mode)
                ok = bool(r["verified"]) and (mode == "collective" or r["comm_mode"].startswith("peer"))
                preflight[mode] = "verified" if ok else ("not available: " + (r["comm_note"] or "fell back") if r["verified"] else "WRONG RESULT")
            log(f"preflight (K=100 e=1, one verified step per transport): {preflight}")
            if preflight["p2p"] != "verified" and args.comm == "p2p":
                args.comm = "collective"
            if preflight["collective"] == "WRONG RESULT" and args.comm == "collective":
                raise SystemExit("preflight: no transport delivers the single-rank result")
            log(f"timing with --comm {args.comm}")
        with Watchdog():
            head = measure(args.K, args.E, args.steps, args.warmup)
        subs = []
        for item in filter(None, args.sub.split(";")):
            ke, st = item.split(":")
            K, E = map(int, ke.split(","))
            if (K, E) == (args.K, args.E):
                continue
            with Watchdog():
                subs.append(measure(K, E, int(st), 1))
            log(f"sub-record K={K} E={E}: {subs[-1]['dt'] / subs[-1]['steps'] * 1e3:.1f} ms/step")
        # PCIe-inclusive rate of the drop-in call gm_map, measured where the transfer weighs most: the cheapest pass (e = 0)
        host_cfg = (args.K, 0) if any((r["K"], r["E"]) == (args.K, 0) for r in [head] + subs) else (args.K, args.E)
        value_host, value_host_pinned = host_rate(*host_cfg) if (world == 1 and not args.no_host_rate) else (None, None)

    # ---- the multi-rank part ends here: every rank but 0 releases its index and leaves; rank 0 counts and prints ----
    bwt_host = sa_host = None
    want_twin = not args.no_counters and g.lib_path(True).exists()
    if rank == 0 and (not args.no_cpu_baseline or want_twin):
        bwt_host = ix.export_bwt()
        if (want_twin or (c5 and not args.no_cpu_baseline)) and args.sampling == 1:
            sa_host = ix.export_sa()
        elif want_twin and args.sampling > 1:
            sa_host = ix.export_sa_sampled()   # (mark words, samples)
    ix.close()
    out = None
    torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- roofline numerators: count node steps / distinct rank lines with the instrumented twin (untimed) ----
    # The timed indexes are released first: the twin needs the same HBM (verification records included) to run the same schedule.
    # N > 1: the twin walks the ranks' shards one after the other with the shard arguments the ranks used.
    counted = {}
    sa_cpu = sa_host if (c5 and args.sampling == 1) else None
    if want_twin:
        try:
            bf, br = bwt_host
            if args.sampling > 1:
                ixp = g.Index.from_sampled(bf, br, sa_host[0], sa_host[1], codes, lens, args.sampling, block_bytes=info["block_bytes"], device=local_rank, profiling=True)
            else:
                ixp = g.Index.from_bwt(bf, br, codes, lens, sa_fwd=sa_host, sampling=args.sampling,
                                       block_bytes=info["block_bytes"], device=local_rank, profiling=True)
            sa_cpu = sa_host if (c5 and args.sampling == 1) else None   # (config C5's CPU baseline locates: the oracle adopts the suffix array)
            sa_host = None
            if ixp.info()["verify_records"] != info["verify_records"]:
                log("warning: the instrumented twin did not get the same verification records as the timed index")
            if c5:   # the --exclude-pseudo pass of every file, share by share, with the arguments the ranks used
                tot = None
                tmp = torch.zeros(max(tl for _, _, tl in head["slices"]) + 16 + max(max_shard_len(rg) for rg in head["ranges"]), dtype=torch.uint16, device=dev)
                for (fs, ns, tl), rg in zip(head["slices"], head["ranges"]):
                    for r in range(world):
                        ixp.map_device(tmp.data_ptr(), head["K"], head["E"], first_seq=fs, n_seq=ns, infix=args.infix, value_bits=16, exclude_pseudo=True, seq_file_id=fid,
                                       kmer_range=rg[r] if world > 1 else None, stream=stream)
                        sp = ixp.last_stats()
                        if tot is None:
                            tot = sp
                        else:
                            for k in ("rank_lines", "roots", "node_steps", "kmers"):
                                tot[k] += sp[k]
                            for k in tot["detail"]:
                                tot["detail"][k] = max(tot["detail"][k], sp["detail"][k]) if k in ("max_stack", "table_q") else tot["detail"][k] + sp["detail"][k]
                counted[(head["K"], head["E"])] = tot
            else:
                tmp = torch.zeros(max(r["plan"].padded_len(n) for r in [head] + subs), dtype=torch.uint8, device=dev)
            for rec in ([] if c5 else [head] + subs):
                tot = None
                for r in range(world):
                    ixp.map_device(tmp.data_ptr(), rec["K"], rec["E"], infix=args.infix, value_bits=8, chunks=rec["plan"].chunk_arg(r), stream=stream)
                    sp = ixp.last_stats()
                    if tot is None:
                        tot = sp
                    else:
                        for k in ("rank_lines", "roots", "node_steps", "kmers"):
                            tot[k] += sp[k]
                        for k in tot["detail"]:
                            tot["detail"][k] = max(tot["detail"][k], sp["detail"][k]) if k in ("max_stack", "table_q") else tot["detail"][k] + sp["detail"][k]
                counted[(rec["K"], rec["E"])] = tot
            ixp.close()
            del tmp
        except Exception as e:  # measurement aid only
            log("counter pass failed:", e)

    def roofline(rec):
        bb = info["block_bytes"]
        sp = counted.get((rec["K"], rec["E"]))
        if not sp or not sp["rank_lines"]:
            return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": "search_kernel", "kernel_ms": rec["kernel_ms"]}
        d = sp["detail"]
        # rank blocks + one q-mer table entry per root + text read once per strand (4-bit packed) + 8-bit output
        # + verification: one 32-byte record per row and 8 needle symbols per chunk -- or, without the records, the SA entry
        # per row and 8 needle + 8 text symbols per chunk
        vi = d["verify_items"]
        ver = 32 * vi + 8 * d["verify_chunks"] if info["verify_records"] else 4 * vi + 16 * d["verify_chunks"]
        # (a root that jumps reads one 16-byte table entry per pattern instead of the single q-mer entry: counted as jump_lookups)
        # ... and one 8-byte bitmap word per group of patterns: jump_words)
        alg = bb * sp["rank_lines"] + 16 * (d.get("jump_lookups", 0) or sp["roots"]) + 8 * d.get("jump_words", 0) + n + n + ver
        # the split search (round 6): every node packet is written once by phase A and read once by the walker
        step = rec["K"] - g.tuned_infix_length(rec["K"], rec["E"]) + 1
        pkt_bytes = 16 * (2 + (rec["K"] + step - 1 + 31) // 32)
        packets = d.get("packets", 0)
        alg_strict = alg
        alg += 2 * pkt_bytes * packets
        # N > 1: per GPU -- a rank's share of the bytes over the slowest rank's kernel time, against one GPU's peak
        ach = alg / world / (rec["kernel_ms"] * 1e-3) / 1e9
        # random requests the kernel issues (rank blocks, table entries, records) against the measured ceiling of the memory system
        # for random reads over a large footprint (48.3 G/s whatever the concurrency: profiles/r03/gather2_concurrency.txt); L2 hits
        # are included, so the figure can exceed the ceiling -- the lines actually fetched are in profiles/<round>/final/pmc_by_config.txt
        issued = sp["rank_lines"] + (d.get("jump_lookups", 0) or sp["roots"]) + d.get("jump_words", 0) + vi
        tr = traffic.get((rec["K"], rec["E"]), {})
        tsum = (tr.get("FETCH_SIZE", 0.0) + tr.get("WRITE_SIZE", 0.0)) if len(tr) == 2 else None
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "random_requests_issued_per_s": issued / world / (rec["kernel_ms"] * 1e-3), "random_read_ceiling_per_s": 4.83e10,
                # bytes per launch of the same configuration under rocprofv3 --pmc (FETCH_SIZE + WRITE_SIZE, separate passes, KB x 1024; the
                # counters tally 64 B per request at the fabric side of the L2, whatever the request's width); null when not collected
                "traffic": tsum, "traffic_fetch_bytes": tr.get("FETCH_SIZE"), "traffic_write_bytes": tr.get("WRITE_SIZE"),
                "traffic_over_algorithmic": (tsum / alg) if tsum else None,
                "per_gpu": True, "kernel": "search_kernel" if not packets else "search phase of one pass: expand_kernel + search_kernel<CountEnv<..,2>> (walker) launches of its slices", "kernel_ms": rec["kernel_ms"], "algorithmic_bytes": alg, "rank_lines": sp["rank_lines"],
                "node_packets": packets, "packet_bytes": pkt_bytes, "slices": d.get("slices", 0), "algorithmic_bytes_without_packets": alg_strict,
                "roots": sp["roots"], "node_steps": sp["node_steps"], "node_steps_per_kmer": sp["node_steps"] / rec["num_kmers"],
                "verify_items": d["verify_items"], "verify_chunks": d["verify_chunks"], "jump_lookups": d.get("jump_lookups", 0), "jump_words": d.get("jump_words", 0),
                "lanes_with_node_per_iteration": d["active_lane_sum"] / max(1, d["wave_iterations"])}

    def roofline_c5(rec):
        """--exclude-pseudo pass (FileSetEnv): rank blocks + one table entry per root + every located row (a 4-byte suffix-array entry, or with a
        sampled array an 8-byte mark word per visit, a 32-byte rank block per LF step and the 4-byte sample) + one 4-byte word of the file
        set per located row + the text once per strand + the 16-bit result"""
        sp = counted.get((rec["K"], rec["E"]))
        tr = traffic.get((rec["K"], rec["E"]), {})
        tsum = (tr.get("FETCH_SIZE", 0.0) + tr.get("WRITE_SIZE", 0.0)) if len(tr) == 2 else None
        base = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": tsum, "traffic_fetch_bytes": tr.get("FETCH_SIZE"), "traffic_write_bytes": tr.get("WRITE_SIZE"),
                "kernel": "search_kernel (FileSetEnv), one launch per FASTA file",
                "kernel_ms": rec["kernel_ms"], "per_gpu": True}
        if not sp or not sp["rank_lines"]:
            return dict(base, achieved=None, frac=None)
        d = sp["detail"]
        loc, lf = d.get("located_rows", 0), d.get("lf_steps", 0)
        loc_bytes = 4 * loc if args.sampling == 1 else 8 * (loc + lf) + info["block_bytes"] * lf + 4 * loc
        alg = info["block_bytes"] * sp["rank_lines"] + 16 * sp["roots"] + loc_bytes + 4 * loc + 2 * n + 2 * n
        ach = alg / world / (rec["kernel_ms"] * 1e-3) / 1e9
        return dict(base, achieved=ach, frac=ach / HBM_PEAK_GBS, algorithmic_bytes=alg, rank_lines=sp["rank_lines"], roots=sp["roots"], node_steps=sp["node_steps"],
                    located_rows=loc, lf_steps=lf, random_requests_issued_per_s=(sp["rank_lines"] + sp["roots"] + 2 * loc + 2 * lf) / world / (rec["kernel_ms"] * 1e-3),
                    random_read_ceiling_per_s=4.83e10, lanes_with_node_per_iteration=d["active_lane_sum"] / max(1, d["wave_iterations"]))

    # ---- roofline.traffic: FETCH_SIZE and WRITE_SIZE of the search kernel, one rocprofv3 --pmc pass each (they do not fit one pass:
    # MI355X_MICROARCH.md, "rocprofv3 PMC slots"), of the same configurations on the same synthetic text in a child process ----
    def pmc_traffic(cfgs):
        import csv, glob, shutil, subprocess, tempfile
        if world != 1 or args.no_traffic or args.fasta or c5 or not shutil.which("rocprofv3"):
            return {}
        res = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="gm_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, str(ROOT / "tools" / "sweep_tuning.py"),
                   "--workload", args.workload, "--scale", str(args.scale), "--sampling", str(args.sampling), "--reps", "1", "--cfg"] + [f"{K},{E},1.0" for K, E in cfgs] + ["--", ""]
            try:
                subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, env=dict(os.environ, TMPDIR="/tmp"), cwd=str(ROOT))
                # the search phase of one pass = every search_kernel / expand_kernel dispatch (the split search of round 6 takes several per
                # pass) up to the pass's finalize kernel; the correction pass (ScatterEnv) is not part of it
                rows = []
                for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r["Counter_Name"] == counter or "finalize" in r["Kernel_Name"]:
                            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) if r["Counter_Name"] == counter else 0.0))
                rows.sort()
                vals, cur, seen_fin = [], 0.0, set()
                for did, name, v in rows:
                    if "finalize" in name:
                        if did not in seen_fin:
                            seen_fin.add(did); vals.append(cur); cur = 0.0
                    elif ("search_kernel" in name or "expand_kernel" in name) and "ScatterEnv" not in name:
                        cur += v
                for i, ke in enumerate(cfgs):   # two passes per configuration (a warm-up and one more): the second
                    if 2 * i + 1 < len(vals):
                        res.setdefault(ke, {})[counter] = vals[2 * i + 1] * 1024.0   # the counter is reported in KB
            except Exception as e:
                log(f"roofline.traffic: rocprofv3 --pmc {counter} failed: {e}")
            shutil.rmtree(d, ignore_errors=True)
        return res

    def pmc_traffic_c5():
        """FETCH_SIZE / WRITE_SIZE of the --exclude-pseudo pass over the five files: this program itself as a child process under rocprofv3
        (one warm-up step and one step, no csv pass): the FileSetEnv dispatches of the last step, summed"""
        import csv, glob, shutil, subprocess, tempfile
        if world != 1 or args.no_traffic or not shutil.which("rocprofv3"):
            return {}
        res, nfiles = {}, len(head["slices"])
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="gm_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, str(ROOT / "bench.py"),
                   "--workload", "bacteria5", "--scale", str(args.scale), "--sampling", str(args.sampling), "--K", str(head["K"]), "--E", str(head["E"]),
                   "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-counters", "--no-traffic", "--no-csv", "--no-host-rate"]
            try:
                subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, env=dict(os.environ, TMPDIR="/tmp"), cwd=str(ROOT))
                acc = {}
                for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        if "search_kernel" in r["Kernel_Name"] and "FileSetEnv" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            acc[int(r["Dispatch_Id"])] = acc.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
                vals = [acc[k] for k in sorted(acc)]
                if len(vals) >= 2 * nfiles:
                    res[counter] = sum(vals[-nfiles:]) * 1024.0
            except Exception as e:
                log(f"roofline.traffic (C5): rocprofv3 --pmc {counter} failed: {e}")
            shutil.rmtree(d, ignore_errors=True)
        return res

    traffic = {}
    if c5:
        t0 = time.time()
        tr5 = pmc_traffic_c5()
        if len(tr5) == 2:
            traffic[(head["K"], head["E"])] = tr5
            log(f"roofline.traffic (C5): FETCH_SIZE / WRITE_SIZE passes under rocprofv3 took {time.time() - t0:.0f} s")
    if not c5:
        t0 = time.time()
        traffic = pmc_traffic([(r["K"], r["E"]) for r in [head] + subs])
        if traffic:
            log(f"roofline.traffic: FETCH_SIZE / WRITE_SIZE passes under rocprofv3 took {time.time() - t0:.0f} s")

    cpu = None
    if not args.no_cpu_baseline:
        try:
            if c5 and sa_cpu is None:
                log("cpu_baseline: config C5 locates, the oracle needs the full suffix array (-S 1): skipped")
            else:
                cpu = CpuBaseline(codes, lens, bwt_host, os.cpu_count() or 1, seq_file_id=fid if c5 else None, sa=sa_cpu)
        except Exception as e:
            log("cpu baseline failed:", e)

    def cpu_rec(rec):
        if cpu is None:
            return None
        guess = {0: 100_000_000, 1: 10_000_000}.get(rec["E"], 1_000_000)
        try:
            if c5:
                return cpu.run_c5(rec["K"], rec["E"], rec["slices"])
            return cpu.run(rec["K"], rec["E"], guess)
        except Exception as e:
            log("cpu baseline failed:", e)
            return None

    def wl(rec):
        if c5:
            return f"{desc}, K={rec['K']} E={rec['E']}, both strands, --exclude-pseudo, 16-bit, common infix {rec['infix']}, -S {args.sampling}"
        return f"{desc}, K={rec['K']} E={rec['E']}, both strands, -fs (8-bit), common infix {rec['infix']}"

    def rank_fields(r, rec):
        pr = rec.get("per_rank")
        if pr:
            r["per_rank_search_ms"] = [x[0] for x in pr]          # search kernel per step, per rank
            r["per_rank_comm_wait_ms"] = [x[1] for x in pr]       # host time per step spent waiting for the gather / the copies
            if len(pr[0]) > 2:
                r["per_rank_correction_ms"] = [x[2] for x in pr]  # the step's ONE correction pass per rank (it runs beside the main search, not behind every launch)
            r["shard_imbalance"] = max(x[0] for x in pr) / max(1e-9, float(np.mean([x[0] for x in pr])))
            r["comm"] = rec.get("comm_mode")
            if rec.get("comm_note"):
                r["comm_note"] = rec["comm_note"]

    if c5:
        parallelism = "one GPU" if world == 1 else f"contiguous shares of whole k-mer blocks per FASTA file and rank, index replicated, one gather of the 16-bit shares to rank 0 ({world} ranks)"
    else:
        parallelism = head["plan"].describe()
    result = {
        "metric": "k-mers/sec (whole node) for (k,e)-mappability on 3.1 Gbp index" if not c5 else "k-mers/sec (whole node) for (k,e)-mappability, config C5 (multi-genome index, --exclude-pseudo)",
        "value": head["num_kmers"] * head["steps"] / head["dt"], "unit": "k-mers/s",
        "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"], "ms_per_step": head["dt"] / head["steps"] * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 ranks / u8 counts" if not c5 else "u32 ranks / u16 counts", "data": data,
        "config": {"workload": wl(head), "K": head["K"], "E": head["E"], "text_len": n, "block_bytes": info["block_bytes"],
                   "parallelism": parallelism, "index_build_s": round(t_build, 2), "index_device_gib": round(info["device_bytes"] / 2**30, 2)},
    }
    if args.tune:
        result["config"]["tune"] = args.tune
    if world > 1:
        result["rccl_ranks"] = rccl_ranks   # dist.get_world_size() of the nccl (= RCCL) process group the gathers ran on (0: another backend)
        result["backend"] = args.backend
        result["rank_features"] = features["per_rank"]
        if not c5:
            result["preflight"] = preflight
    if value_host is not None:   # gm_map with the result vector delivered to host memory (PCIe-inclusive); never `value`
        result["value_host"] = {"K": host_cfg[0], "E": host_cfg[1], "pageable": value_host, "pinned": value_host_pinned, "unit": "k-mers/s"}
    if c5:
        result["roofline"] = roofline_c5(head)
        result["csv"] = head["csv"]
    else:
        result["roofline"] = roofline(head)
    rank_fields(result, head)
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_rec(head)
    if subs:
        result["sub"] = []
        for rec in subs:
            r = {"workload": wl(rec), "K": rec["K"], "E": rec["E"], "value": rec["num_kmers"] * rec["steps"] / rec["dt"], "unit": "k-mers/s",
                 "steps": rec["steps"], "warmup": rec["warmup"], "ms_per_step": rec["dt"] / rec["steps"] * 1e3, "roofline": roofline(rec)}
            rank_fields(r, rec)
            if not args.no_cpu_baseline:
                r["cpu_baseline"] = cpu_rec(rec)
            result["sub"].append(r)
    # ---- the other BASELINE configurations, each on an index of its own (a child process of this program after the main index is gone):
    # C2 = chr1-like 249 Mbp at K=30 e=0, C5 = five bacteria, K=24 e=1 --exclude-pseudo.  Three timed steps, roofline from the twin's counters and
    # a short cpu_baseline each; no counter-traffic passes (they are the minutes of this program). ----
    if world == 1 and args.workload == "grch38" and not args.fasta and args.scale == 1.0 and args.extra_configs:
        import subprocess
        cpu = None   # (the oracle index of the 3.09 Gbp text: tens of GB of host memory)
        torch.cuda.empty_cache()
        for wlname in [w for w in args.extra_configs.split(",") if w]:
            cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", wlname, "--steps", "3", "--warmup", "1", "--sub", "", "--no-traffic", "--no-host-rate",
                   "--cpu-seconds", "3", "--extra-configs", ""] + (["--K", "30", "--E", "0"] if wlname == "chr1" else ["--no-csv"])
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            t0 = time.time()
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, cwd=str(ROOT))
                rec = json.loads(p.stdout.decode().strip().splitlines()[-1])
                r = {"workload": rec["config"]["workload"], "K": rec["config"]["K"], "E": rec["config"]["E"], "value": rec["value"], "unit": rec["unit"], "steps": rec["steps"],
                     "warmup": rec["warmup"], "ms_per_step": rec["ms_per_step"], "roofline": rec.get("roofline"), "cpu_baseline": rec.get("cpu_baseline"),
                     "config": rec["config"], "baseline_config": "C2" if wlname == "chr1" else "C5", "own_index": True}
                result.setdefault("sub", []).append(r)
                log(f"extra config {wlname}: {rec['value']:.4g} {rec['unit']}, {rec['ms_per_step']:.2f} ms/step ({time.time() - t0:.0f} s)")
            except Exception as e:
                log(f"extra config {wlname} failed: {e}")
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

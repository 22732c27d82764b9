#!/usr/bin/env python3
"""bench.py -- (k,e)-mappability throughput on MI355X.

  python bench.py [--gpus N --steps K --warmup W] [--workload chr1|grch38|ecoli] [--scale f] [--K 30 --E 0]

A "step" is one complete computeMappability pass (src/algo.hpp:405-483) over the synthetic genome with the
index already resident in HBM: memset of the accumulators, the search kernel, finalize/resetLimits, and -- for
N > 1 -- the RCCL gather of the ranks' shards of the frequency vector to rank 0.  Each rank holds a full index
replica and computes a disjoint range of k-mer positions (strong scaling: the genome is fixed).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
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


def cpu_baseline(codes, lens, bwt, K, E, threads):
    """The oracle (a port of the reference algorithm, kind="port") on this box's host cores, on a bounded
    sample: whole k-mer blocks from the start of the text, sized so the run takes roughly 10-30 s."""
    sys.path.insert(0, str(ROOT / "tests"))
    import helpers as H
    t0 = time.time()
    ora = H.OracleIndex(codes, lens, keep_sa=False, bwt=bwt)
    log(f"cpu_baseline: oracle adopted the GPU-built BWTs in {time.time() - t0:.1f} s")
    n = len(codes)
    skip = min(n // 10, 20_000)               # stay clear of the leading N block
    avail = n - skip - K
    sample, dt = min(1_000_000, avail), 0.0
    while True:                                # grow the sample until the timed run takes >= 10 s (or covers the text)
        t0 = time.time()
        ora.mappability(K, E, value_bits=8, threads=threads, intervals=[(skip, skip + sample)])
        dt = time.time() - t0
        if dt >= 10.0 or sample >= avail:
            break
        sample = int(min(avail, max(sample * 2, sample * 14.0 / max(dt, 1e-3))))
    return {"value": sample / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": f"{sample} consecutive k-mer positions from offset {skip} of the same index, K={K} E={E}, both strands, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="chr1")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--K", type=int, default=30)
    ap.add_argument("--E", type=int, default=0)
    ap.add_argument("--block-bytes", type=int, default=0)
    ap.add_argument("--infix", type=int, default=0, help="common-infix length (SearchParams.overlap); 0 = library default")
    ap.add_argument("--sampling", type=int, default=1, help="1: keep the suffix array resident (narrow nodes are verified against the text); 0: rank queries only")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the single-GPU rehearsal of the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
    args = ap.parse_args()

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

    t0 = time.time()
    codes, lens, desc = synth.workload(args.workload, args.scale)
    n = int(len(codes))
    log(f"workload {desc}: generated in {time.time() - t0:.1f} s")
    K, E = args.K, args.E
    t0 = time.time()
    ix = g.Index.build(codes, lens, sampling=args.sampling, block_bytes=args.block_bytes, device=local_rank)
    t_build = time.time() - t0
    info = ix.info()
    log(f"index built on the GPU in {t_build:.1f} s: {info['n_rows']} rows, {info['block_bytes']}-B blocks, {info['device_bytes'] / 2**30:.2f} GiB")

    infix = args.infix or g.tuned_infix_length(K, E)
    step_sz = K - infix + 1
    num_kmers = n - K + 1
    from genmap_amd.distributed import gather_frequency, max_shard_len, shard_ranges
    ranges = shard_ranges(num_kmers, step_sz, world)   # contiguous shards of whole k-mer blocks
    kb, ke = ranges[rank]
    max_shard = max_shard_len(ranges)

    out = torch.zeros(n + max_shard, dtype=torch.uint8, device=dev)        # -fs: 8-bit frequencies
    gathered = [torch.empty(max_shard, dtype=torch.uint8, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream
    search_ms = []

    def one_step():
        ix.map_device(out.data_ptr(), K, E, infix=args.infix, value_bits=8, kmer_range=(kb, ke) if world > 1 else None, stream=stream)
        if world > 1:
            gather_frequency(out, ranges, rank, world, dist, recv_bufs=gathered, stage_on_host=(args.backend != "nccl"))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        search_ms.append(None)
    sync()
    dt = time.perf_counter() - t0
    st = ix.last_stats()   # HIP events of the last step on the launch stream
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        if args.backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # per-launch kernel time over a few extra (untimed-by-wall) launches, each measured with HIP events
    kms = []
    for _ in range(min(args.steps, 5)):
        ix.map_device(out.data_ptr(), K, E, infix=args.infix, value_bits=8, kmer_range=(kb, ke) if world > 1 else None, stream=stream)
        kms.append(ix.last_stats()["search_ms"])
    kernel_ms = float(np.mean(kms))

    result = None
    if rank == 0:
        value = num_kmers * args.steps / dt
        result = {
            "metric": "k-mers/sec (whole node) for (k,e)-mappability", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 ranks / u8 counts",
            "data": "synthetic",
            "config": {"workload": f"{desc}, K={K} E={E}, both strands, -fs (8-bit), common infix {infix}", "K": K, "E": E,
                       "text_len": n, "block_bytes": info["block_bytes"], "parallelism": f"text-range shards x{world}, index replicated",
                       "index_build_s": round(t_build, 2)},
        }
    # ---- roofline numerator: count node steps / distinct rank lines with the instrumented twin (untimed) ----
    lines = steps_cnt = None
    roots = 0
    v_items = v_chunks = 0
    if rank == 0 and world == 1 and not args.no_counters and g.lib_path(True).exists():
        try:
            bf, br = ix.export_bwt()
            ixp = g.Index.from_bwt(bf, br, codes, lens, sa_fwd=(ix.export_sa() if args.sampling == 1 else None), sampling=args.sampling,
                                   block_bytes=info["block_bytes"], device=local_rank, profiling=True)
            tmp = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
            ixp.map_device(tmp.data_ptr(), K, E, infix=args.infix, value_bits=8, stream=stream)
            sp = ixp.last_stats()
            lines, steps_cnt, roots = sp["rank_lines"], sp["node_steps"], sp["roots"]
            v_items, v_chunks = sp["detail"]["verify_items"], sp["detail"]["verify_chunks"]
            ixp.close()
            del tmp
        except Exception as e:  # measurement aid only
            log("counter pass failed:", e)
    if rank == 0:
        bb = info["block_bytes"]
        if lines:
            # rank blocks + one q-mer table entry per root + text read once per strand (4-bit packed) + 8-bit output
            # + verification (SA entry per row, 8 needle + 8 text symbols per chunk)
            alg = bb * lines + 16 * roots + n + n + 4 * v_items + 16 * v_chunks
            ach = alg / (kernel_ms * 1e-3) / 1e9
            traffic = None
            tf = ROOT / "profiles" / "pmc_traffic.json"
            if tf.exists():
                try:
                    rec = json.loads(tf.read_text())
                    if rec.get("workload") == result["config"]["workload"]:
                        traffic = rec.get("hbm_bytes_per_launch")
                except Exception:
                    pass
            result["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                  "traffic": traffic, "kernel": "search_kernel", "kernel_ms": kernel_ms,
                                  "algorithmic_bytes": alg, "rank_lines": lines, "roots": roots, "node_steps": steps_cnt,
                                  "node_steps_per_kmer": steps_cnt / num_kmers,
                                  "verify_items": v_items, "verify_chunks": v_chunks}
        else:
            result["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                  "kernel": "search_kernel", "kernel_ms": kernel_ms}
        if world == 1 and not args.no_cpu_baseline:
            try:
                bwt = ix.export_bwt()
                result["cpu_baseline"] = cpu_baseline(codes, lens, bwt, K, E, os.cpu_count() or 1)
            except Exception as e:
                log("cpu baseline failed:", e)
                result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    ix.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Group the dispatches of a rocprofv3 csv (kernel trace or counter collection) of tools/sweep_tuning.py into PASSES: one gm_map_device call =
every expand_kernel / search_kernel dispatch up to the call's finalize kernel (the split search of round 6 takes many dispatches per pass).
Usage: passes.py trace <kernel_trace.csv> <cfg> <cfg> ... [--reps R]     -> per configuration: ms of phase A, of the walker / one-loop kernel, slices
       passes.py pmc <counter_collection.csv> <COUNTER> <cfg> ...        -> per configuration: the counter summed over the pass's search + expand dispatches"""
import csv, sys, collections


def load(path):
    return list(csv.DictReader(open(path)))


def passes(rows, key, val):
    """rows sorted by `key`; returns a list of passes, each {kernel class: [values]}"""
    rows = sorted(rows, key=key)
    out, cur = [], collections.defaultdict(list)
    seen_fin = set()
    for r in rows:
        n = r["Kernel_Name"]
        if "finalize" in n:
            k = key(r)
            if k not in seen_fin:
                seen_fin.add(k)
                out.append(cur); cur = collections.defaultdict(list)
        elif "expand_kernel" in n:
            cur["phase_a"].append(val(r))
        elif "search_kernel" in n and "ScatterEnv" in n:
            cur["correction"].append(val(r))
        elif "search_kernel" in n:
            cur["walker" if "CountEnv<1, 2>" in n or "CountEnv<3, 2>" in n else "one_loop"].append(val(r))
    return out


def main():
    mode, path = sys.argv[1], sys.argv[2]
    rows = load(path)
    if mode == "trace":
        args = sys.argv[3:]
        reps = 2
        if "--reps" in args:
            i = args.index("--reps"); reps = int(args[i + 1]); args = args[:i] + args[i + 2:]
        ps = passes(rows, lambda r: int(r["Start_Timestamp"]), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        per = reps + 1
        print(f"# rocprofv3 --kernel-trace of tools/sweep_tuning.py --reps {reps} ({per} passes per configuration, the first an untimed warm-up); ms summed over the dispatches of a pass")
        for k, c in enumerate(args):
            for j, p in enumerate(ps[per * k:per * k + per]):
                a, w, o = sum(p["phase_a"]), sum(p["walker"]), sum(p["one_loop"])
                nz = [x for x in p["walker"] if x > 0.2]
                print(f"cfg {c} pass {j}: phase A (expand_kernel) {a:9.3f} ms in {len(p['phase_a'])} dispatches | walker {w:9.3f} ms in {len(p['walker'])} dispatches ({len(nz)} with packets) | "
                      f"one-loop search_kernel {o:9.3f} ms in {len(p['one_loop'])} | correction pass {sum(p['correction']):8.3f} ms | search phase {a + w + o:9.3f} ms")
    else:
        counter, args = sys.argv[3], sys.argv[4:]
        rows = [r for r in rows if r["Counter_Name"] == counter or "finalize" in r["Kernel_Name"]]
        ps = passes(rows, lambda r: int(r["Dispatch_Id"]), lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == counter else 0.0)
        for k, c in enumerate(args):   # two passes per configuration (--reps 1): the second
            if 2 * k + 1 < len(ps):
                p = ps[2 * k + 1]
                a, w, o = sum(p["phase_a"]), sum(p["walker"]), sum(p["one_loop"])
                print(f"{counter} cfg {c}: phase A {a:.6g} | walker {w:.6g} | one-loop {o:.6g} | search phase {a + w + o:.6g}")


if __name__ == "__main__":
    main()

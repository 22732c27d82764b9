#!/usr/bin/env python3
"""Device-side step statistics of one computeMappability call (instrumented twin library; never timed)."""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import genmap_amd as g
from genmap_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="chr1"); ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--cfg", nargs="+", default=["30,0", "30,1", "30,2", "100,1"])
ap.add_argument("--frac", type=float, default=1.0, help="share of the k-mers (a range from the middle of the text)")
ap.add_argument("--settings", nargs="*", default=[""], help="knob settings 'a=1,b=2' to compare (gm_index_set_tuning)")
a = ap.parse_args()
codes, lens, desc = synth.workload(a.workload, a.scale)
ix = g.Index.build(codes, lens, sampling=1, profiling=True)
out = torch.zeros(len(codes) + 16, dtype=torch.uint8, device="cuda:0")
DEFAULTS = dict(verify_t=-1, lds_stack=-1, blocks_per_cu=4, qtable=-1, sat_min_w=256, fetch_batch=-1, probation=-1, verify_cost=3, skip_dup=-1, coop=-1, use_ctx=1, steal=-1, part_bias=0, oss_weights=-1, jump=-1, self_hit=1, jump_filter=1, range_add=1, verify_t_ext=-1, jump_groups=-1, expand=-1, expand_mb=-1, expand_chunk=-1, sat_draw_w=-1, expand_occ=-1, expand_overlap=-1, expand_two_pass=-1, expand_share=-1, win2=-1)
for cfg, setting in [(c, s2) for c in a.cfg for s2 in a.settings]:
    K, E = map(int, cfg.split(","))
    knobs = dict(DEFAULTS)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("="); knobs[k] = int(v)
    ix.set_tuning(**knobs)
    nk0 = len(codes) - K + 1
    step = K - g.tuned_infix_length(K, E) + 1
    span = int(nk0 * a.frac) // step * step
    kb = ((nk0 - span) // 2) // step * step
    ix.map_device(out.data_ptr(), K, E, value_bits=8, kmer_range=None if a.frac >= 1.0 else (kb, kb + span))
    st = ix.last_stats()
    nk = st["kmers"]
    d = st["detail"]
    print(json.dumps({"workload": desc, "K": K, "E": E, "settings": setting, "kmers": nk, "jump_lookups_per_kmer": d.get("jump_lookups", 0) / nk, "jump_words_per_kmer": d.get("jump_words", 0) / nk, "jump_filtered_per_kmer": d.get("jump_filtered", 0) / nk, "packets_per_kmer": d.get("packets", 0) / nk, "slices": d.get("slices", 0), "correction_us": d.get("correction_us", 0), "steps_per_kmer": st["node_steps"] / nk, "lines_per_kmer": st["rank_lines"] / nk,
                      "oss_frac": d["steps_oss"] / st["node_steps"], "ext_w1_frac": d["ext_w1"] / st["node_steps"], "ext_w2_4_frac": d["ext_w2_4"] / st["node_steps"],
                      "oss_w1_frac": d["oss_w1"] / st["node_steps"], "pushes_per_step": d["pushes"] / st["node_steps"], "verify_items_per_kmer": d["verify_items"] / nk, "verify_items_oss_frac": d["verify_items_oss"] / max(1, d["verify_items"]), "chunks_per_item": d["verify_chunks"] / max(1, d["verify_items"]), "lanes_active_per_iteration": d["active_lane_sum"] / max(1, d["wave_iterations"]), "wave_iterations": d["wave_iterations"], "cyc_fetch_frac": d["cyc_fetch"] / max(1, d["cyc_fetch"] + d["cyc_verify"] + d["cyc_step"]), "cyc_verify_frac": d["cyc_verify"] / max(1, d["cyc_fetch"] + d["cyc_verify"] + d["cyc_step"]), "cyc_step_frac": d["cyc_step"] / max(1, d["cyc_fetch"] + d["cyc_verify"] + d["cyc_step"]), "cyc_per_iter": (d["cyc_fetch"] + d["cyc_verify"] + d["cyc_step"]) / max(1, d["wave_iterations"]), "verify_rounds": d["verify_rounds"], "fetch_parts_pop_share_st32_st1": [round(d[k] / max(1, d["cyc_fetch"] + d["cyc_verify"] + d["cyc_step"]), 3) for k in ("cyc_pop", "cyc_share", "cyc_stage32", "cyc_stage1")], "stolen_per_kmer": d["stolen"] / nk, "search_ms_instrumented": st["search_ms"],
                      # how often a wavefront executes a region, per loop iteration (x the region's instruction count = its share of the work)
                      "wave_passes_per_iteration": {k[2:]: round(d[k] / max(1, d["wave_iterations"]), 4) for k in d if k.startswith("w_")}}))

#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/profile_round.sh) -> profiles/<tag>_*: the bench line, the rocprofv3 kernel
statistics of the same command and the gather's HBM traffic from the PMC passes.
usage: python tools/collect_profiles.py r02"""
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, f"{tag}_bench_dp1.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_dp1_kernel_stats.csv"))
shutil.copy(os.path.join(src, "gather_trace_groups.csv"), os.path.join(dst, f"{tag}_bench_dp1_gather_groups.csv"))


def pmc(counter):
    path = os.path.join(src, f"pmc_{counter}.txt")
    if not os.path.exists(path):
        return None
    best = None
    for ln in open(path):
        m = re.search(r"n=\s*(\d+) avg \S+=\s*([\d.]+)", ln)
        if m and "true" in ln.split("(")[0] + ln:       # staged variant: emb_fwd_pair<..., true>
            if best is None or int(m.group(1)) > best[0]:
                best = (int(m.group(1)), float(m.group(2)))
    return best


r128, r64, r32 = pmc("TCC_EA0_RDREQ_128B_sum"), pmc("TCC_EA0_RDREQ_64B_sum"), pmc("TCC_EA0_RDREQ_32B_sum")
w64, wall = pmc("TCC_EA0_WRREQ_64B_sum"), pmc("TCC_EA0_WRREQ_sum")
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "bench_kernel_stats.csv")))}
gk = [k for k in stats if k.startswith("void emb_fwd_pair") and "true" in k]
out = {"kernel": gk[0] if gk else None,
       "command": "tools/profile_round.sh: rocprofv3 --pmc <one counter> --kernel-trace -- python3 bench.py --steps 40 --warmup 8 "
                  "--no-cpu-baseline --no-extra-configs (one pass per counter)"}
if r128 and w64:
    fetch = r128[1] * 128 + (r64[1] if r64 else 0) * 64 + (r32[1] if r32 else 0) * 32
    wr = w64[1] * 64 + ((wall[1] - w64[1]) * 32 if wall else 0)
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    out.update({
        "launches": r128[0],
        "TCC_EA0_RDREQ_128B_avg": r128[1], "TCC_EA0_RDREQ_64B_avg": r64[1] if r64 else None,
        "TCC_EA0_RDREQ_32B_avg": r32[1] if r32 else None, "TCC_EA0_WRREQ_64B_avg": w64[1],
        "TCC_EA0_WRREQ_avg": wall[1] if wall else None,
        "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": fetch + wr,
        "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": (fetch + wr) / algo,
        "note": "bytes from the L2's memory-side request counters by request size (128-B / 64-B / 32-B reads, 64-B / 32-B "
                "writes) - no correction factor needed.  Every 64-B row of a table is fetched as ONE 128-B request "
                "(tools/microbench_gather2: the same with sc0 / sc1 / nt loads and with uncached or fine-grained table "
                "memory), so reads are (128 + 8)/(64 + 4 + 8) = 1.8x the algorithmic bytes by construction of the cache "
                "hierarchy; writes are the algorithmic 10.24 MB + S = sum_f e (0.26 MB) + the staged batch (1.08 MB)."})
if gk:
    out["kernel_trace_avg_us"] = float(stats[gk[0]]["AverageNs"]) / 1e3
    out["kernel_trace_min_us"] = float(stats[gk[0]]["MinNs"]) / 1e3
    out["kernel_trace_calls"] = int(stats[gk[0]]["Calls"])
    out["frac_of_8TBps_at_kernel_trace_avg"] = bench["roofline"]["algorithmic_bytes_per_launch"] / (out["kernel_trace_avg_us"] * 1e-6) / 8e12
json.dump(out, open(os.path.join(dst, f"{tag}_gather_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))


# ---- tools/profile_extras.sh: layers, the field-sharded step, the simulated rank, the replicated tail
def last_json(path):
    lines = [ln for ln in open(path) if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


for name, to in (("layer_cin_kernel_stats.csv", "cin_cfg3_kernel_stats.csv"), ("layer_attn_kernel_stats.csv", "attention_cfg4_kernel_stats.csv"),
                 ("sharded_dp1_kernel_stats.csv", "sharded_dp1_kernel_stats.csv"), ("sharded_sim1.txt", "sharded_sim_rank0_of_1.txt"),
                 ("sharded_sim8.txt", "sharded_sim_rank0_of_8.txt"), ("replicated_tail.txt", "replicated_tail.txt"),
                 ("layer_cin.txt", "cin_cfg3_layer_time.txt"), ("layer_attn.txt", "attention_cfg4_layer_time.txt")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{to}"))
dp = {}
for name, key in (("sharded_bench_untraced.json", "field_sharded_one_rank_rccl"), ("replicated_bench_untraced.json", "replicated_one_rank_rccl")):
    path = os.path.join(src, name)
    if os.path.exists(path):
        b = last_json(path)
        if b:
            dp[key] = {k: b[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")} | {"parallelism": b["config"]["parallelism"],
                                                                                       "steps_per_graph": b["config"]["steps_per_graph"]}
if dp:
    dp["plain_single_gpu"] = {"value": bench["value"], "ms_per_step": bench["ms_per_step"]}
    dp["command"] = "DFM_FORCE_DP_PATH=1 python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs [--dp-mode replicated]"
    json.dump(dp, open(os.path.join(dst, f"{tag}_dp_structure_one_gpu.json"), "w"), indent=1)
    print(json.dumps(dp, indent=1))

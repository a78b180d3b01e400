#!/usr/bin/env python3
"""Copies a collection made by tools/collect_profiles.sh (gpurun_out/final_<tag>/) into profiles/ under the round's names:
   python tools/publish_profiles.py <tag> <round, e.g. r03>
Bench lines, kernel-stats CSVs and PMC summaries are copied as they are; hw_queues keeps its comment header; traffic.json gets
the k_accum_affine figures of the FETCH_SIZE / WRITE_SIZE passes (gfx950: bytes = 2 x FETCH_SIZE KB + WRITE_SIZE KB)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"final_{tag}")
dst = os.path.join(ROOT, "profiles")


def cp(a, b):
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}"))
    print("profiles/%s_%s" % (rnd, b))


cp("bench.json", "bench.json")
cp("bench_u64.json", "bench_u64.json")
cp("bench_sharded_world1.json", "bench_sharded_world1_rccl.json")
cp("bench_under_rocprof.json", "bench_under_rocprof.json")
cp("bench_paths_under_rocprof.json", "bench_paths_under_rocprof.json")
cp("stats/s_kernel_stats.csv", "bench_kernel_stats.csv")
cp("stats_paths/s_kernel_stats.csv", "paths_kernel_stats.csv")
cp("pmc_fetch.summary.json", "pmc_fetch.json")
cp("pmc_write.summary.json", "pmc_write.json")
cp("pmc_sq.summary.json", "pmc_sq_single_msm.json")
cp("sweep.jsonl", "sweep.jsonl")
for a, b in (("stress_callers.txt", "stress_callers.txt"), ("ntt_probe.txt", "ntt_probe.txt"), ("bench_sharded_block_world1.json", "bench_sharded_block_world1.json"),
             ("pmc_ntt_sq.summary.json", "pmc_ntt_2e20_sq.json"), ("pmc_ntt_lds.summary.json", "pmc_ntt_2e20_lds.json"),
             ("pmc_ntt_fetch.summary.json", "pmc_ntt_2e20_fetch.json"), ("pmc_ntt_write.summary.json", "pmc_ntt_2e20_write.json"),
             ("skew.txt", "skewed_scalars.txt")):
    if os.path.exists(os.path.join(src, a)):
        cp(a, b)
# the fuzz record is a history of runs (seeds, durations, builds): the collection's run is appended, never replaces it
fz = os.path.join(src, "fuzz.txt")
if os.path.exists(fz):
    last = [ln for ln in open(fz).read().splitlines() if "fuzz done" in ln][-1:]
    with open(os.path.join(dst, f"{rnd}_fuzz.txt"), "a") as f:
        f.write("# collection %s:\n%s\n" % (tag, "\n".join(last)))
    print("profiles/%s_fuzz.txt (appended)" % rnd)
hw = os.path.join(dst, f"{rnd}_hw_queues.txt")
head = [ln for ln in open(hw).read().splitlines() if ln.startswith("#")] if os.path.exists(hw) else []
open(hw, "w").write("\n".join(head + open(os.path.join(src, "hw_queues.txt")).read().splitlines()) + "\n")
print("profiles/%s_hw_queues.txt" % rnd)


def kernel_avg(summary, name, counter):
    d = json.load(open(os.path.join(src, summary)))
    v = d[name][counter]
    return v["avg"], v["launches"]


tj = os.path.join(dst, "traffic.json")
t = json.load(open(tj))
f_kb, n = kernel_avg("pmc_fetch.summary.json", "k_accum_affine", "FETCH_SIZE")
w_kb, _ = kernel_avg("pmc_write.summary.json", "k_accum_affine", "WRITE_SIZE")
t["raw_fetch_kb"], t["raw_write_kb"] = round(f_kb, 1), round(w_kb, 1)
t["k_accum_affine_bytes_per_launch"] = int(round((2 * f_kb + w_kb) * 1024))
t["ratio_to_algorithmic"] = round(t["k_accum_affine_bytes_per_launch"] / t["algorithmic_bytes_per_launch"], 1)
json.dump(t, open(tj, "w"), indent=1)
print("profiles/traffic.json", t["k_accum_affine_bytes_per_launch"], t["ratio_to_algorithmic"])

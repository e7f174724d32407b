"""gpurun_out/<round> (tools/gpu_evidence.sh) -> the small files committed under profiles/.

  python tools/summarize_round.py r03
"""
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_prof as SP  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", ROUND)
DST = os.path.join(ROOT, "profiles")


def first(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return g[0] if g else None


for w in ("c1", "c2", "c3", "c4", "c5", "k2f"):
    b = os.path.join(SRC, f"bench_{w}.json")
    if os.path.exists(b) and os.path.getsize(b) > 0:
        shutil.copy(b, os.path.join(DST, f"{ROUND}_{w}_bench.json"))
    b = os.path.join(SRC, f"bench_{w}_cpu.json")
    if os.path.exists(b) and os.path.getsize(b) > 0:
        shutil.copy(b, os.path.join(DST, f"{ROUND}_{w}_bench_with_cpu_baseline.json"))
for w in ("c2", "c3", "c4", "c5", "k2f"):
    st = first(f"prof_{w}/**/{w}_kernel_stats.csv")
    if st:
        SP.stats(st, os.path.join(DST, f"{ROUND}_{w}_kernel_stats.csv"))
    pb = os.path.join(SRC, f"prof_{w}_bench.json")
    if os.path.exists(pb) and os.path.getsize(pb) > 0:
        shutil.copy(pb, os.path.join(DST, f"{ROUND}_{w}_bench_under_rocprof.json"))
    f, wr = first(f"pmc_fetch_{w}/**/{w}_counter_collection.csv"), first(f"pmc_write_{w}/**/{w}_counter_collection.csv")
    bench = os.path.join(SRC, f"bench_{w}.json")
    if f and wr and os.path.exists(bench):
        d = json.load(open(bench))
        alg = d["roofline"]["algorithmic_bytes_per_row"] * d["config"]["rows_per_gpu"]
        SP.pmc(f, wr, os.path.join(DST, f"pmc_{w}.json"), alg, 1)
        shutil.copy(os.path.join(DST, f"pmc_{w}.json"), os.path.join(DST, f"{ROUND}_pmc_{w}.json"))
# SQ counters of the C5 kernel, per wave
acc = {}
for path in glob.glob(os.path.join(SRC, "sq_c5_*", "**", "c5_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if acc:
    with open(os.path.join(DST, ROUND + "_c5_sq_counters.txt"), "w") as f:
        waves = sum(acc.get("SQ_WAVES", [0])) or 1
        f.write("# C5 kernels (pre-pass + main), SQ counters summed over the profiled launches; per wave = / SQ_WAVES\n")
        f.write("# (main kernel: one wave = one wave tile = 512 rows = 8 sub-tiles of 64 rows; the pre-pass walks several tiles per wave; launches = every Evaluate of the profiled command, pre-warm steps included)\n")
        for k, v in sorted(acc.items()):
            f.write(f"{k:22s} total {sum(v):.4g}  launches {len(v)}  per_wave {sum(v) / waves * (len(acc.get('SQ_WAVES', [1])) / len(v)):.1f}\n")
for name in ("make_latency.txt", "micro_benchmarks.txt", "latency_sweep.txt", "smoke.log", "hbm_ceiling.txt",
             "small_batches.txt", "registry_tail_timing.txt", "flat_only_plans.txt", "inproc_bench.txt", "multi_device.txt",
             "c5_variants.txt", "filter_project_chain.txt", "filter_project.txt", "c5_nonascii.txt", "c4_repeat.txt",
             "bench_two_ranks.txt", "filter_project_traffic.txt", "filter_project_shapes.txt", "filter_string_chain.txt",
             "pytest_gpu_tier0.txt", "bench_c2_time.txt", "kcache_size.txt"):
    p = os.path.join(SRC, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, ROUND + "_" + name))
st = first("prof_fp/**/fp_kernel_stats.csv")
if st:
    SP.stats(st, os.path.join(DST, f"{ROUND}_filter_project_kernel_stats.csv"))
p = os.path.join(SRC, "pytest_gpu_full.log")
if os.path.exists(p):
    lines = [l for l in open(p) if "passed" in l or "failed" in l or l.startswith(("FAILED", "ERROR"))]
    open(os.path.join(DST, ROUND + "_pytest_gpu_summary.txt"), "w").writelines(lines[-10:])

# ---- one table for DESIGN.md §4: bench line + rocprofv3 kernel stats + PMC traffic per workload
def kernel_rows(path):
    out = []
    if path and os.path.exists(path):
        for r in csv.DictReader(open(path)):
            n = r["Name"].strip('"')
            # the generated kernels + the ahead-of-time scan / emission kernels (not the ceiling instrument's streams)
            short = n if n.startswith("gdv_k_") else next((k for k in ("EmitIndices", "ScanReduce", "ScanSpine", "ScanApply") if k in n), None)
            if short:
                out.append(("same" if False else short[:24], int(r["Calls"]), float(r["AverageNs"]) / 1e6))
    return out


with open(os.path.join(DST, ROUND + "_summary.md"), "w") as f:
    f.write(f"<!-- written by tools/summarize_round.py {ROUND} from gpurun_out/{ROUND} (one run of tools/gpu_evidence.sh) -->\n")
    f.write("| workload | ms / Evaluate (bench) | kernel_ms (HIP events) | rocprofv3 averages of the same command | rows/s | achieved | frac of 8 TB/s | frac of measured ceiling | PMC traffic / algorithmic | output placement (device pool candidates; frac on the first plain allocation) |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for w in ("c2", "c3", "c4", "c5", "k2f", "c1"):
        b = os.path.join(DST, f"{ROUND}_{w}_bench.json")
        if not os.path.exists(b):
            continue
        try:
            d = json.loads([l for l in open(b) if l.startswith("{")][-1])
        except Exception:
            continue
        r = d["roofline"]
        ks = kernel_rows(os.path.join(DST, f"{ROUND}_{w}_kernel_stats.csv"))
        ks = ", ".join(f"{'the same kernel' if n == r['kernel_name'] else n} {a:.3f} ms x{c}" for n, c, a in ks if a >= 0.05) or "-"
        pm = os.path.join(DST, f"{ROUND}_pmc_{w}.json")
        tr = "-"
        if os.path.exists(pm):
            pj = json.load(open(pm))
            tr = f"{pj['traffic_over_algorithmic']:.4f} x" + ("" if pj["kernel"] == r["kernel_name"] else f" (on {pj['kernel'][:22]}: NOT the timed kernel)")
        f.write(f"| {w.upper()} | {d['ms_per_step']} | {r['kernel_ms']} on {r['kernel_name']} | {ks} | "
                f"{d['value'] / 1e3:.1f} G | {r['achieved'] / 1e3:.2f} TB/s | {r['frac']} | {r.get('frac_of_measured_ceiling', '-')} | {tr} | {(('pool: ' + str(r['placement']['sets'][0]['rates_gbs']) + ' GB/s, first plain allocation ' + str(r.get('frac_first_plain_allocation'))) if r.get('placement') else None) or r.get('placement_trials_ms') or '-'} |\n")
    # the driver's own command (python bench.py): the sub-lines of the ONE bench line
    b = os.path.join(DST, f"{ROUND}_c2_bench.json")
    try:
        d = json.loads([l for l in open(b) if l.startswith("{")][-1])
    except Exception:
        d = {}
    if d.get("workloads"):
        f.write("\n<!-- the sub-lines of the same run's ONE bench line (python bench.py): \"workloads\" -->\n")
        f.write("| in the driver's line | ms_per_step | kernel_ms (HIP events) | frac of 8 TB/s | PMC traffic / algorithmic bytes | verified | cpu_baseline (port) | wall s |\n|---|---|---|---|---|---|---|---|\n")
        f.write(f"| **c2 (headline)** | {d['ms_per_step']} | {d['roofline']['kernel_ms']} | {d['roofline']['frac']} | "
                f"{(d['roofline']['traffic'] or 0) / (d['roofline']['algorithmic_bytes_per_row'] * d['config']['rows_per_gpu']):.4f} | {d['verified']} | "
                f"{d.get('cpu_baseline', {}).get('value')} M rows/s, {d.get('cpu_baseline', {}).get('cores')} thr | - |\n")
        for k, v in d["workloads"].items():
            if "roofline" not in v:
                f.write(f"| {k} | failed: {v.get('error')} | | | | False | | |\n")
                continue
            r = v["roofline"]
            tr = r["traffic"] / (r["algorithmic_bytes_per_row"] * v["config"]["rows"]) if r.get("traffic") else None
            cb = v.get("cpu_baseline", {})
            f.write(f"| {k} | {v['ms_per_step']} | {v['kernel_ms']} | {r['frac']} | {('%.4f' % tr) if tr else '-'} | {v['verified']} | "
                    f"{cb.get('value')} M rows/s, {cb.get('cores')} thr | {v.get('wall_s')} |\n")
print(sorted(os.listdir(DST)))

"""Summarises rocprofv3 --pmc passes over tools/box_states.py (round 5, verdict item 6): every call of time_round is
17 launches of the C2 kernel on one set of buffers; per group of 17 consecutive gdv_k_ dispatches: mean kernel
duration (the trace's own timestamps) and the mean of every collected counter.

  python tools/box_states_pmc.py <dir with *_counter_collection.csv> ...
"""
import csv, glob, os, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        disp = {}
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith("gdv_k_"):
                continue
            e = disp.setdefault(int(r["Dispatch_Id"]), {"t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(disp)
        names = sorted({n for i in ids for n in disp[i]["c"]})
        print(f"# {f}: {len(ids)} C2 dispatches; counters: {', '.join(names)}")
        for g in range(0, len(ids) - 16, 17):
            grp = [disp[i] for i in ids[g + 5:g + 17]]     # the 12 timed launches of the group
            ms = sum(x["t"] for x in grp) / len(grp)
            cs = "  ".join(f"{n} {sum(x['c'].get(n, 0.0) for x in grp) / len(grp):.4g}" for n in names)
            rnd, again = divmod(g // 17, 2)
            print(f"round {rnd}{' again' if again else '      '}: kernel {ms:.3f} ms   {cs}")

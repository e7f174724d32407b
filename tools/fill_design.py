"""Puts profiles/<round>_summary.md (tools/summarize_round.py) between the table markers of DESIGN.md §4.
  python tools/fill_design.py r05"""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
table = open(os.path.join(root, "profiles", rnd + "_summary.md")).read().strip()
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
begin, end = "<!-- evidence table: begin -->", "<!-- evidence table: end -->"
if begin not in s:
    s = s.replace("R05_MEASUREMENT_TABLE", begin + "\n" + end)
s = re.sub(re.escape(begin) + r".*?" + re.escape(end), lambda m: begin + "\n" + table + "\n" + end, s, flags=re.S)
open(p, "w").write(s)
print("DESIGN.md §4 table <-", rnd)

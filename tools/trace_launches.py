#!/usr/bin/env python3
"""Per-launch durations of the kernels of ONE settled bench search, in launch order, from a rocprofv3 kernel trace
(csv): python tools/trace_launches.py <dir with *kernel_trace.csv> [searches in the trace, default 2]
Prints the last search's launches: kernel, duration in ms (development tool)."""
import csv, glob, os, re, sys
root = sys.argv[1]
nsearch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
    if not m:
        continue
    rows.append((int(r["Start_Timestamp"]), m.group(1) + (m.group(2) or ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
# a search starts with two fills back to back (bootstrap); take the launches after the last k_node_cache* launch
last_build = max([i for i, r in enumerate(rows) if r[1].startswith("k_node_cache")] + [-1])
rows = rows[last_build + 1:]
fills = [i for i, r in enumerate(rows) if r[1].startswith(("k_assemble_dense", "k_assemble_union", "k_assemble_cached"))]
per = len(fills) // max(1, nsearch - 1) if nsearch > 1 else len(fills)
start = fills[-per] if per and len(fills) >= per else 0
tot = {}
k = 0
for t, name, ms in rows[start:]:
    tot[name] = tot.get(name, 0.0) + ms
    if ms >= 0.05:
        print(f"{name:42s} {ms:8.3f} ms")
print("totals of the last search:", {a: round(b, 2) for a, b in sorted(tot.items(), key=lambda x: -x[1])})

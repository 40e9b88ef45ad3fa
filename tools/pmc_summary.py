#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel and derive the executed-work figures bench.py's
`roofline` object uses.
  python tools/pmc_summary.py <dir with pass*/.../*_counter_collection.csv> <out.json>

Derived per kernel (all per launch = counter sum / launches seen in that pass):
  fp64_flop_issued_per_launch = 64 lanes x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) + 512 x MFMA_MOPS_F64
                                (wave-level instruction counts: every issued instruction occupies the
                                 FP64 pipe for all 64 lanes whether or not they are active)
  lane_utilisation            = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU)
  fp64_share_of_valu          = (ADD + MUL + FMA + TRANS)_F64 / SQ_INSTS_VALU
  hbm_fetch/write_bytes_per_launch = FETCH_SIZE / WRITE_SIZE (KB) x 1024, RAW (gfx950: wide reads are
                                 tallied at half their bytes, MI355X_MICROARCH.md HBM section)
"""
import csv, glob, json, os, re, sys

def short(name):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]

root, out = sys.argv[1], sys.argv[2]
searches = int(sys.argv[3]) if len(sys.argv) > 3 else 2  # root searches (bench steps) the profiled program ran
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
acc, launches = {}, {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen:
        launches.setdefault(k, {})[f] = launches.setdefault(k, {}).get(f, 0) + 1
import bench
res = {"kernels": {}, "searches": searches, "source_sha16": bench.kernel_source_sha16(),
       "note": "per kernel: counter sums over all launches of `searches` root searches (steps) of the workload; "
               "source_sha16 = sha256 of emme_amd/csrc/*.h* at collection time (bench.py marks the summary stale when it differs)"}
for k, c in acc.items():
    n = max(launches[k].values())
    d = dict(c)
    d["launches"] = n
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_INSTS_VALU"):
        d["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"])
    if "SQ_INSTS_VALU_FMA_F64" in c:
        f64 = sum(c.get("SQ_INSTS_VALU_%s_F64" % t, 0.0) for t in ("ADD", "MUL", "FMA", "TRANS"))
        flop = 64.0 * (f64 + c.get("SQ_INSTS_VALU_FMA_F64", 0.0)) + 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        d["fp64_flop_issued_per_launch"] = flop / n
        d["fp64_mfma_flop_per_launch"] = 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / n
        d["fp64_valu_insts_per_launch"] = f64 / n
        d["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU", 0.0) / n
        if c.get("SQ_INSTS_VALU"):
            d["fp64_share_of_valu"] = f64 / c["SQ_INSTS_VALU"]
    if c.get("SQ_WAVE_CYCLES"):
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES"):
            if nm in c:
                d[nm + "_per_wave_cycle"] = c[nm] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0))
    if "FETCH_SIZE" in c:  # KB (rocprofv3 derived metric), per launch in bytes
        d["hbm_fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0 / n
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n
    if c.get("SQ_INSTS_VMEM_RD") and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        d["l1_lines_per_wave_load"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (c["SQ_INSTS_VMEM_RD"] + c.get("SQ_INSTS_VMEM_WR", 0.0))
    res["kernels"][k] = d
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {a: b for a, b in v.items() if not a.startswith(("SQ_", "TCC_", "TCP_", "GRBM", "FETCH", "WRITE")) or a.endswith("cycle")} for k, v in res["kernels"].items()}, indent=1))

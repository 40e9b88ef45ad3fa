#!/usr/bin/env python3
"""Development analysis (CPU, oracle only): how well do the adaptive trees of a 16-pair x 16-omega
tile overlap?  Decides the tile shape / routing of the dense (MFMA) fill.
  python tests/analysis/tree_union_stats.py [step]   (step: Newton step of the cfg3 golden chains, -1 = guesses)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.binding import Oracle  # noqa: E402

orc = Oracle()
lib = orc.lib
lib.oracle_trace_set.argtypes = [C.c_void_p, C.c_long]
lib.oracle_trace_count.restype = C.c_long

d = bench.workload_dict(256)
po = orc.params(d)
N = d["npoints"]
eta, dx = orc.grid(d["length"], N)
step = int(sys.argv[1]) if len(sys.argv) > 1 else -1
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg3_chains.npz"))
done = z["done"].astype(bool)
if step < 0:
    omegas = z["guesses"][done]
else:
    act = done & (z["iters"] > step)
    omegas = z["iterates"][act, step]
print(f"step {step}: {len(omegas)} omegas")

buf = np.zeros(4096, dtype=np.int64)


def tree(i, j, w):
    lib.oracle_trace_set(buf.ctypes.data, len(buf))
    orc.kappa(po, 0, eta[i], eta[j], complex(w))
    n = lib.oracle_trace_count()
    lib.oracle_trace_set(None, 0)
    return frozenset(buf[:n].tolist())


rng = np.random.default_rng(1)
# cost-sort omegas by a sample of pairs
probe = [(int(a), int(b)) for a, b in zip(rng.integers(0, N - 1, 6), rng.integers(1, N, 6)) if a < b]
cost = np.array([sum(len(tree(i, j, w)) for i, j in probe) for w in omegas])
order = np.argsort(-cost)
chunks = [order[k:k + 16] for k in range(0, len(order), 16)]

def tiles(kind, n):
    out = []
    for _ in range(n):
        if kind == "diag":
            off = int(rng.integers(1, N - 16))
            i0 = int(rng.integers(0, N - off - 15)) if N - off - 15 > 0 else 0
            out.append([(i0 + k, i0 + k + off) for k in range(16) if i0 + k + off < N])
        else:
            i = int(rng.integers(0, N - 17))
            j0 = int(rng.integers(i + 1, N - 15))
            out.append([(i, j0 + k) for k in range(16)])
    return out


for kind in ("diag", "row"):
    tot_elem, tot_union, tot_rounds_need = 0, 0, []
    per_depth_union = np.zeros(32)
    per_depth_elem = np.zeros(32)
    for t in tiles(kind, 6):
        for ch in chunks[:: max(1, len(chunks) // 3)]:
            trees = [tree(i, j, omegas[w]) for (i, j) in t for w in ch]
            union = set().union(*trees)
            tot_elem += sum(len(x) for x in trees)
            tot_union += len(union) * len(trees)
            cnt = {}
            for x in trees:
                for k in x:
                    cnt[k] = cnt.get(k, 0) + 1
            tot_rounds_need += list(cnt.values())
            for k, c in cnt.items():
                per_depth_union[k >> 56] += len(trees)
                per_depth_elem[k >> 56] += c
    need = np.array(tot_rounds_need)
    print(f"{kind}: mean tree {tot_elem / (tot_union / np.mean([len(need)])) if 0 else 0:.0f} efficiency (elements needing / 256 per round) = {tot_elem / tot_union:.3f}; "
          f"rounds {len(need)}, need-count quantiles 10/50/90: {np.percentile(need, [10, 50, 90])}")
    dd = np.nonzero(per_depth_union)[0]
    print("   depth: rounds-share, efficiency:", " ".join(f"{k}:{per_depth_union[k] / per_depth_union.sum():.2f}/{per_depth_elem[k] / per_depth_union[k]:.2f}" for k in dd))
    for T in (8, 16, 32, 64):
        dense = need[need >= T]
        sparse = need[need < T]
        print(f"   T={T}: dense rounds {len(dense)} ({dense.sum() / need.sum():.2f} of element-intervals), sparse element-intervals {sparse.sum()}")

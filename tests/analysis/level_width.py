#!/usr/bin/env python3
"""Development analysis (CPU, oracle only): how wide are the bisection levels of the union tree of a
16-pair tile for ONE omega?  (The dense fill keeps a level's intervals in a list of bounded length.)
  python tests/analysis/level_width.py [re im]
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
w = complex(float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else complex(-0.00552674, -0.73419159)
buf = np.zeros(1 << 16, dtype=np.int64)


def tree(i, j):
    lib.oracle_trace_set(buf.ctypes.data, len(buf))
    orc.kappa(po, 0, eta[i], eta[j], w)
    n = lib.oracle_trace_count()
    lib.oracle_trace_set(None, 0)
    return buf[:n].copy()


rng = np.random.default_rng(2)
worst = 0
for t in range(12):
    off = int(rng.integers(1, N - 16))
    i0 = int(rng.integers(0, max(1, N - off - 15)))
    pairs = [(i0 + k, i0 + k + off) for k in range(16) if i0 + k + off < N]
    trees = [tree(i, j) for i, j in pairs]
    union = set().union(*[set(x.tolist()) for x in trees])
    depth = np.array([k >> 56 for k in union])
    width = np.bincount(depth)
    own = max(np.bincount(x >> 56).max() for x in trees)
    # frontier of a FIFO walk: intervals of level d not yet processed + children of level d so far <= w[d] + w[d+1]
    front = max(width[k] + (width[k + 1] if k + 1 < len(width) else 0) for k in range(len(width)))
    worst = max(worst, width.max())
    print(f"tile off={off:3d} i0={i0:3d}: intervals per integral {np.mean([len(x) for x in trees]):.0f}, union {len(union)}, "
          f"widest level {width.max()} (depth {width.argmax()}), widest of one integral {own}, two adjacent levels {front}, depth {len(width) - 1}")
print("widest level over the sampled tiles:", worst)

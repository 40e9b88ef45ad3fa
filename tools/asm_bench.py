#!/usr/bin/env python3
"""Micro-benchmark of the assembly kernel alone (development tool, run on the GPU box):
  python tools/asm_bench.py [--n 256] [--batch 32] [--reps 3] [--stell]
Prints ms per matrix, integrand evaluations/s and the fp64-vector fraction (parity is the tests' job:
python -m pytest tests -m gpu)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import emme_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--stell", action="store_true")
    a = ap.parse_args()
    import torch
    if a.stell:
        d = dict(bench.STELLARATOR, npoints=a.n)
        g = np.linspace(-1.8, -1.4, a.batch) + 1j * np.linspace(2.2, 2.7, a.batch)
    else:
        d = bench.workload_dict(a.n)
        g = bench.lattice(1, 0)[:: max(1, 128 // a.batch)][: a.batch]
    p = emme_amd.params_from_dict(d)
    ctx = emme_amd.Context(p)
    dim = ctx.dim
    buf = torch.zeros((len(g), dim, dim), dtype=torch.complex128, device="cuda")
    ctx.assemble(g, out_device_ptr=buf.data_ptr())  # warm-up
    ctx.profile(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        iv = ctx.assemble(g, out_device_ptr=buf.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pr = ctx.profile_read()
    evals = pr.integrand_evals
    s = pr.assemble_ms * 1e-3
    print(f"union rounds {pr.union_rounds}, lane fill {pr.gk_intervals / (16.0 * max(pr.union_rounds, 1)):.3f}")
    print(f"N={a.n} batch={len(g)} reps={a.reps}: {pr.assemble_ms / a.reps / len(g):.4f} ms/matrix "
          f"({pr.assemble_ms / a.reps:.2f} ms/launch, wall {dt / a.reps * 1e3:.2f} ms), "
          f"{evals / s / 1e9:.2f} G evals/s, {evals * 900 / s / 1e12:.2f} TF-eq "
          f"({evals * 900 / s / 78.6e12 * 100:.1f}% of fp64 vector peak), "
          f"mean intervals/integral {iv.sum() / (len(g) * p.npoints * (p.npoints - 1) / 2 * (1 if dim == p.npoints else 3)):.2f}")


if __name__ == "__main__":
    main()

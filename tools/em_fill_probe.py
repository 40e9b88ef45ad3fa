#!/usr/bin/env python3
"""The electromagnetic fill alone (BASELINE configs[3]: stellarator, N = 256, dim 512, GK31): plain assembly of the 128
lattice guesses of share 0, kernel time per launch from the library's profile.  python tools/em_fill_probe.py [nrep]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
p = emme_amd.params_from_dict(dict(bench.STELLARATOR, npoints=256))
g = bench.lattice_cfg4(0)
ctx = emme_amd.Context(p)
M, iv = ctx.assemble(g, want_intervals=True)
ctx.profile(True); ctx.profile_read(reset=True)
for _ in range(nrep):
    M2, iv2 = ctx.assemble(g, want_intervals=True)
pr = ctx.profile_read()
print(f"kernel {ctx.fill_kernel()}: {pr.assemble_ms / max(pr.assemble_launches, 1):.3f} ms per launch ({pr.assemble_launches} launches), "
      f"deferred {pr.deferred_ms / nrep:.3f} ms, other {pr.other_ms / nrep:.3f} ms; intervals per integral "
      f"{iv2.sum() / (len(g) * 3 * 256 * 255 / 2):.2f}; checksum {np.abs(M2).sum():.6e} repeat diff {np.abs(M2 - M).max():.1e}")

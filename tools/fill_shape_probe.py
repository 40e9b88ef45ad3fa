#!/usr/bin/env python3
"""The dense fill against the independent-lane kernel for every (quadrature order, electrostatic / electromagnetic) shape
at N = 256: 128 omegas of the headline lattice (tokamak; electromagnetic: beta_e = 0.01), kernel time per launch from the
library's profile, interval counts compared.  python tools/fill_shape_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
g = bench.lattice(1, 0)
for pts in (15, 31):
    for em in (False, True):
        d = dict(bench.workload_dict(256), integration_start_points=pts, beta_e=0.01 if em else 0.0)
        p = emme_amd.params_from_dict(d)
        res = {}
        for name, opts in (("dense", {}), ("lanes", dict(fill=emme_amd.FILL_LANES))):
            with emme_amd.Context(p, **opts) as ctx:
                ctx.cache_settle(g)
                ctx.profile(True); ctx.profile_read(reset=True)
                for _ in range(3):
                    iv = ctx.assemble(g[:], out_device_ptr=None, want_intervals=True)[1] if False else None
                    M = None
                    ctx.assemble_rc(g)
                pr = ctx.profile_read()
                res[name] = ((pr.assemble_ms + pr.deferred_ms) / 3, ctx.fill_kernel_symbol(), ctx.node_cache_gib())
        print(f"GK{pts} {'EM' if em else 'ES'}: dense {res['dense'][0]:.3f} ms ({res['dense'][1]}, {res['dense'][2]:.0f} GiB) "
              f"lanes {res['lanes'][0]:.3f} ms ({res['lanes'][1]}, {res['lanes'][2]:.0f} GiB)", flush=True)

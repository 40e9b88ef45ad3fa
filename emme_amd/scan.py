"""Scan-axis sharding: the (parameter, omega-guess) items of a scan are independent Newton
chains (no exchange while iterating), so they are dealt round-robin to the ranks -- one
process per GPU -- and the found roots are collected with ONE all-gather at the end
(RCCL over xGMI with backend "nccl"; gloo on CPU in the tests).

The reference runs its parameter scan sequentially in one process (src/main.cpp:264-324);
this is the multi-GPU replacement for that loop's independent part.
"""
from __future__ import annotations

import numpy as np


def shard(items, world: int, rank: int):
    """Round-robin deal: interleaving spreads cheap/expensive regions of a lattice evenly."""
    return items[rank::world]


def shard_sizes(n: int, world: int):
    return [len(range(r, n, world)) for r in range(world)]


def pack(roots, iters, info) -> np.ndarray:
    out = np.empty((len(roots), 4), dtype=np.float64)
    out[:, 0] = np.real(roots)
    out[:, 1] = np.imag(roots)
    out[:, 2] = iters
    out[:, 3] = info
    return out


def unpack(packed: np.ndarray):
    return packed[:, 0] + 1j * packed[:, 1], packed[:, 2].astype(np.int32), packed[:, 3].astype(np.int32)


def gather_roots(roots, iters, info, world: int, n_total: int | None = None, force_dist: bool = False):
    """All-gather {w_re, w_im, iters, info} (32 B per item) and restore the global item
    order of `shard`.  Returns (roots, iters, info) for ALL items on every rank."""
    local = pack(roots, iters, info)
    if world == 1 and not force_dist:
        return unpack(local)
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    n_total = n_total if n_total is not None else len(roots) * world
    sizes = shard_sizes(n_total, world)
    m = max(sizes)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = torch.full((m, 4), float("nan"), dtype=torch.float64, device=dev)
    buf[:len(local)] = torch.from_numpy(local).to(dev)
    out = torch.empty((world, m, 4), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out.view(world * m, 4), buf)
    out = out.cpu().numpy()
    glob = np.empty((n_total, 4), dtype=np.float64)
    for r in range(world):
        glob[r::world] = out[r, :sizes[r]]
    assert len(local) == sizes[rank]
    return unpack(glob)

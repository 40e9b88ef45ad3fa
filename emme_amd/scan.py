"""Scan-axis sharding: the (parameter, omega-guess) items of a scan are independent Newton
chains (no exchange while iterating), so they are dealt round-robin to the ranks -- one
process per GPU -- and the found roots are collected with ONE all-gather at the end.

The collective itself is C++ behind the C ABI: `emme_gather_roots` (emme_amd/csrc/gather_rccl.cpp),
one `ncclAllGather` of 32 B per item over xGMI.  `ScanGather` only carries the 128-byte RCCL unique
id from rank 0 to the other ranks -- through `torch.distributed` when a process group exists
(bench.py under torch.distributed.run), or a file -- and calls it.  On CPU (the gloo tests, which
have no RCCL) the same packing goes through `torch.distributed.all_gather_into_tensor`.

The reference runs its parameter scan sequentially in one process (src/main.cpp:264-324);
this is the multi-GPU replacement for that loop's independent part.
"""
from __future__ import annotations

import os
import time

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


def _check_share(n_local: int, n_total: int, world: int, rank: int):
    # BEFORE any collective: ranks that disagree about the shares would enter it with
    # different buffer sizes (a hang or corruption on RCCL)
    want = shard_sizes(n_total, world)[rank]
    if n_local != want:
        raise ValueError(f"rank {rank}: {n_local} local items, but the round-robin share of "
                         f"{n_total} items over {world} ranks is {want}")


def gather_roots(roots, iters, info, world: int, n_total: int | None = None, force_dist: bool = False):
    """All-gather {w_re, w_im, iters, info} (32 B per item) through torch.distributed (the CPU /
    gloo form of `emme_gather_roots`) and restore the global item order of `shard`.  Returns
    (roots, iters, info) for ALL items on every rank.  `n_total` is required when world > 1."""
    local = pack(roots, iters, info)
    if world == 1 and not force_dist:
        return unpack(local)
    import torch
    import torch.distributed as dist

    if n_total is None:
        if world > 1:
            raise ValueError("gather_roots: n_total is required when world > 1 (ragged shares "
                             "cannot be derived from the local count)")
        n_total = len(local)
    rank = dist.get_rank()
    _check_share(len(local), n_total, world, rank)
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
    return unpack(glob)


class ScanGather:
    """The scan's one collective on GPUs: RCCL through the C ABI (`emme_comm_*`,
    `emme_gather_roots`).  Collective constructor: every rank must create it."""

    def __init__(self, rank: int, world: int, device: int = -1, id_file: str | None = None):
        import emme_amd
        self.rank, self.world = rank, world
        if id_file is None:
            import torch.distributed as dist
            if world > 1 and not dist.is_initialized():
                raise RuntimeError("ScanGather needs a torch.distributed process group or an id_file "
                                   "to distribute the RCCL unique id")
            box = [emme_amd.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            uid = box[0]
        else:
            if rank == 0:
                uid = emme_amd.comm_unique_id()
                with open(id_file + ".tmp", "wb") as f:
                    f.write(uid)
                os.replace(id_file + ".tmp", id_file)
            else:
                t0 = time.time()
                while not os.path.exists(id_file):
                    if time.time() - t0 > 120:
                        raise TimeoutError(f"no RCCL unique id at {id_file}")
                    time.sleep(0.05)
                uid = open(id_file, "rb").read()
        self.comm = emme_amd.Comm(uid, rank, world, device)

    def gather(self, roots, iters, info, n_total: int, stream_handle: int = 0):
        _check_share(len(roots), n_total, self.world, self.rank)
        return self.comm.gather_roots(roots, iters, info, n_total, stream_handle)

    def close(self):
        self.comm.close()

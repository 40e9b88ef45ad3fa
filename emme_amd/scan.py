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
    # the slot mapping is the C ABI's (emme_gather_pack / emme_gather_unpack, host-only functions of
    # gather_rccl.cpp): this path differs from emme_gather_roots only in who moves the bytes
    import emme_amd
    send = emme_amd.gather_pack(rank, world, roots, iters, info, n_total)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = torch.from_numpy(send).to(dev)
    out = torch.empty(world * len(send), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf)
    return emme_amd.gather_unpack(world, n_total, out.cpu().numpy())


# ---- rendezvous of the RCCL unique id through a file (no process group) -----------------------------------------
# File = 32-byte sha256 of the run id + the 128-byte id.  Rank 0 unlinks a stale file BEFORE it asks RCCL for an id,
# writes atomically (rename), and removes the file again in close(); the other ranks accept only a file that
# carries THEIR run id, so the id of an earlier run of the same script (same path) is never taken for this one's
# -- ncclCommInitRank on a mismatched id hangs.
def _run_tag(run_id: str) -> bytes:
    import hashlib
    return hashlib.sha256(run_id.encode()).digest()


def write_id_file(path: str, uid: bytes, run_id: str) -> None:
    with open(path + ".tmp", "wb") as f:
        f.write(_run_tag(run_id) + uid)
    os.replace(path + ".tmp", path)


def clear_id_file(path: str) -> None:
    for q in (path, path + ".tmp"):
        try:
            os.unlink(q)
        except FileNotFoundError:
            pass


def read_id_file(path: str, run_id: str, timeout: float = 120.0, poll: float = 0.05) -> bytes:
    tag, t0 = _run_tag(run_id), time.time()
    while True:
        try:
            data = open(path, "rb").read()
        except FileNotFoundError:
            data = b""
        if len(data) >= 32 and data[:32] == tag:
            return data[32:]
        if time.time() - t0 > timeout:
            raise TimeoutError(f"no RCCL unique id of run {run_id!r} at {path}"
                               + (" (a file of another run is there)" if data else ""))
        time.sleep(poll)


class ScanGather:
    """The scan's one collective on GPUs: RCCL through the C ABI (`emme_comm_*`,
    `emme_gather_roots`).  Collective constructor: every rank must create it.

    With a torch.distributed process group the ranks AGREE before and after the collective
    ncclCommInitRank (an all-reduce of an ok flag over the existing group): either every rank ends up
    with the C-ABI communicator or every rank raises `ScanGatherUnavailable` -- never some ranks on
    RCCL-through-the-ABI and others on a fall-back, never a rank left waiting in the initialisation for
    peers that gave up before it."""

    def __init__(self, rank: int, world: int, device: int = -1, id_file: str | None = None,
                 run_id: str | None = None):
        import emme_amd
        self.rank, self.world, self.comm, self._id_file = rank, world, None, None
        if id_file is None:
            import torch
            import torch.distributed as dist
            if world > 1 and not dist.is_initialized():
                raise RuntimeError("ScanGather needs a torch.distributed process group or an id_file "
                                   "to distribute the RCCL unique id")

            def agree(ok: bool) -> bool:
                if world == 1:
                    return ok
                dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(flag.item())

            # 1. not a collective: can this process bind RCCL at all, and can rank 0 make an id?
            uid, why = None, ""
            try:
                if not emme_amd.comm_available():
                    raise emme_amd.EmmeError(-3, "RCCL cannot be bound in this process")
                if rank == 0:
                    uid = emme_amd.comm_unique_id()
            except Exception as e:  # noqa: BLE001
                why = str(e)
            if not agree(not why):
                raise ScanGatherUnavailable(why or "another rank cannot bind RCCL")
            # 2. the id travels over the existing group, then the collective initialisation
            box = [uid]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            try:
                self.comm = emme_amd.Comm(box[0], rank, world, device)
            except Exception as e:  # noqa: BLE001
                why = str(e)
            if not agree(self.comm is not None):
                self.close()
                raise ScanGatherUnavailable(why or "ncclCommInitRank failed on another rank")
        else:
            run_id = run_id or os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("EMME_RUN_ID")
            if not run_id:
                raise ValueError("ScanGather(id_file=...) needs a run_id shared by the ranks of THIS run (or "
                                 "TORCHELASTIC_RUN_ID / EMME_RUN_ID in the environment): the file of an earlier run "
                                 "must not be taken for this one's")
            if rank == 0:
                clear_id_file(id_file)
                self._id_file = id_file
                uid = emme_amd.comm_unique_id()
                write_id_file(id_file, uid, run_id)
            else:
                uid = read_id_file(id_file, run_id)
            self.comm = emme_amd.Comm(uid, rank, world, device)

    def gather(self, roots, iters, info, n_total: int, stream_handle: int = 0):
        _check_share(len(roots), n_total, self.world, self.rank)
        return self.comm.gather_roots(roots, iters, info, n_total, stream_handle)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self._id_file:
            clear_id_file(self._id_file)
            self._id_file = None


class ScanGatherUnavailable(RuntimeError):
    """Raised on EVERY rank alike when the C-ABI communicator cannot be had."""

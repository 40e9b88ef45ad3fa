"""N>1 path on CPU: two gloo ranks shard a guess lattice, each 'solves' its share with a
stand-in for the device call, and one all-gather restores the global list."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    import torch.distributed as dist

    from emme_amd.scan import gather_roots, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = np.arange(n_total) * (1.0 + 0.5j)
    mine = shard(items, world, rank)
    roots = mine * 2.0  # stand-in for Context.solve_roots on this rank's GPU
    iters = (np.real(mine) % 7).astype(np.int32)
    info = np.zeros(len(mine), dtype=np.int32)
    R, I, F = gather_roots(roots, iters, info, world, n_total)
    q.put((rank, R, I, F))
    dist.barrier()
    dist.destroy_process_group()


def _worker_unavailable(rank, world, port, n_total, q):
    """bench.py's choice of the gather: without a GPU the C-ABI communicator cannot be created (rank 0 fails in
    ncclGetUniqueId, rank 1 would have gone on to the collective initialisation): BOTH ranks must end up with
    ScanGatherUnavailable -- agreed over the process group -- and then use the same fall-back collective."""
    import torch.distributed as dist

    from emme_amd.scan import ScanGather, ScanGatherUnavailable, gather_roots, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    outcome = "created"
    try:
        ScanGather(rank, world)
    except ScanGatherUnavailable:
        outcome = "unavailable"
    items = np.arange(n_total) * (1.0 + 0.5j)
    mine = shard(items, world, rank)
    R, I, F = gather_roots(mine * 2.0, (np.real(mine) % 7).astype(np.int32), np.zeros(len(mine), dtype=np.int32), world, n_total)
    q.put((rank, outcome, R))
    dist.barrier()
    dist.destroy_process_group()


def test_all_ranks_agree_when_the_c_abi_gather_is_unavailable():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the communicator can be created")
    res = _run(2, 9, target=_worker_unavailable)
    assert sorted(r[1] for r in res) == ["unavailable", "unavailable"]
    for _, _, R in res:
        assert np.array_equal(R, np.arange(9) * (1.0 + 0.5j) * 2.0)


def _run(world, n_total, target=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target or _worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_ranks_gather_restores_global_order():
    n_total = 37  # ragged: 19 + 18
    for rank, R, I, F in _run(2, n_total):
        want = np.arange(n_total) * (1.0 + 0.5j) * 2.0
        assert np.array_equal(R, want)
        assert np.array_equal(I, (np.arange(n_total) % 7).astype(np.int32))
        assert (F == 0).all()


def test_shard_helpers():
    from emme_amd.scan import shard, shard_sizes
    x = np.arange(10)
    assert [list(shard(x, 3, r)) for r in range(3)] == [[0, 3, 6, 9], [1, 4, 7], [2, 5, 8]]
    assert shard_sizes(10, 3) == [4, 3, 3] and shard_sizes(2, 4) == [1, 1, 0, 0]


def test_bench_lattice_is_partitioned_without_overlap():
    import bench
    full = set()
    for r in range(4):
        g = bench.lattice(4, r)
        assert len(g) == 128
        full |= {complex(v) for v in g}
    assert len(full) == 512
    assert np.array_equal(bench.lattice(1, 0)[:16].real, np.linspace(-1.2, -0.4, 16))


def test_gather_slot_mapping_of_the_c_abi_ragged_shares():
    """emme_gather_pack / emme_gather_unpack (the host-only half of emme_gather_roots, gather_rccl.cpp): every
    rank's send buffer concatenated = what ncclAllGather delivers; unpacking restores item order -- for ragged
    shares (n_total % world != 0), ranks without any item (rank >= n_total) and a world of one; a wrong share is
    refused before anything is packed."""
    import emme_amd
    import pytest
    lib = emme_amd.load()
    for world, n_total in [(1, 5), (2, 37), (3, 10), (4, 2), (8, 1024), (8, 1021), (8, 3), (5, 5)]:
        items = (np.arange(n_total) + 1) * (1.0 - 0.25j)
        sends = []
        for rank in range(world):
            mine = items[rank::world]
            assert lib.emme_gather_share(n_total, world, rank) == len(mine)
            sends.append(emme_amd.gather_pack(rank, world, mine * 3.0, np.arange(len(mine)) + 10 * rank,
                                              -np.ones(len(mine), dtype=np.int32) * rank, n_total))
            assert len(sends[-1]) == 4 * lib.emme_gather_slots(n_total, world)
            assert np.isnan(sends[-1][4 * len(mine):]).all()
        R, I, F = emme_amd.gather_unpack(world, n_total, np.concatenate(sends))
        assert np.array_equal(R, items * 3.0)
        for k in range(n_total):
            assert I[k] == k // world + 10 * (k % world) and F[k] == -(k % world)
    with pytest.raises(emme_amd.EmmeError):
        emme_amd.gather_pack(1, 2, np.zeros(3, dtype=complex), np.zeros(3), np.zeros(3), 5)  # rank 1's share of 5 is 2
    assert lib.emme_gather_slots(0, 2) < 0 and lib.emme_gather_share(5, 2, 2) < 0


def test_id_file_rendezvous_ignores_another_runs_file(tmp_path):
    """ScanGather's file rendezvous: rank 0 removes a stale file before it creates its id, the file carries a tag of
    the run, and a reader takes only a file of ITS run (an old id would make ncclCommInitRank hang)."""
    import threading
    import time

    import pytest

    from emme_amd.scan import clear_id_file, read_id_file, write_id_file
    path = str(tmp_path / "rccl_id")
    write_id_file(path, b"A" * 128, "run-1")
    assert read_id_file(path, "run-1", timeout=1.0) == b"A" * 128
    with pytest.raises(TimeoutError):
        read_id_file(path, "run-2", timeout=0.3)          # a stale file of another run is not accepted
    got = []
    t = threading.Thread(target=lambda: got.append(read_id_file(path, "run-2", timeout=10.0)))
    t.start()
    time.sleep(0.2)
    clear_id_file(path)                                   # rank 0 of run 2: unlink, then the new id
    write_id_file(path, b"B" * 128, "run-2")
    t.join(timeout=10)
    assert got == [b"B" * 128]
    clear_id_file(path)
    assert not os.path.exists(path)


def test_predicted_share_imbalance_of_the_deal():
    """No multi-GPU node is available to the builder: the per-rank cost of the deal is MEASURED on one GPU, share after
    share (tools/scaling_prediction.py -> profiles/r03_scaling_prediction.json), and pinned here: with the skewed
    deal of bench.lattice the slowest share of the weak-scaling lattice stays within 1.3x the mean for 2, 4 and 8
    ranks (plain round-robin: 2.4x at 8 ranks -- a rank then holds the same Re omega columns in every row, and the
    chains that never converge start in a few of those columns), configs[3]'s shares within 1.05x.  The real curve
    is the driver's SCALE_rNN.json; none has been measured (no 8-GPU node in this pool so far)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_scaling_prediction.json")
    d = json.load(open(path))
    lat = d["config3_headline_lattice"]
    for world in ("2", "4", "8"):
        assert len(lat[world]["share_ms"]) == int(world)
        assert lat[world]["max_over_mean"] <= 1.3, (world, lat[world])
    if "config4_shares" in d:
        assert d["config4_shares"]["max_over_mean"] <= 1.05

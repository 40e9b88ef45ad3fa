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


def _run(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
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

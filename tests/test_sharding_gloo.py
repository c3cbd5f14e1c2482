"""CPU-only, world_size 2, gloo: the multi-GPU path's host logic.
One process per rank; rank 0 owns the weights and broadcasts the packed blob once; each
rank processes its contiguous utterance shard with NO data-path collective; outputs
gathered for the check only.  The per-shard compute here is the CPU oracle (this
container has no GPU); on the GPU box the same `shard_range` / `broadcast_weights` feed
bench.py's per-rank device pipeline, and tests/test_gpu_parity.py proves sharding is
bit-identical on the device kernels."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, q):
    import torch
    import torch.distributed as dist
    import oracle as O
    import bench
    from nntoolkitcore_amd.sharding import broadcast_weights, gather_shards, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parts = bench.make_weights("gru", seed=5)
        flat = bench.pack(parts)
        mine = flat.copy() if rank == 0 else np.full_like(flat, np.nan)     # only rank 0 has the weights
        got = broadcast_weights(mine, torch, dist)
        assert np.array_equal(got, flat)
        w = bench.unpack(got, parts)
        x = np.random.default_rng(9).standard_normal((B, 6, 128)).astype(np.float32)     # same on every rank
        lo, hi = shard_range(B, world, rank)
        local = O.gru(x[lo:hi], w["g1_W"], w["g1_U"], w["g1_bi"], w["g1_bh"]) if hi > lo else \
            np.zeros((0, 6, 256), np.float32)
        full = gather_shards(local, torch, dist)
        if rank == 0:
            ref = O.gru(x, w["g1_W"], w["g1_U"], w["g1_bi"], w["g1_bh"])
            q.put(("ok", bool(np.array_equal(full, ref)), (lo, hi)))
    except Exception as e:                                                   # pragma: no cover
        q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2])
def test_two_rank_sharded_run_equals_single_process(B):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    status, same, rng0 = q.get(timeout=5)
    assert status == "ok" and same is True
    assert rng0 == (0, (B + 1) // 2)


def test_shard_ranges_partition_the_batch():
    from nntoolkitcore_amd.sharding import shard_range
    for B in (0, 1, 7, 8, 512, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_weight_packing_round_trip():
    import bench
    for wl in ("stack", "conv", "gru", "spectrogram"):
        parts = bench.make_weights(wl, 3)
        flat = bench.pack(parts)
        back = bench.unpack(flat, parts)
        assert all(np.array_equal(back[k], v) for k, v in parts.items())

"""world_size-2 gloo tests (CPU) of the data-parallel host logic of the training step (trainer.Comm) and of the
replica sharding bench.py uses for the reconstruction metric."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neural_photo_editor_amd.trainer import Comm
        comm = Comm(bucket_bytes=1000)          # tiny buckets: 250 floats -> many pieces
        assert comm.world == world and comm.rank == rank
        # gradient all-reduce: per-rank shard gradients sum to the full-batch gradient, whatever the bucket cut
        rs = np.random.RandomState(0)
        full = rs.randn(world, 1777).astype(np.float32)
        g = torch.from_numpy(full[rank].copy())
        works = comm.all_reduce_buckets(g, async_op=True)
        for w in works:
            w.wait()
        ok1 = np.allclose(g.numpy(), full.sum(0), rtol=1e-6, atol=1e-6)
        # SyncBN statistics: sums of (x, x^2) over shards == sums over the whole batch
        x = rs.randn(world * 8, 16).astype(np.float32)
        local = torch.from_numpy(x[rank * 8:(rank + 1) * 8])
        s = torch.stack([local.sum(0), (local ** 2).sum(0)]).reshape(-1)
        comm.all_reduce_sum(s)
        ok2 = np.allclose(s.numpy(), np.concatenate([x.sum(0), (x ** 2).sum(0)]), rtol=1e-5, atol=1e-5)
        # MinibatchLayer all-gather: rank r owns rows [r*n, (r+1)*n)
        out = torch.zeros(world * 8, 16)
        comm.all_gather_rows(local, out)
        ok3 = np.array_equal(out.numpy(), x)
        q.put((rank, ok1, ok2, ok3))
    finally:
        dist.destroy_process_group()


def test_comm_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def test_comm_single_process_is_a_noop():
    from neural_photo_editor_amd.trainer import Comm
    c = Comm()
    assert c.world == 1 and c.rank == 0
    t = torch.ones(5)
    assert c.all_reduce_buckets(t) == [] and c.all_reduce_sum(t) is None
    out = torch.zeros(5)
    c.all_gather_rows(t, out)
    assert torch.equal(out, t)


def test_replica_sharding_of_the_reconstruction_metric():
    """bench.py: N ranks process disjoint shards; whole-job value = sum of per-rank images / max-over-ranks time."""
    per_rank, steps = 64, 10
    times = [0.016, 0.017]
    value = len(times) * per_rank * steps / max(times)
    assert value == pytest.approx(2 * 64 * 10 / 0.017)


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (torch.distributed.run on
    127.0.0.1), rendezvous, take the max over ranks and print ONE line with n_gpus == 2.  --dry-run keeps the GPU out
    of it so the control flow is covered on CPU (gloo); the timed HIP legs are the same code after the rendezvous."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env["IAN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["dry_run"] is True

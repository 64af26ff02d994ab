"""The sharded training step against the single-process step on the same GLOBAL minibatch, at any (world, per-rank batch):
`world` ranks time-share cuda:0 over gloo (RCCL needs one GPU per rank; the collective call sequence is the same) and run the
REAL C++ sequencer (csrc/ian_trainer.cpp: HIP kernels, gradient buckets handed over during backward, SyncBN combine in rank
order, MinibatchLayer all-gather over the global minibatch) in `exact` mode; one process then runs the same minibatch at batch
world * n.  Used by tests/test_gpu_dp.py (2 x 2, 4 x 32) and by scripts/exp/config5_rehearsal.py at BASELINE.json configs[4]'s
real shape, 8 ranks x 128 images = global batch 1024 (train_IAN.py:116-149 is a function of the WHOLE minibatch: Lasagne batch
statistics, layers.py:506-520 pairwise abs_dif over all B samples)."""
import os
import socket
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py")


def inputs(B):
    """Well separated samples (see test_gpu_train.diverse_images): a random-init decoder emits near-identical images, for which the
    MinibatchLayer's |a_b - a_b'| gradients flip sign under 1e-7 perturbations (such as another partial-sum order of the batch
    statistics); the encoder passes on X_hat / X_gen are therefore fed images through the Trainer.forward test hook."""
    from oracle import ian_oracle as O
    rs = np.random.RandomState(11)
    if B == 4:      # the round-1..4 case, kept bit for bit
        s = np.array([0.2, 0.5, 0.8, 1.0], np.float32).reshape(-1, 1, 1, 1)
        o = np.array([-0.5, 0.3, -0.1, 0.0], np.float32).reshape(-1, 1, 1, 1)
    else:
        s = np.linspace(0.2, 1.0, B).astype(np.float32).reshape(-1, 1, 1, 1)
        o = rs.uniform(-0.5, 0.3, B).astype(np.float32).reshape(-1, 1, 1, 1)
    X = np.clip(O.make_images(B, seed=3) * s + o, -1, 1).astype(np.float32)
    Z = O.make_latents(B, seed=8)
    eps = np.random.RandomState(9).randn(B, 100).astype(np.float32)
    Xh = np.clip(O.make_images(B, seed=4)[::-1] * s + o, -1, 1).astype(np.float32)
    Xg = np.clip(np.roll(O.make_images(B, seed=5), 1, 0) * s[::-1] + o, -1, 1).astype(np.float32)
    return X, Z, eps, Xh, Xg


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, out_dir, B, timed):
    import torch
    import torch.distributed as dist
    torch.set_num_threads(4)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ian_oracle as O
        from oracle.train_twin import make_train_params
        from neural_photo_editor_amd.trainer import Trainer, Comm
        torch.cuda.set_device(0)
        P = make_train_params(O.make_params("IAN", 1))
        n = B // world
        tr = Trainer(CFG, P, batch=n, comm=Comm(), exact=True)
        assert tr.N == B and tr.exact
        X, Z, eps, Xh, Xg = inputs(B)
        sl = slice(rank * n, (rank + 1) * n)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).cuda()
        res = {}
        for which in ("gen", "discrim"):
            upd = "dec" if which == "gen" else "enc"
            # first sweep of this kind: the gradient-write order is recorded, buckets are reduced after backward
            tr.forward(d(X), d(Z), d(eps), xhat_override=d(Xh), xgen_override=d(Xg))
            m = tr.metrics()
            tr.backward(which)
            tr._finish_allreduce(which)
            torch.cuda.synchronize()
            first = {g: tr.groups[g].g.clone() for g in (upd, "Z")}
            n_first = len(tr.overlap_log)
            # second sweep, same inputs and parameters: every bucket is handed to the all-reduce right after its last
            # writer, while backward is still being issued
            tr.forward(d(X), d(Z), d(eps), xhat_override=d(Xh), xgen_override=d(Xg))
            tr.backward(which)
            tr._finish_allreduce(which)
            torch.cuda.synchronize()
            for g in (upd, "Z"):
                assert torch.equal(first[g], tr.groups[g].g), "overlapped all-reduce changed the %s gradients" % g
                res["%s/%s" % (which, g)] = tr.groups[g].g.cpu().numpy()
            log = list(tr.overlap_log)[n_first:]
            assert len(log) == tr.plan_size(which) and all(r["which"] == which for r in log)
            early = [r for r in log if r["issued_at_write"] < r["writes_in_backward"]]
            res["%s/early" % which] = np.array([len(early), len(log)])
            zb = [r for r in log if r["bucket"][0] == "Z"]
            assert zb and all(r["issued_at_write"] < 0.8 * r["writes_in_backward"] for r in zb), zb   # Z_params: long before the end
            assert len(early) >= len(log) - 1, log     # at most the bucket holding the very last written tensor waits for the end
            res["%s/metrics" % which] = np.array([m[k] for k in sorted(m)])
        if timed:
            # the one-call entry on the same shard: wall time per update with the other ranks time-sharing the GPU (NOT an 8-GPU
            # number), the compute stream's stall at wait_all and its time inside the exact-mode all-gathers
            tr.measure_exposed = True
            for which in ("gen", "discrim"):
                for _ in range(2):
                    tr.step(which, d(X), d(Z), d(eps), return_metrics=False)
                torch.cuda.synchronize()
                dist.barrier()
                tr.measure_exposed = True
                t0 = time.perf_counter()
                for _ in range(timed):
                    tr.step(which, d(X), d(Z), d(eps), return_metrics=False)
                torch.cuda.synchronize()
                res["%s/wall_ms" % which] = np.array((time.perf_counter() - t0) / timed * 1e3)
                res["%s/exposed_ms" % which] = np.array(tr.allreduce_exposed_ms()[which])
                ag = tr.allgather_ms()[which]
                res["%s/gather" % which] = np.array([ag["ms"], ag["calls"]])
        if rank == 0:
            np.savez(os.path.join(out_dir, "dp.npz"), **res)
        tr.close()
    finally:
        dist.destroy_process_group()


def run_ranks(world, B, out_dir, timed=0, timeout=1500):
    """`world` processes sharing cuda:0; returns rank 0's record (gradients of the updated groups after the all-reduce, metrics,
    bucket hand-over counts)."""
    import torch.multiprocessing as mp
    port = free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, world, port, str(out_dir), B, timed)) for r in range(world)]
    for p in procs:
        p.start()
    deadline = time.time() + timeout
    for p in procs:
        p.join(timeout=max(1.0, deadline - time.time()))
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert codes == [0] * world, codes
    return np.load(os.path.join(str(out_dir), "dp.npz"))


def single_process_errors(dp, B):
    """One process at the GLOBAL batch B against the sharded record: {key: [(relative L2 error, tensor name), ...] worst first},
    metrics checked at 1e-5."""
    import torch
    from oracle import ian_oracle as O
    from oracle.train_twin import make_train_params
    from neural_photo_editor_amd.trainer import Trainer
    P = make_train_params(O.make_params("IAN", 1))
    tr = Trainer(CFG, P, batch=B)
    X, Z, eps, Xh, Xg = inputs(B)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # relative L2 error per tensor.  The batch statistics are rank-order invariant (per-image chunks + pairwise tree,
    # kernels_train.hip / allreduce_ordered), so the forward activations of the two runs are IDENTICAL and no leaky-ReLU / |.|
    # branch can flip; what remains is the float32 summation order of the weight gradients (per-rank partial sums added by the
    # all-reduce vs one sum over the whole minibatch)
    rel = lambda a, b: float(np.linalg.norm((a - b).astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))
    diag = {}
    for which in ("gen", "discrim"):
        tr.forward(d(X), d(Z), d(eps), xhat_override=d(Xh), xgen_override=d(Xg))
        m = tr.metrics()
        tr.backward(which)
        tr._finish_allreduce(which)          # world 1: joins the weight-gradient stream (no collective)
        got, want = dp["%s/metrics" % which], np.array([m[k] for k in sorted(m)])
        # a saturated cross-entropy (-log of a probability that underflowed: +inf on these well-separated synthetic images at
        # large batches) must saturate identically in both runs; the finite ones agree to 1e-5
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin) and np.array_equal(got[~fin], want[~fin]), (which, got, want)
        assert np.allclose(got[fin], want[fin], rtol=1e-5, atol=1e-6), (which, got, want)
        diag["%s/metrics_max_rel" % which] = float(np.max(np.abs(got[fin] - want[fin]) / (np.abs(want[fin]) + 1e-6)))
        diag["%s/metrics_nonfinite_in_both" % which] = int((~fin).sum())
        for g in (("dec" if which == "gen" else "enc"), "Z"):
            ref = tr.groups[g].g.cpu().numpy()
            grp = tr.groups[g]
            diag["%s/%s" % (which, g)] = sorted(((rel(dp["%s/%s" % (which, g)][o:o + c], ref[o:o + c]), n) for n, (o, c, _) in grp.offsets.items()),
                                                reverse=True)
    tr.close()
    return diag

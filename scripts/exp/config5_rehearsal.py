"""BASELINE.json configs[4] at its REAL shape, rehearsed on the one-GPU box: 8 ranks x 128 images = global batch 1024, `exact` mode
(SyncBN statistics combined in rank order by the 8-way tree, MinibatchLayer over all 1024 rows -- layers.py:506-520 couples the
whole minibatch, train_IAN.py:116-149 normalises over it), through ian_train_step's sequencer, against ONE process at batch 1024.

  python scripts/exp/config5_rehearsal.py [world=8] [global_batch=1024]  ->  gpurun_out/$REC_NAME (default r06_config5_rehearsal.json)

Part 1 (parity): 8 gloo ranks time-sharing the MI355X (tests/dp_rehearsal.py, the worker of tests/test_gpu_dp.py) vs the single
process: losses, all three gradient groups per tensor (relative L2), bucket hand-over during backward at the 8-rank plans.
Part 2 (cost): what `exact` mode adds on the COMPUTE side at world 8, measured without the time-sharing noise of part 1: one
process at 128 images with a stand-in communicator whose all-gathers replicate the rank's own rows 8 times on the device
(so the 1024-row MinibatchLayer and the 8-way ordered tree run for real) and whose all-reduce is the identity, against the
plain single-GPU step at 128 images.  The xGMI side (8 real ranks) cannot be measured here."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["IAN_OPTS"] = "tg_split=0"        # per-image results bitwise independent of the per-rank batch (tests/test_gpu_dp.py)

import dp_rehearsal  # noqa: E402


class ReplicatingComm:
    """world ranks that all hold THIS rank's data: all-gather = `world` device copies, all-reduce = identity."""
    active, group, bucket_bytes, filler = True, None, 16 << 20, "stand-in (replicating, no wire)"

    def __init__(self, world):
        self.world, self.rank = world, 0

    def ops(self, torch):
        from neural_photo_editor_amd.trainer import build_ops, device_view
        self.errors = []

        def on(stream):
            return torch.cuda.stream(torch.cuda.default_stream() if not stream else torch.cuda.ExternalStream(int(stream)))

        def allreduce(buf, count, stream):
            pass

        def wait_all(stream):
            pass

        def allgather(src, dst, count, stream):
            with on(stream):
                s = device_view(torch, src, (1, int(count)))
                device_view(torch, dst, (self.world, int(count))).copy_(s.expand(self.world, -1))

        return build_ops(self.world, self.rank, allreduce, wait_all, allgather, self.errors)


def cost_of_exact(world, n, iters=4):
    import torch
    from neural_photo_editor_amd import synthetic as S
    from neural_photo_editor_amd.trainer import Trainer
    P = S.make_train_params(S.make_params("IAN", 1))
    rs = np.random.RandomState(0)
    X = torch.from_numpy(S.make_images(n, seed=1)).cuda()
    Z = torch.from_numpy(rs.randn(n, 100).astype(np.float32)).cuda()
    eps = torch.from_numpy(rs.randn(n, 100).astype(np.float32)).cuda()
    out = {}
    for label, comm in (("single_gpu", None), ("exact_world%d_standin" % world, ReplicatingComm(world))):
        tr = Trainer(dp_rehearsal.CFG, P, batch=n, comm=comm, exact=True)
        tr.autotune()
        rec = {}
        for which in ("gen", "discrim"):
            for _ in range(2):
                tr.step(which, X, Z, eps, return_metrics=False)
            torch.cuda.synchronize()
            if comm is not None:
                tr.measure_exposed = True
            t0 = time.perf_counter()
            for _ in range(iters):
                tr.step(which, X, Z, eps, return_metrics=False)
            torch.cuda.synchronize()
            rec["update_%s_ms" % which] = (time.perf_counter() - t0) / iters * 1e3
            if comm is not None:
                ag = tr.allgather_ms()[which]
                rec["allgather_device_copies_ms_%s" % which] = ag["ms"]
                rec["allgathers_per_update_%s" % which] = ag["calls"]
        out[label] = rec
        tr.close()
    a, b = out["single_gpu"], out["exact_world%d_standin" % world]
    out["exact_mode_compute_side_cost_ms"] = {w: b["update_%s_ms" % w] - a["update_%s_ms" % w] for w in ("gen", "discrim")}
    return out


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n = B // world
    rec = {"what": "BASELINE.json configs[4]: train_IAN.py step, global batch %d = %d ranks x %d images, exact mode; ranks time-share ONE "
                   "MI355X over gloo (RCCL needs one GPU per rank), single process at batch %d as the reference" % (B, world, n, B),
           "world": world, "per_rank_batch": n, "global_batch": B, "IAN_OPTS": os.environ["IAN_OPTS"]}
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        dp = dp_rehearsal.run_ranks(world, B, tmp, timed=2, timeout=2400)
        dp = {k: dp[k] for k in dp.files}
    rec["sharded_wall_s"] = time.time() - t0
    for which in ("gen", "discrim"):
        rec["buckets_handed_over_before_backward_ended_%s" % which] = [int(v) for v in dp["%s/early" % which]]
        rec["shared_gpu_update_wall_ms_%s" % which] = float(dp["%s/wall_ms" % which])
        rec["allreduce_exposed_ms_%s" % which] = float(dp["%s/exposed_ms" % which])
        rec["allgather_ms_and_calls_%s" % which] = [float(v) for v in dp["%s/gather" % which]]
    rec["note_wall"] = ("shared_gpu_update_wall_ms: %d processes time-slicing one GPU with gloo collectives through host memory -- a functional "
                        "record, NOT an %d-GPU time" % (world, world))
    t0 = time.time()
    diag = dp_rehearsal.single_process_errors(dp, B)
    rec["single_process_wall_s"] = time.time() - t0
    worst = 0.0
    for k, v in diag.items():
        if isinstance(v, list):
            rec["grad_rel_l2/" + k] = {"worst": v[:5], "median": float(np.median([e for e, _ in v])), "tensors": len(v)}
            worst = max(worst, v[0][0])
        else:
            rec[k] = v
    rec["grad_rel_l2_worst"] = worst
    rec["bar"] = {"losses_rel": 1e-5, "grad_rel_l2_per_tensor": 3e-5, "pass": bool(worst < 3e-5)}
    os.environ.pop("IAN_OPTS")                   # part 2 times the production schedules (split-K on, autotuned)
    rec["exact_mode_cost"] = cost_of_exact(world, n)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", os.environ.get("REC_NAME", "r06_config5_rehearsal.json")), "w") as fh:
        json.dump(rec, fh, indent=1, default=lambda o: float(o) if isinstance(o, (np.floating, float)) else str(o))
    print(json.dumps({k: rec[k] for k in rec if not k.startswith("grad_rel_l2/")}, default=str)[:3000])
    print("worst gradient tensor error %.3g (bar 3e-5): %s" % (worst, "PASS" if worst < 3e-5 else "FAIL"))


if __name__ == "__main__":
    main()

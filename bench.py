#!/usr/bin/env python
"""Benchmark of the IAN hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
encode -> z -> decode of IAN_simple at batch 64 (BASELINE.json configs[1]).  `value` = whole-job
64x64 reconstructions per second over all ranks (replicas: the path has no data-path collective).

Extra fields in the same JSON line:
  roofline      dominant kernel (tapgemm, fp32 MFMA): algorithmic FLOP per launch / mean launch duration,
                measured with HIP events recorded on the launch stream inside libian (ian_profile_*),
                in a separate profiled pass of the same workload (event records perturb the timed pass);
  cpu_baseline  the torch-CPU twin of the oracle ("port": the reference's Theano path cannot run here),
                rank 0, N=1 only, bounded sample;
  edit_step     p50/p95 latency of one NPE latent-brush step (BASELINE.json configs[3]): imgradRGB +
                Z update (reference gradient descent, NPE.py:199-209) + sample_at, batch 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
FLOP_PER_RECON = {"IAN_simple": 2.592e9, "IAN": 8.463e9}


def cpu_baseline(arch, P, batch, budget_s=20.0):
    import torch
    from oracle.torch_twin import TorchTwin      # the ONLY use of oracle/ in this file: the timed CPU baseline
    from neural_photo_editor_amd import synthetic as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    tw = TorchTwin(arch, P)
    x = torch.from_numpy(O.make_images(batch, seed=0))
    with torch.no_grad():
        tw.decode(tw.encode(x))  # warm-up
        t0 = time.time(); n = 0
        while True:
            tw.decode(tw.encode(x)); n += 1
            if time.time() - t0 > budget_s or n >= 20:
                break
        dt = (time.time() - t0) / n
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": batch / dt, "unit": "reconstructions/s", "cores": cores, "kind": "port",
            "sample": "%d passes of batch %d through the torch-CPU restatement (not Theano), %s" % (n, batch, model)}


def pmc_traffic(arch, B):
    """HBM bytes per tapgemm launch from the committed rocprofv3 PMC passes of this same command (profiles/, made by
    scripts/profile_round.sh + scripts/summarize_profile.py; FETCH_SIZE x2-corrected per MI355X_MICROARCH.md).
    PMC counters cannot be read from inside the process, so the latest committed summary is quoted; None if absent."""
    import glob
    tag = "ian_simple_b64" if (arch, B) == ("IAN_simple", 64) else ("ian_b256" if (arch, B) == ("IAN", 256) else None)
    if tag is None:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % tag)))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            v = json.load(fh).get("tapgemm_traffic_bytes_per_launch")
        return (float(v) if v else None), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def train_step_bench(batch, rank, world, iters=3):
    """ms per update_gen / update_discrim (train_IAN.py:309-329) of the full IAN at `batch` images per GPU."""
    import torch
    from neural_photo_editor_amd.trainer import Trainer, Comm
    from neural_photo_editor_amd import synthetic as O
    P = O.make_train_params(O.make_params("IAN", 1))
    tr = Trainer(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), P, batch=batch, comm=Comm(), exact=True)
    rs = np.random.RandomState(50 + rank)
    X = torch.from_numpy(O.make_images(batch, seed=200 + rank)).cuda()
    Z = torch.from_numpy(rs.randn(batch, 100).astype(np.float32)).cuda()
    eps = torch.from_numpy(rs.randn(batch, 100).astype(np.float32)).cuda()
    out = {}
    for which in ("gen", "discrim"):
        tr.step(which, X, Z, eps, return_metrics=False)          # warm-up (schedules, workspaces)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            tr.step(which, X, Z, eps, return_metrics=False)
        torch.cuda.synchronize()
        out["update_%s_ms" % which] = (time.perf_counter() - t) / iters * 1e3
    pair = out["update_gen_ms"] + out["update_discrim_ms"]
    out.update({"images_per_s": 2 * batch * world / (pair * 1e-3), "per_gpu_batch": batch, "global_batch": batch * world,
                "parallelism": "dp%d, RCCL all-reduce of flat gradient groups, SyncBN statistics + MinibatchLayer all-gather (exact)" % world,
                "note": "one update_gen + one update_discrim (strict alternation, train_IAN.py:497-504) over synthetic data"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="IAN_simple", choices=["IAN_simple", "IAN"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 64 for IAN_simple, 256 for IAN)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-edit", action="store_true")
    ap.add_argument("--train", action="store_true", help="also time the train_IAN.py step (default: only on 1 GPU)")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-batch", type=int, default=128, help="per-GPU minibatch of the training step (config 5: 1024 over 8 GPUs)")
    ap.add_argument("--host-io", action="store_true", help="also time the API.py-style call: host numpy in, host numpy out (PCIe inclusive)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # IAN_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a single-GPU box (ranks share cuda:0)
    backend = os.environ.get("IAN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local % torch.cuda.device_count() if backend != "nccl" else local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)

    from neural_photo_editor_amd import IAN
    from neural_photo_editor_amd import synthetic as O   # seeded synthetic parameters / images (no trained weights exist)

    arch = args.arch
    B = args.batch or (64 if arch == "IAN_simple" else 256)
    P = O.make_params(arch, seed=1)
    model = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py"), True, params=P)
    h = model.handle
    for kv in filter(None, os.environ.get("IAN_OPTS", "").split(",")):  # tuning knobs, e.g. IAN_OPTS=tg_cfg=0
        k, v = kv.split("=")
        h.set_option(k, int(v))
    x = torch.from_numpy(O.make_images(B, seed=100 + rank)).cuda()
    out = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        h.call("ian_reconstruct", x, B, out, stream=stream)

    step()
    if not os.environ.get("IAN_NO_AUTOTUNE"):
        h.autotune(B, 1, stream=stream)  # untimed: pick tile shape / split-K per layer for this batch on this GPU
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    result = None
    if rank == 0:
        # --- roofline: profiled pass (HIP events around every tapgemm launch, on the launch stream) ---
        h.profile_enable(True)
        for _ in range(min(args.steps, 20)):
            step()
        pr = h.profile_read()
        h.profile_enable(False)
        launches = max(pr["tapgemm_launches"], 1)
        flops_per_launch = pr["tapgemm_flops"] / launches
        avg_ms = pr["tapgemm_ms"] / launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(arch, B)
        roofline = {"bound": "mfma", "kernel": "tapgemm_kernel (fp32 v_mfma_f32_32x32x2_f32)", "achieved": achieved,
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 PMC, %s)" % traffic_src if traffic else None,
                    "flop_per_launch": flops_per_launch, "launches_per_step": launches / min(args.steps, 20),
                    "avg_launch_ms": avg_ms, "tapgemm_share_of_step": pr["tapgemm_ms"] / max(pr["total_ms"], 1e-9),
                    "whole_step_tflops": FLOP_PER_RECON[arch] * B / (ms_per_step * 1e-3) / 1e12}
        edit = None
        if arch == "IAN_simple" and not args.no_edit:
            z = O.make_latents(1, seed=2)
            rgb = np.full((1, 3, 64, 64), -1.0, np.float32); rgb[:, 0] = 1.0
            c1, r1, c2, r2 = 26, 26, 30, 30
            model.imgradRGB(c1, r1, c2, r2, rgb, z)
            if not os.environ.get("IAN_NO_AUTOTUNE"):
                h.autotune(1, 3)
            lat = []
            for i in range(120):
                t = time.perf_counter()
                g = model.imgradRGB(c1, r1, c2, r2, rgb, z)
                z = z - 0.05 * g * (1 + (c2 - c1))
                model.sample_at(z)
                lat.append((time.perf_counter() - t) * 1e3)
            lat = np.array(lat[20:])
            os.environ["IAN_NO_DEC_CACHE"] = "1"   # every call recomputes the decoder forward (2 fwd + 1 bwd per step)
            lat2 = []
            for i in range(60):
                t = time.perf_counter()
                g = model.imgradRGB(c1, r1, c2, r2, rgb, z)
                z = z - 0.05 * g * (1 + (c2 - c1))
                model.sample_at(z)
                lat2.append((time.perf_counter() - t) * 1e3)
            del os.environ["IAN_NO_DEC_CACHE"]
            edit = {"p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)), "steps": len(lat),
                    "p50_ms_no_forward_cache": float(np.percentile(lat2[10:], 50)),
                    "update": "gradient descent (reference, NPE.py:199-209)",
                    "calls": "imgradRGB + sample_at through the API.py surface (host numpy in/out); imgradRGB reuses the "
                             "decoder activations sample_at left for the same latent",
                    "includes": "host<->device copies of z, rgb, image"}
        host_io = None
        if args.host_io:
            xh = x.cpu().numpy()
            model.reconstruct(xh)
            t = time.perf_counter()
            for _ in range(10):
                model.reconstruct(xh)
            host_io = {"value": 10 * B / (time.perf_counter() - t), "unit": "reconstructions/s",
                       "note": "host numpy in -> host numpy out per call (pageable memory, PCIe inclusive); never `value`"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(arch, P, B if arch == "IAN_simple" else 32)
        result = {
            "metric": "64x64 IAN reconstructions/sec", "value": value, "unit": "reconstructions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s encode->z->decode reconstruction, batch %d per GPU, inputs resident in HBM"
                                   % (arch, B), "parallelism": "replicas x%d (no data-path collective)" % world},
            "roofline": roofline, "cpu_baseline": cpu, "edit_step": edit,
        }
        if host_io:
            result["host_io"] = host_io
    # ---- train_IAN.py step (BASELINE.json configs[4]): full IAN, data parallel, RCCL gradient all-reduce ----------
    train = None
    if (args.train or world == 1) and not args.no_train:
        try:
            train = train_step_bench(args.train_batch, rank, world)
        except Exception as exc:  # never let the secondary measurement take the headline number down
            train = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rank == 0 and result is not None:
        result["train_step"] = train
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()

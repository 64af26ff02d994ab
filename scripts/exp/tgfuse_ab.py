"""Round-5 verdict, item 1(a): did the in-launch split-K combine epilogue that round 5 added to the small tap-GEMM tiles (tg_fuse;
+10 SGPRs, unused by default) cost anything?  A/B of the COMPILED CODE in ONE process on one box:
    A = libian.so         (HEAD)
    B = libian_nofuse.so  (same sources, -DIAN_NO_TG_FUSE: the epilogue compiled out of every tile = the round-4 object code of those tiles;
                           IAN_NOFUSE_BUILD=1 python -c "from neural_photo_editor_amd import build; build.build()")
Both models replay the SAME autotune choices (A tunes and writes IAN_TUNE_CACHE, B reads it), and the three workloads are timed
alternately A, B, A, B ...:
    step64   IAN_simple encode -> z -> decode at batch 64 on device buffers (HIP events, 100 steps per sample)
    b1_recon the same at batch 1 (HIP events, 200 calls per sample)
    brush    one ian_brush_step call per event through the Python surface (perf_counter p50 of 200 events per sample)
  python scripts/exp/tgfuse_ab.py  ->  gpurun_out/r06_tgfuse_ab.json"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neural_photo_editor_amd import IAN, synthetic as O  # noqa: E402
from neural_photo_editor_amd import lib as L  # noqa: E402

CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN_simple.py")
NOFUSE = os.environ.get("AB_LIB_B") or os.path.join(ROOT, "neural_photo_editor_amd", "libian_nofuse.so")   # AB_LIB_B: any other build of the library (e.g. the previous commit's)
LABEL_B = os.environ.get("AB_LABEL", "nofuse")
OUT_NAME = os.environ.get("AB_OUT", "r06_tgfuse_ab.json")
ROUNDS = int(os.environ.get("AB_ROUNDS", "5"))


def load_both():
    a = L.load_library()
    L._lib = None
    os.environ["IAN_LIB"] = NOFUSE
    b = L.load_library()
    del os.environ["IAN_LIB"]
    L._lib = a
    assert a is not b
    return a, b


def main():
    assert os.path.exists(NOFUSE), "build libian_nofuse.so first (IAN_NOFUSE_BUILD=1)"
    os.environ["IAN_TUNE_CACHE"] = os.path.join(tempfile.mkdtemp(), "tune.txt")
    libs = dict(zip("AB", load_both()))
    P = O.make_params("IAN_simple", 1)
    st = torch.cuda.current_stream().cuda_stream
    x64 = torch.from_numpy(O.make_images(64, seed=100)).cuda()
    o64 = torch.empty_like(x64)
    x1 = torch.from_numpy(O.make_images(1, seed=5)).cuda()
    o1 = torch.empty_like(x1)
    z0 = O.make_latents(1, seed=2)
    rgb = np.full((1, 3, 64, 64), -1.0, np.float32)
    rgb[:, 0] = 1.0
    models = {}
    for k in "AB":                                   # A first: it tunes and writes the cache B replays
        L._lib = libs[k]
        m = IAN(CFG, True, params=P)
        h = m.handle
        h.call("ian_reconstruct", x64, 64, o64, stream=st)
        h.autotune(64, 1, stream=st)
        h.call("ian_reconstruct", x1, 1, o1, stream=st)
        h.autotune(1, 1, stream=st)
        m.imgradRGB(26, 26, 30, 30, rgb, z0)
        h.autotune(1, 3)
        models[k] = m
    L._lib = libs["A"]
    outs = {}
    for k in "AB":                                    # same results? (same schedules, same MFMA order: must be bitwise)
        models[k].handle.call("ian_reconstruct", x64, 64, o64, stream=st)
        torch.cuda.synchronize()
        outs[k] = o64.clone()
    same = bool(torch.equal(outs["A"], outs["B"]))

    def dev_ms(h, x, n, o, reps):
        for _ in range(10):
            h.call("ian_reconstruct", x, n, o, stream=st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            h.call("ian_reconstruct", x, n, o, stream=st)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def brush_ms(m):
        z = z0.copy()
        lat = []
        for _ in range(240):
            t = time.perf_counter()
            z, _ = m.brush_step(26, 26, 30, 30, z, RGB=rgb, weight=0.05)
            lat.append((time.perf_counter() - t) * 1e3)
        return float(np.percentile(lat[40:], 50))

    res = {k: {"step64_ms": [], "b1_recon_ms": [], "brush_p50_ms": []} for k in "AB"}
    for r in range(ROUNDS):
        for k in ("AB" if r % 2 == 0 else "BA"):
            h = models[k].handle
            res[k]["step64_ms"].append(dev_ms(h, x64, 64, o64, 100))
            res[k]["b1_recon_ms"].append(dev_ms(h, x1, 1, o1, 200))
            res[k]["brush_p50_ms"].append(brush_ms(models[k]))
    out = {"what": __doc__.split("\n\n")[0] if False else ("in-process A/B of libian.so (A, HEAD) vs libian_nofuse.so (B, -DIAN_NO_TG_FUSE: round-4 small-tile object code); same autotune cache; alternated" if LABEL_B == "nofuse" else "in-process A/B of libian.so (A, HEAD) vs %s (B, %s); same autotune cache; alternated" % (os.path.basename(NOFUSE), LABEL_B)),
           "rounds": ROUNDS, "bitwise_equal_batch64_output": same, "samples": res, "median": {}, "B_over_A": {}}
    for w in ("step64_ms", "b1_recon_ms", "brush_p50_ms"):
        ma, mb = float(np.median(res["A"][w])), float(np.median(res["B"][w]))
        out["median"][w] = {"A_head": ma, "B_" + LABEL_B: mb}
        out["B_over_A"][w] = mb / ma
    out["verdict"] = "HEAD costs %.2f %% at batch 64, %.2f %% on the batch-1 reconstruction, %.2f %% on the brush event (positive = HEAD slower than B)" % tuple(
        100.0 * (1.0 / out["B_over_A"][w] - 1.0) for w in ("step64_ms", "b1_recon_ms", "brush_p50_ms"))
    try:
        out["box"] = L.box_probe(900, 250, stream=st)
    except Exception as exc:  # noqa: BLE001
        out["box"] = {"error": str(exc)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", OUT_NAME), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("bitwise_equal_batch64_output", "median", "B_over_A", "verdict", "box")}, indent=1))


if __name__ == "__main__":
    main()

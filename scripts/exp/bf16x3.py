"""The opt-in split-bf16 tap-GEMM (ian_set_option("tg_bf16x3", 1); kernels_tapgemm.hip tapgemm_bf16x3_kernel) against the exact-fp32
path, in ONE process: ms per reconstruction step (HIP events, alternated), per-launch tap-GEMM time from the library's own profile
events, and the error against the exact-fp32 output.  Labelled secondary: never the bench `value`.
  python scripts/exp/bf16x3.py [IAN_simple|IAN] [batch]  ->  gpurun_out/r06_bf16x3_<arch>.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neural_photo_editor_amd import IAN, synthetic as O  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "IAN_simple"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if arch == "IAN_simple" else 256)
FLOP = {"IAN_simple": 2.592e9, "IAN": 7.907e9}[arch]
cfg = os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py")
P = O.make_params(arch, 1)
st = torch.cuda.current_stream().cuda_stream
x = torch.from_numpy(O.make_images(B, seed=100)).cuda()
models, outs = {}, {}
for name, on in (("fp32", 0), ("bf16x3", 1)):
    m = IAN(cfg, True, params=P)
    h = m.handle
    if on:
        h.set_option("tg_bf16x3", 1)
        for kv in filter(None, os.environ.get("BF_OPTS", "").split(",")):
            k, v = kv.split("=")
            h.set_option(k, int(v))
    o = torch.empty_like(x)
    h.call("ian_reconstruct", x, B, o, stream=st)
    h.autotune(B, 1, stream=st)
    h.call("ian_reconstruct", x, B, o, stream=st)
    torch.cuda.synchronize()
    models[name], outs[name] = m, o


def ms(h, o, reps=50):
    for _ in range(5):
        h.call("ian_reconstruct", x, B, o, stream=st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        h.call("ian_reconstruct", x, B, o, stream=st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {"fp32": [], "bf16x3": []}
for r in range(4):
    for name in (("fp32", "bf16x3") if r % 2 == 0 else ("bf16x3", "fp32")):
        res[name].append(ms(models[name].handle, outs[name]))
prof = {}
for name in res:
    h = models[name].handle
    h.profile_enable(True)
    for _ in range(10):
        h.call("ian_reconstruct", x, B, outs[name], stream=st)
    pr = h.profile_read()
    h.profile_enable(False)
    prof[name] = {"tapgemm_avg_launch_us": pr["tapgemm_ms"] / max(pr["tapgemm_launches"], 1) * 1e3,
                  "tapgemm_tflops_fp32_equivalent": pr["tapgemm_flops"] / max(pr["tapgemm_ms"], 1e-9) / 1e9,
                  "tapgemm_share_of_step": pr["tapgemm_ms"] / max(pr["total_ms"], 1e-9)}
a, b = outs["bf16x3"].cpu().numpy().astype(np.float64), outs["fp32"].cpu().numpy().astype(np.float64)
out = {"arch": arch, "batch": B, "ms_per_step": {k: float(np.median(v)) for k, v in res.items()}, "samples_ms": res,
       "recon_per_s": {k: B / (float(np.median(v)) * 1e-3) for k, v in res.items()},
       "speedup": float(np.median(res["fp32"]) / np.median(res["bf16x3"])), "profile": prof,
       "whole_step_fp32_equivalent_tflops": {k: FLOP * B / (float(np.median(v)) * 1e-3) / 1e12 for k, v in res.items()},
       "max_rel_err_vs_fp32_hip": float(np.abs(a - b).max() / np.abs(b).max()),
       "bf16_dense_peak_tflops": 2500.0, "flops_executed_bf16_per_fp32_flop": 3}
out["bf16x3_frac_of_bf16_peak"] = 3 * out["profile"]["bf16x3"]["tapgemm_tflops_fp32_equivalent"] / 2500.0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_bf16x3_%s.json" % arch), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("ms_per_step", "speedup", "profile", "max_rel_err_vs_fp32_hip", "bf16x3_frac_of_bf16_peak")}, indent=1))

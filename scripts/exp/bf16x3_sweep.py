"""Schedules / weight pre-split of the opt-in split-bf16 tap-GEMM, one process: ms per IAN_simple batch-64 step (and full IAN batch 256 with
`IAN`), error against the exact-fp32 path.  python scripts/exp/bf16x3_sweep.py [arch] [batch] -> gpurun_out/r06_bf16x3_sweep_<arch>.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neural_photo_editor_amd import IAN, synthetic as O  # noqa: E402
arch = sys.argv[1] if len(sys.argv) > 1 else "IAN_simple"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if arch == "IAN_simple" else 256)
cfg = os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py")
P = O.make_params(arch, 1)
st = torch.cuda.current_stream().cuda_stream
x = torch.from_numpy(O.make_images(B, seed=100)).cuda()
CONFIGS = [("fp32", {})] + [("bf16x3 wsplit=%d sched=%d" % (w, s), {"tg_bf16x3": 1, "tg_bf16x3_wsplit": w, "tg_bf16x3_sched": s})
                            for w in (1, 0) for s in (0, 1, 2)]
if os.environ.get("SWEEP_ONLY"):
    keep = os.environ["SWEEP_ONLY"].split(";")
    CONFIGS = [c for c in CONFIGS if c[0] == "fp32" or any(k in c[0] for k in keep)]
models = {}
for name, opts in CONFIGS:
    m = IAN(cfg, True, params=P)
    for k, v in opts.items():
        m.handle.set_option(k, v)
    o = torch.empty_like(x)
    m.handle.call("ian_reconstruct", x, B, o, stream=st)
    m.handle.autotune(B, 1, stream=st)
    m.handle.call("ian_reconstruct", x, B, o, stream=st)
    torch.cuda.synchronize()
    models[name] = (m, o)


def ms(h, o, reps=40):
    for _ in range(5):
        h.call("ian_reconstruct", x, B, o, stream=st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        h.call("ian_reconstruct", x, B, o, stream=st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {n: [] for n, _ in CONFIGS}
for r in range(3):
    for n, _ in (CONFIGS if r % 2 == 0 else CONFIGS[::-1]):
        res[n].append(ms(models[n][0].handle, models[n][1]))
ref = models["fp32"][1].cpu().numpy().astype(np.float64)
out = {"arch": arch, "batch": B, "configs": {}}
for n, _ in CONFIGS:
    h = models[n][0].handle
    h.profile_enable(True)
    for _ in range(10):
        h.call("ian_reconstruct", x, B, models[n][1], stream=st)
    pr = h.profile_read()
    h.profile_enable(False)
    a = models[n][1].cpu().numpy().astype(np.float64)
    out["configs"][n] = {"ms_per_step": float(np.median(res[n])), "samples": res[n],
                         "tapgemm_avg_launch_us": pr["tapgemm_ms"] / max(pr["tapgemm_launches"], 1) * 1e3,
                         "max_rel_err_vs_fp32": float(np.abs(a - ref).max() / np.abs(ref).max())}
    print("%-28s %.4f ms/step   tapgemm %.1f us/launch   err %.2e" % (n, out["configs"][n]["ms_per_step"], out["configs"][n]["tapgemm_avg_launch_us"],
                                                                      out["configs"][n]["max_rel_err_vs_fp32"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_bf16x3_sweep_%s.json" % arch), "w"), indent=1)

"""Timing-only ablations of tapgemm's production K loop (schedule 7) on the batch-64 step with the AUTOTUNED tiles / splits
(libian_ablation.so): every tapgemm launch forced to schedule 7 / 2 / 17 (no loads + LDS stores) / 18 (no barrier) / 19 (no fragment
reads) / 20 (17 + 18) / 21 (MFMAs and control flow only).  Results of 17..21 are WRONG by construction; only the time counts.
  IAN_LIB=neural_photo_editor_amd/libian_ablation.so python scripts/exp/tg_ablate7.py [arch] [batch]  ->  gpurun_out/r06_tg_ablate7.json"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from neural_photo_editor_amd import IAN, synthetic as O
arch = sys.argv[1] if len(sys.argv) > 1 else "IAN_simple"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py"), True, params=O.make_params(arch, 1))
h = m.handle
x = torch.from_numpy(O.make_images(B, seed=100)).cuda()
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
step = lambda: h.call("ian_reconstruct", x, B, out, stream=st)
step(); h.autotune(B, 1, stream=st)
res = {}
VARS = [7, 2, 17, 18, 19, 20, 21, 22, 23]
for rep in range(3):
    for v in VARS:
        h.set_option("tg_variant_force", v)
        for _ in range(10): step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): step()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) / 100)
h.set_option("tg_variant_force", -1)
for rep in range(3):   # the production schedule without anything behind the K loop (no epilogue arithmetic, no stores; reduce launches still run)
    h.set_option("tg_noepi", 1)
    for _ in range(10): step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): step()
    e1.record(); torch.cuda.synchronize()
    res.setdefault(99, []).append(e0.elapsed_time(e1) / 100)
h.set_option("tg_noepi", 0)
VARS = VARS + [99]
names = {99: "schedule 7, tapgemm returns after its K loop", 7: "schedule 7 (production)", 2: "schedule 2 (round 5)", 17: "no global loads / LDS stores", 18: "no barrier", 19: "no fragment reads", 20: "no loads, no barrier", 21: "MFMAs + control flow only", 22: "loads issued and awaited, no LDS stores", 23: "LDS stores only (no loads)"}
outd = {"arch": arch, "batch": B, "ms_per_step": {names[v]: float(np.median(res[v])) for v in VARS}, "runs": {names[v]: res[v] for v in VARS}}
for v in VARS: print("%-32s %s  median %.4f ms" % (names[v], " ".join("%.4f" % t for t in res[v]), float(np.median(res[v]))))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(outd, open(os.path.join(ROOT, "gpurun_out", "r06_tg_ablate7.json"), "w"), indent=1)

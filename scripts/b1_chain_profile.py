"""Batch-1 chains under rocprofv3 (kernel trace / PMC passes): 100 brush events (ian_brush_step, one call each, graph replay) and
100 batch-1 reconstructions (ian_reconstruct on device buffers), each phase bracketed by marker launches (torch.arange) so that
scripts/summarize_b1_profile.py can cut the dispatch list: HBM bytes per brush event and per reconstruction for
bench.py's edit_step.roofline.traffic / b1_recon.roofline.traffic (round-4 verdict, weak #9)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_photo_editor_amd import IAN, synthetic as O
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(os.environ.get("REPS", "100"))
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN_simple.py"), True, params=O.make_params("IAN_simple", 1))
z = O.make_latents(1, seed=2)
rgb = np.full((1, 3, 64, 64), -1.0, np.float32); rgb[:, 0] = 1.0
x1 = O.make_images(1, seed=0)
m.reconstruct(x1)
m.imgradRGB(26, 26, 30, 30, rgb, z)
m.handle.autotune(1, 3)
for _ in range(5):
    z, _ = m.brush_step(26, 26, 30, 30, z, RGB=rgb, weight=0.05)          # eager pass + capture
torch.cuda.synchronize()
torch.arange(7001, device="cuda"); torch.cuda.synchronize()               # marker 1
for _ in range(N):
    z, _ = m.brush_step(26, 26, 30, 30, z, RGB=rgb, weight=0.05)
torch.cuda.synchronize()
torch.arange(7002, device="cuda"); torch.cuda.synchronize()               # marker 2
h = m.handle
xd = torch.from_numpy(x1).cuda(); od = torch.empty_like(xd)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    h.call("ian_reconstruct", xd, 1, od, stream=st)
torch.cuda.synchronize()
torch.arange(7003, device="cuda"); torch.cuda.synchronize()               # marker 3
for _ in range(N):
    h.call("ian_reconstruct", xd, 1, od, stream=st)
torch.cuda.synchronize()
torch.arange(7004, device="cuda"); torch.cuda.synchronize()               # marker 4
print("done", N)

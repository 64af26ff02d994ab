"""Run under rocprofv3 on the GPU box: a few training steps of the full IAN at the per-GPU batch of config 5."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_photo_editor_amd import synthetic as O
from neural_photo_editor_amd.trainer import Trainer
B = int(os.environ.get("B", "128"))
P = O.make_train_params(O.make_params("IAN", 1))
tr = Trainer(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), P, B)     # ian_train_step (C entry)
rs = np.random.RandomState(0)
X = torch.from_numpy(O.make_images(B, seed=1)).cuda()
Z = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
eps = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
if not os.environ.get("IAN_NO_AUTOTUNE"):
    tr.autotune()                                    # as bench.py does before timing the step
for it in range(2):                                  # warm-up: schedules, workspaces, the all-reduce plan
    tr.step("gen" if it % 2 == 0 else "discrim", X, Z, eps, return_metrics=False)
torch.cuda.synchronize()
torch.arange(7777, device="cuda")                    # marker launch: scripts/summarize_train_profile.py keeps what follows it
for it in range(int(os.environ.get("ITERS", "4"))):
    tr.step("gen" if it % 2 == 0 else "discrim", X, Z, eps, return_metrics=False)
torch.cuda.synchronize()
print("done")

"""ms per (update_gen + update_discrim) pair of the full-IAN training step at 128 images (ian_train_step), one process, after the
layer autotune: the quick A/B companion of bench.py's train_step leg (options through IAN_OPTS, which every layer object reads
when it is created).   python scripts/exp/train_pair_ms.py [batch] [pairs]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from neural_photo_editor_amd import synthetic as O
from neural_photo_editor_amd.trainer import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tr = Trainer(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), O.make_train_params(O.make_params("IAN", 1)), B)
rs = np.random.RandomState(0)
X = torch.from_numpy(O.make_images(B, seed=1)).cuda()
Z = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
eps = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
if not os.environ.get("IAN_NO_AUTOTUNE"):
    tr.autotune()
for w in ("gen", "discrim", "gen", "discrim"):
    tr.step(w, X, Z, eps, return_metrics=False)
torch.cuda.synchronize()
res = {}
for w in ("gen", "discrim"):
    t = time.perf_counter()
    for _ in range(PAIRS):
        tr.step(w, X, Z, eps, return_metrics=False)
    torch.cuda.synchronize()
    res[w] = (time.perf_counter() - t) / PAIRS * 1e3
flop = {"gen": 46159200000.0, "discrim": 32952600000.0}
pair = res["gen"] + res["discrim"]
print("IAN_OPTS=%s  update_gen %.2f ms  update_discrim %.2f ms  pair %.2f ms  -> %.0f images/s, %.4f of the fp32-MFMA peak" % (
    os.environ.get("IAN_OPTS", ""), res["gen"], res["discrim"], pair, 2 * B / (pair * 1e-3), B * (flop["gen"] + flop["discrim"]) / (pair * 1e-3) / 157.3e12))

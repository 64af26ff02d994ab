import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ian_oracle as O
from oracle.torch_twin import TorchTwin
from neural_photo_editor_amd import IAN
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(32)
def rel(a,b): return float(np.abs(np.asarray(a,np.float64)-np.asarray(b,np.float64)).max()/(np.abs(np.asarray(b)).max()+1e-30))
P = O.make_params("IAN", 1)
z = O.make_latents(3, seed=21)
rgb = np.random.RandomState(5).uniform(-1, 1, (1, 3, 64, 64)).astype(np.float32)
for flip in (False, True):
    tw = TorchTwin("IAN", P, deconv_flip=flip, dtype=torch.float64)
    ref = {p: tw.imgrad(*p, z[:1]) for p in ((0, 0, 64, 64), (0, 0, 64, 32), (0, 32, 64, 64), (16, 16, 48, 48))}
    for opts in ({}, {"tg_reduce_kp": 1}, {"tg_split": 0}, {"mdc_head": 0}, {"edit_graph": 0}):
        m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py"), True, params=P, deconv_flip=flip)
        for k, v in opts.items():
            m.handle.set_option(k, v)
        print("flip", flip, opts, {str(p): "%.2e" % rel(m.imgrad(*p, z[:1]), r) for p, r in ref.items()}, flush=True)
        m.close()

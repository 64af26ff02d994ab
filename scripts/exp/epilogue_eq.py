import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from neural_photo_editor_amd import IAN, synthetic as O
for arch, B in (("IAN_simple", 5), ("IAN", 3)):
    m = IAN(os.path.join("neural_photo_editor_amd", "configs", arch + ".py"), True, params=O.make_params(arch, 1))
    x = O.make_images(B, seed=3)
    outs = []
    for v in (0, 1):
        m.handle.set_option("tg_fast_epilogue", v)
        outs.append(m.reconstruct(x))
    z = np.random.RandomState(0).randn(1, 100).astype(np.float32)
    gs = []
    for v in (0, 1):
        m.handle.set_option("tg_fast_epilogue", v)
        gs.append(m.imgradRGB(10, 12, 30, 28, np.zeros((1, 3, 64, 64), np.float32) + 0.3, z))
    print(arch, "recon bitwise equal:", np.array_equal(outs[0], outs[1]), float(np.abs(outs[0] - outs[1]).max()), "| brush grad equal:", np.array_equal(gs[0], gs[1]), float(np.abs(gs[0] - gs[1]).max()))

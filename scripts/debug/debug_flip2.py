import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ian_oracle as O
from neural_photo_editor_amd import IAN
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def rel(a,b): return float(np.abs(a.astype(np.float64)-b.astype(np.float64)).max()/(np.abs(b).max()+1e-30))
P = O.make_params("IAN", 1)
z = O.make_latents(3, seed=21)
cfgp = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py")
def run(opts):
    m = IAN(cfgp, True, params=P, deconv_flip=False)
    for k, v in opts.items():
        m.handle.set_option(k, v)
    g = m.imgrad(0, 32, 64, 64, z[:1])
    out = {"dz": g}
    low = m.lowered
    for op in low.ops:
        if op.segment != 2:
            continue
        for sl in (op.src, op.dst):
            name = "slot%d" % sl
            if name in out:
                continue
            try:
                out[name] = m.handle.read_slot_grad(sl, 1)
            except Exception as e:
                out[name] = None
    info = [(op.name, op.kind, op.src, op.src2, op.dst) for op in low.ops if op.segment == 2]
    m.close()
    return out, info
good, info = run({"tg_reduce_kp": 1})
for opts in ({}, {"tg_split": 0}):
    bad, _ = run(opts)
    print("==== ", opts, "dz rel diff vs kp=1:", rel(bad["dz"], good["dz"]))
    for (name, kind, src, src2, dst) in info:
        a, b = bad.get("slot%d" % src), good.get("slot%d" % src)
        if a is None or b is None:
            print("  %-14s kind %d src slot %d: n/a" % (name, kind, src)); continue
        print("  %-14s kind %d  grad of src slot %2d (dst %2d): rel diff %.2e  max|g| %.3e" % (name, kind, src, dst, rel(a, b), np.abs(b).max()))

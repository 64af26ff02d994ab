"""Experiment: does overlapping the fill/drain of consecutive layer launches help?  One B=64 reconstruction per step on
one stream vs two B=32 halves on two streams (two handles)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_photo_editor_amd import IAN, synthetic as O
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
arch = os.environ.get("ARCH", "IAN_simple")
cfg = os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py")
P = O.make_params(arch, seed=1)

def make(B, stream):
    m = IAN(cfg, True, params=P)
    x = torch.from_numpy(O.make_images(B, seed=100)).cuda()
    out = torch.empty_like(x)
    f = lambda: m.handle.call("ian_reconstruct", x, B, out, stream=stream.cuda_stream)
    f(); m.handle.autotune(B, 1, stream=stream.cuda_stream)
    return m, f

def timeit(fs, steps=300):
    for f in fs: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for f in fs: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in (int(os.environ.get("B", "64")),):
    m0, f0 = make(B, s1)
    t = timeit([f0]); print("one stream  B=%d: %.3f ms/step -> %.0f recon/s" % (B, t, B / t * 1e3))
    ma, fa = make(B // 2, s1); mb, fb = make(B // 2, s2)
    t = timeit([fa]); print("one stream  B=%d: %.3f ms/step -> %.0f recon/s" % (B // 2, t, B / 2 / t * 1e3))
    t = timeit([fa, fb]); print("two streams 2xB=%d: %.3f ms/step -> %.0f recon/s" % (B // 2, t, B / t * 1e3))
    mc, fc = make(B, s2)
    t = timeit([f0, fc]); print("two streams 2xB=%d: %.3f ms/step -> %.0f recon/s" % (B, t, 2 * B / t * 1e3))

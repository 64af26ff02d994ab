"""Where the host side of a brush event goes: api.IAN.brush_step (numpy in / out, argument checks) vs the bare ctypes call of
ian_brush_step with preallocated buffers vs the device timeline (profiles/r05_batch1_chains.md).  p50 over 300 events each."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_photo_editor_amd import IAN, synthetic as O
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN_simple.py"), True, params=O.make_params("IAN_simple", 1))
z = O.make_latents(1, seed=2)
rgb = np.full((1, 3, 64, 64), -1.0, np.float32); rgb[:, 0] = 1.0
m.reconstruct(O.make_images(1, seed=0)); m.imgradRGB(26, 26, 30, 30, rgb, z); m.handle.autotune(1, 3)
def p50(f, n=300):
    lat = []
    for _ in range(n):
        t = time.perf_counter(); f(); lat.append((time.perf_counter() - t) * 1e3)
    return float(np.percentile(lat[50:], 50))
state = {"z": z}
def api():
    state["z"], _ = m.brush_step(26, 26, 30, 30, state["z"], RGB=rgb, weight=0.05)
a = p50(api)
h = m.handle
zin = np.ascontiguousarray(state["z"][:1]); zout = np.empty((1, 100), np.float32); x = np.empty((1, 3, 64, 64), np.float32)
pz, pzo, px, prgb = [C.c_void_p(v.ctypes.data) for v in (zin, zout, x, rgb)]
null = C.c_void_p(0)
fn = h.lib.ian_brush_step
def bare():
    rc = fn(h._h, 26, 26, 30, 30, prgb, pz, -0.05, 5.0, pzo, null, px, None, null)
    assert rc == 0
    zin[:] = zout
b = p50(bare)
def bare_noimg():
    rc = fn(h._h, 26, 26, 30, 30, prgb, pz, -0.05, 5.0, pzo, null, null, None, null)
    assert rc == 0
    zin[:] = zout
c = p50(bare_noimg)
print("api.brush_step p50 %.4f ms | bare ctypes ian_brush_step %.4f ms | bare, no image copy-out %.4f ms" % (a, b, c))

"""p50 latency of one NPE brush event (ian_brush_step: gradient + latent update + decoder, one call) -- quick A/B helper.
usage (GPU box): python scripts/edit_latency.py [key=value ...]   e.g. b1_conv=0"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_photo_editor_amd import IAN, synthetic as O
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN_simple.py"), True, params=O.make_params("IAN_simple", 1))
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    m.handle.set_option(k, int(v))
z = O.make_latents(1, seed=2)
rgb = np.full((1, 3, 64, 64), -1.0, np.float32); rgb[:, 0] = 1.0
m.reconstruct(O.make_images(1, seed=0))
m.imgradRGB(26, 26, 30, 30, rgb, z)
m.handle.autotune(1, 3)
lat = []
for i in range(200):
    t = time.perf_counter()
    z, _ = m.brush_step(26, 26, 30, 30, z, RGB=rgb, weight=0.05)
    lat.append((time.perf_counter() - t) * 1e3)
lat = np.array(lat[50:])
print("brush_step one call: p50 %.4f ms  p95 %.4f ms  min %.4f ms  (%s)" % (np.percentile(lat, 50), np.percentile(lat, 95), lat.min(), " ".join(sys.argv[1:]) or "defaults"))

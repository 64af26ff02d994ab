"""Edit loop for a rocprofv3 kernel trace: 60 steps of imgradRGB + sample_at (IAN_simple, batch 1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_photo_editor_amd import IAN, synthetic as O
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN_simple.py"), True, params=O.make_params("IAN_simple", 1))
if os.environ.get("EDIT_EAGER"):
    m.handle.set_option("edit_graph", 0)
z = O.make_latents(1, seed=2)
rgb = np.full((1, 3, 64, 64), -1.0, np.float32); rgb[:, 0] = 1.0
m.reconstruct(O.make_images(1, seed=0))
m.imgradRGB(26, 26, 30, 30, rgb, z)
m.handle.autotune(1, 3)
for i in range(60):
    g = m.imgradRGB(26, 26, 30, 30, rgb, z)
    z = z - 0.05 * g * 5
    m.sample_at(z)

"""Batch-1 reconstruction (encode_images -> sample_at on device buffers, IAN_simple) for a rocprofv3 kernel trace:
200 back-to-back ian_reconstruct calls after the batch-1 autotune (encoder B=1 137 MB / decoder B=1 77 MB)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_photo_editor_amd import IAN, synthetic as O
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN_simple.py"), True, params=O.make_params("IAN_simple", 1))
h = m.handle
x = torch.from_numpy(O.make_images(1, seed=5)).cuda()
o = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
h.call("ian_reconstruct", x, 1, o, stream=st)
h.autotune(1, 1, stream=st)
for _ in range(int(os.environ.get("REPS", "200"))):
    h.call("ian_reconstruct", x, 1, o, stream=st)
torch.cuda.synchronize()

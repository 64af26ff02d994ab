"""Within-process A/B of one runtime option on the batch-64 reconstruction step (same handle, same autotune choices):
python scripts/ab_step.py key=valueA,valueB [arch] [batch]   -> ms per step for A and B, alternated 4 times."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_photo_editor_amd import IAN, synthetic as O
key, vals = sys.argv[1].split("=")
va, vb = [int(v) for v in vals.split(",")]
arch = sys.argv[2] if len(sys.argv) > 2 else "IAN_simple"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py"), True, params=O.make_params(arch, 1))
h = m.handle
x = torch.from_numpy(O.make_images(B, seed=100)).cuda()
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
step = lambda: h.call("ian_reconstruct", x, B, out, stream=st)
step(); h.autotune(B, 1, stream=st)
res = {va: [], vb: []}
for rep in range(4):
    for v in (va, vb):
        h.set_option(key, v)
        for _ in range(10): step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): step()
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 100)
for v in (va, vb):
    print("%s=%d: %s ms/step (median %.4f)" % (key, v, " ".join("%.4f" % t for t in res[v]), float(np.median(res[v]))))

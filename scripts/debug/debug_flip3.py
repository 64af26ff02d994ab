import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ian_oracle as O
from neural_photo_editor_amd import IAN
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = O.make_params("IAN", 1)
z = O.make_latents(3, seed=21)
cfgp = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py")
def run(opts, patch=(0, 32, 64, 64)):
    m = IAN(cfgp, True, params=P, deconv_flip=False)
    for k, v in opts.items():
        m.handle.set_option(k, v)
    m.imgrad(*patch, z[:1])
    g = {s: m.handle.read_slot_grad(s, 1)[0] for s in (21, 22, 23, 25)}
    m.imgrad(*patch, z[:1])
    g2 = m.handle.read_slot_grad(21, 1)[0]
    m.close()
    return g, g2
A, A2 = run({"tg_reduce_kp": 1})
B, B2 = run({})
C, C2 = run({"tg_split": 0})
print("repeat-call equal: kp1", np.array_equal(A[21], A2), "default", np.array_equal(B[21], B2), "nosplit", np.array_equal(C[21], C2))
for s in (22, 23, 25):
    print("slot", s, "inputs equal A/B:", np.array_equal(A[s], B[s]), "A/C:", np.array_equal(A[s], C[s]), "nan", np.isnan(A[s]).sum(), "absmax", np.abs(A[s]).max())
for name, X in (("default", B), ("nosplit", C)):
    d = np.abs(X[21] - A[21])          # (128, 64, 64)
    ref = np.abs(A[21]).max()
    print(name, "max diff/ref", d.max() / ref)
    print("  per 32-channel block:", [float("%.2e" % (d[c:c + 32].max() / ref)) for c in range(0, 128, 32)])
    print("  per 8-row block     :", [float("%.2e" % (d[:, r:r + 8].max() / ref)) for r in range(0, 64, 8)])
    print("  per 8-col block     :", [float("%.2e" % (d[:, :, c:c + 8].max() / ref)) for c in range(0, 64, 8)])
    bad = np.argwhere(d > 0.01 * ref)
    print("  n bad", len(bad), "first", bad[:6].tolist())
    if len(bad):
        c, y, x = bad[0]
        print("   values", X[21][c, y, x], A[21][c, y, x], " ratio", X[21][c, y, x] / A[21][c, y, x])

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from neural_photo_editor_amd import IAN
from oracle import ian_oracle as O
from oracle.torch_twin import TorchTwin
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
arch = "IAN_simple"
P = O.make_params(arch, 1)
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd/configs/IAN_simple.py"), True, params=P)
tw = TorchTwin(arch, P); orc = O.Oracle(arch, P)
z1 = O.make_latents(1)
# forward at n=1
xh = m.sample_at(z1)
for name, a in orc.decoder_activations(z1):
    nm = {"dec_fc2": "l_dec_fc2"}.get(name, name)
    print("fwd n=1", name, rel(m.activation(nm, 1), a))
# twin with retained pre-affine accumulators
z = torch.tensor(z1, requires_grad=True)
pre = []
a = z @ tw.P["l_dec_fc2.W"]; a.retain_grad(); pre.append(("l_dec_fc2", a))
h = torch.relu(tw.bn(a, "bnorm_dec_fc2")).reshape(-1, 1024, 4, 4)
for i in (1, 2, 3):
    a = tw.deconv(h, "dec_conv%d" % i); a.retain_grad(); pre.append(("dec_conv%d" % i, a))
    h = torch.relu(tw.bn(a, "bnorm_dc%d" % i))
a = tw.deconv(h, "dec_out"); a.retain_grad(); pre.append(("dec_out", a))
xo = torch.tanh(a)
r1, r2, c1, c2 = 26, 30, 26, 30
loss = xo[0, :, r1:r2, c1:c2].mean(); loss.backward()
g = m.imgrad(c1, r1, c2, r2, z1)
print("dz", rel(g, z.grad.numpy()))
for name, a in pre[:-1]:
    got = m.handle.read_slot_grad(m.lowered.slot_by_name(name), 1)
    ref = a.grad.numpy().reshape(got.shape)
    print("grad pre", name, rel(got, ref), float(np.abs(ref).max()), float(np.abs(got).max()))

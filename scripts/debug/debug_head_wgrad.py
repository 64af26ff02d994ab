"""Debug: few-filter MDCL backward-weight (mdc_head_wgrad_kernel) vs float64 autograd across batch / extent."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_photo_editor_amd.lib import load_train_library
from neural_photo_editor_amd import trainer as T
lib = load_train_library()
cs = lambda c: (c + 31) // 32 * 32
def to_nhwc(x):
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cs(c)); out[..., :c] = torch.from_numpy(x).permute(0, 2, 3, 1); return out.cuda()
for (n, h) in ((8, 16), (8, 64), (32, 64), (64, 64), (128, 64), (128, 16)):
    cin, cout, sc = 128, 2, [2, 3, 4]
    rs = np.random.RandomState(n * 100 + h)
    x = rs.randn(n, cin, h, h).astype(np.float32)
    W = (rs.randn(cout, cin, 3, 3) * 0.05).astype(np.float32)
    coeffs = [rs.uniform(0.5, 1.5, cout).astype(np.float32) for _ in range(1 + len(sc))]
    xt = torch.tensor(x, dtype=torch.float64)
    params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)] + [torch.tensor(c, dtype=torch.float64, requires_grad=True) for c in coeffs]
    y = F.conv2d(xt, params[0], padding=1) * params[1].reshape(1, -1, 1, 1)
    for i, s in enumerate(sc):
        y = y + F.conv2d(xt, params[0], padding=s, dilation=s) * params[2 + i].reshape(1, -1, 1, 1)
    dy = rs.randn(*y.shape).astype(np.float32)
    gp = torch.autograd.grad(y, params, torch.tensor(dy, dtype=torch.float64))
    layer = T.Layer(lib, T.K_MDC, cin, cout, h, h, scales=sc)
    layer.set_params([torch.from_numpy(p.detach().numpy().astype(np.float32).ravel()).cuda() for p in params])
    dp = [torch.zeros(int(np.prod(p.shape)), device="cuda") for p in params]
    layer.backward_weight(to_nhwc(x), to_nhwc(dy), n, dp)
    torch.cuda.synchronize()
    out = []
    for got, ref in zip(dp, gp):
        g, r = got.cpu().numpy().reshape(ref.shape).astype(np.float64), ref.numpy()
        out.append("err %.2e ratio %.3f nan %d" % (np.abs(g - r).max() / np.abs(r).max(), float((g * r).sum() / (r * r).sum()), int(np.isnan(g).sum())))
    print(n, h, " | ".join(out), flush=True)
    layer.close()

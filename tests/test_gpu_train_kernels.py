"""GPU unit parity of the training building blocks (include/ian_train.h) against float64 torch-CPU autograd:
ian_layer_{forward,backward_data,backward_weight} for every layer kind, and the batch-norm / MinibatchLayer /
IAF / loss / regulariser / Adam kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


def cs(c):
    return (c + 31) // 32 * 32


def to_nhwc(x, stride=None):
    """(n,c,h,w) numpy -> cuda tensor (n,h,w,stride) zero padded"""
    n, c, h, w = x.shape
    s = stride or cs(c)
    out = np.zeros((n, h, w, s), np.float32)
    out[..., :c] = x.transpose(0, 2, 3, 1)
    return torch.from_numpy(out).cuda()


def from_nhwc(t, c):
    return t.cpu().numpy()[..., :c].transpose(0, 3, 1, 2)


@pytest.fixture(scope="module")
def env():
    from neural_photo_editor_amd.lib import load_train_library
    from neural_photo_editor_amd import trainer as T
    lib = load_train_library()
    return lib, T, T.K(lib)


def dparams(shapes):
    return [torch.zeros(int(np.prod(s)), device="cuda") for s in shapes]


CASES = [
    ("conv", dict(cin=32, cout=64, h=16)), ("conv", dict(cin=3, cout=32, h=32)), ("conv", dict(cin=64, cout=160, h=8)),
    ("deconv", dict(cin=64, cout=32, h=8)), ("deconv", dict(cin=32, cout=160, h=4)),
    ("mdc", dict(cin=32, cout=32, h=16, scales=[0, 2])), ("mdc", dict(cin=64, cout=64, h=8, scales=[0, 2, 3])),
    ("mdc", dict(cin=32, cout=2, h=32, scales=[2, 3, 4])), ("mdc", dict(cin=2, cout=2, h=32, scales=[2, 3, 4])),
    ("mdc", dict(cin=4, cout=2, h=32, scales=[2, 3, 4])),
    # few-filter layers wide enough for the VALU head kernels (forward: mdc_head_kernel, backward-weight:
    # mdc_head_wgrad_kernel) instead of the padded MFMA tiles
    ("mdc", dict(cin=128, cout=2, h=16, scales=[2, 3, 4])), ("mdc", dict(cin=64, cout=2, h=32, scales=[0, 2])),
    ("mdc", dict(cin=128, cout=3, h=16, scales=[2, 3, 4])),
]


@pytest.mark.parametrize("kind,g", CASES)
@pytest.mark.parametrize("n", [3, 8])
def test_spatial_layers(env, kind, g, n):
    lib, T, k = env
    rs = np.random.RandomState(hash((kind, g["cin"], g["cout"], n)) % 2 ** 31)
    cin, cout, h = g["cin"], g["cout"], g["h"]
    x = rs.randn(n, cin, h, h).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    if kind == "conv":
        W = (rs.randn(cout, cin, 5, 5) * 0.1).astype(np.float32)
        params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)]
        y = F.conv2d(xt, params[0], stride=2, padding=2)
        layer = T.Layer(lib, T.K_CONV, cin, cout, h, h)
    elif kind == "deconv":
        W = (rs.randn(cin, cout, 5, 5) * 0.1).astype(np.float32)
        params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)]
        y = F.conv_transpose2d(xt, torch.flip(params[0], (2, 3)), stride=2, padding=2, output_padding=1)
        layer = T.Layer(lib, T.K_DECONV, cin, cout, h, h)
    else:
        sc = g["scales"]
        W = (rs.randn(cout, cin, 3, 3) * 0.2).astype(np.float32)
        coeffs = [rs.uniform(0.5, 1.5, cout).astype(np.float32) for _ in range(1 + len(sc))]
        params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)] + [torch.tensor(c, dtype=torch.float64, requires_grad=True) for c in coeffs]
        y = F.conv2d(xt, params[0], padding=1) * params[1].reshape(1, -1, 1, 1)
        for i, s in enumerate(sc):
            cf = params[2 + i].reshape(1, -1, 1, 1)
            if s == 0:
                y = y + F.conv2d(xt, params[0].mean((2, 3), keepdim=True)) * cf
            else:
                y = y + F.conv2d(xt, params[0], padding=s, dilation=s) * cf
        layer = T.Layer(lib, T.K_MDC, cin, cout, h, h, scales=sc)
    dy = rs.randn(*y.shape).astype(np.float32)
    gx, *gp = torch.autograd.grad(y, [xt] + params, torch.tensor(dy, dtype=torch.float64))
    dev_params = [torch.from_numpy(p.detach().numpy().astype(np.float32).ravel()).cuda() for p in params]
    layer.set_params(dev_params)
    oh = y.shape[2]
    xd, dyd = to_nhwc(x), to_nhwc(dy)
    yd = torch.zeros(n, oh, oh, cs(cout), device="cuda")
    layer.forward(xd, n, yd)
    assert rel(from_nhwc(yd, cout), y.detach().numpy()) < TOL
    dxd = torch.zeros(n, h, h, cs(cin), device="cuda")
    layer.backward_data(dyd, n, dxd)
    assert rel(from_nhwc(dxd, cin), gx.numpy()) < TOL
    layer.backward_data(dyd, n, dxd, accumulate=True)
    assert rel(from_nhwc(dxd, cin), 2 * gx.numpy()) < TOL
    dp = dparams([p.shape for p in params])
    layer.backward_weight(xd, dyd, n, dp)
    for got, ref in zip(dp, gp):
        assert rel(got.cpu().numpy().reshape(ref.shape), ref.numpy()) < TOL
    layer.backward_weight(xd, dyd, n, dp, accumulate=True)
    assert rel(dp[0].cpu().numpy().reshape(gp[0].shape), 2 * gp[0].numpy()) < TOL
    layer.close()


@pytest.mark.parametrize("kind,g", [("conv", dict(cin=32, cout=64, h=16)), ("conv", dict(cin=64, cout=160, h=16)), ("deconv", dict(cin=64, cout=32, h=8)),
                                    ("deconv", dict(cin=32, cout=160, h=8)), ("mdc", dict(cin=64, cout=64, h=16, scales=[0, 2, 3]))])
@pytest.mark.parametrize("n", [3, 8, 32])
def test_gemm_epilogue_statistics_equal_colstats(env, kind, g, n, monkeypatch):
    """Round 5: the batch statistics of a normalised tensor computed in the epilogue of the GEMM that stores it
    (ian_layer_stats_next -> TgStats, kernels_tapgemm.hip) against the colstats pass that re-reads the tensor: same outputs bit for
    bit (the statistics are a by-product), same float64 sums up to the summation order -- mode 1 (sum v, sum v^2: every element
    widened first) to 1e-13, mode 2 (g = dA act'(a), g * xhat: float32 products as in colstats) to float32 round-off of the terms.
    Ragged row tiles (n = 3), several column tiles (160 channels), four parity classes (stride-2 conv backward), accumulate."""
    lib, T, k = env
    import ctypes as C
    from neural_photo_editor_amd.lib import is_ablation_build
    if not is_ablation_build():
        # a measured negative result (4 % slower per update): compiled into libian_ablation.so only; the
        # product library answers every request with "0 chunks, run colstats" -- checked here -- and tests/test_gpu_ablation.py
        # runs this test against the ablation library
        monkeypatch.setenv("IAN_OPTS", "tg_split=0")
        L0 = T.Layer(lib, T.K_CONV, 32, 64, 16, 16)
        L0.set_params([torch.randn(64 * 32 * 25, device="cuda") * 0.1])
        wsx = torch.zeros(1 << 16, dtype=torch.float64, device="cuda")
        assert lib.ian_layer_stats_next(L0.h, 1, None, None, None, None, 0, C.c_void_p(wsx.data_ptr()), 1 << 16) == 0
        L0.forward(torch.randn(n, 16, 16, 32, device="cuda"), n, torch.zeros(n, 8, 8, 64, device="cuda"))
        assert lib.ian_layer_stats_chunks(L0.h) == 0
        L0.close()
        pytest.skip("GEMM-epilogue statistics are compiled into libian_ablation.so only (see tests/test_gpu_ablation.py)")
    # these toy layers would be split over K by the scheduling heuristic (few row tiles), and split-K launches do not carry the
    # statistics (the caller falls back to colstats: asserted at the end); the training step's big layers are not split
    monkeypatch.setenv("IAN_OPTS", "tg_split=0")
    rs = np.random.RandomState(7 + n)
    cin, cout, h = g["cin"], g["cout"], g["h"]
    if kind == "conv":
        layer, oh, params = T.Layer(lib, T.K_CONV, cin, cout, h, h), h // 2, [(rs.randn(cout, cin, 5, 5) * 0.1).astype(np.float32)]
    elif kind == "deconv":
        layer, oh, params = T.Layer(lib, T.K_DECONV, cin, cout, h, h), 2 * h, [(rs.randn(cin, cout, 5, 5) * 0.1).astype(np.float32)]
    else:
        sc = g["scales"]
        layer, oh = T.Layer(lib, T.K_MDC, cin, cout, h, h, scales=sc), h
        params = [(rs.randn(cout, cin, 3, 3) * 0.2).astype(np.float32)] + [rs.uniform(0.5, 1.5, cout).astype(np.float32) for _ in range(1 + len(sc))]
    layer.set_params([torch.from_numpy(p.ravel()).cuda() for p in params])
    cap = 1 << 22
    ws = torch.zeros(cap, dtype=torch.float64, device="cuda")
    ws2 = torch.zeros(cap, dtype=torch.float64, device="cuda")

    def arm(mode, a=None, yraw=None, mean=None, istd=None, act=0):
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        assert lib.ian_layer_stats_next(layer.h, mode, p(a), p(yraw), p(mean), p(istd), act, C.c_void_p(ws.data_ptr()), cap) == 0

    chunks = lambda: int(lib.ian_layer_stats_chunks(layer.h))
    # ---- forward: statistics of the raw output (Lasagne batch_norm in training mode) -------------------------------------
    x = torch.randn(n, h, h, cs(cin), device="cuda")
    x[..., cin:] = 0
    y0 = torch.zeros(n, oh, oh, cs(cout), device="cuda")
    y1 = torch.zeros_like(y0)
    layer.forward(x, n, y0)
    assert chunks() == 0                                          # nothing armed: nothing produced
    arm(1)
    layer.forward(x, n, y1)
    nch = chunks()
    assert nch > 0 and torch.equal(y0, y1)
    rows = n * oh * oh
    full, ref = torch.zeros(2 * cout, dtype=torch.float64, device="cuda"), torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
    k.tree_sum(ws, nch, 2 * cout, full)
    k.colstats(0, y0, None, None, None, None, rows, cout, cs(cout), 0, ws2, min(rows, 256), ref)
    yy = y0.cpu().numpy().astype(np.float64).reshape(rows, -1)[:, :cout]
    assert np.allclose(full.cpu().numpy()[:cout], yy.sum(0), rtol=0, atol=1e-12 * np.abs(yy).sum(0).max())
    assert np.allclose(full.cpu().numpy()[cout:], (yy ** 2).sum(0), rtol=1e-13, atol=0)
    assert np.allclose(full.cpu().numpy(), ref.cpu().numpy(), rtol=1e-12, atol=1e-12 * np.abs(yy).sum(0).max())
    layer.forward(x, n, y1)
    assert chunks() == 0                                          # one shot
    # ---- backward-data: statistics of g = dA * lrelu'(a) and g * xhat of the gradient the launch stores -------------------
    dy = torch.randn(n, oh, oh, cs(cout), device="cuda")
    dy[..., cout:] = 0
    a = torch.randn(n, h, h, cs(cin), device="cuda")             # the forward activation of the tensor whose gradient is produced
    yraw = torch.randn(n, h, h, cs(cin), device="cuda")
    mean, istd = torch.randn(cin, device="cuda") * 0.1, torch.rand(cin, device="cuda") + 0.5
    for accumulate in (False, True):
        dx0 = torch.randn(n, h, h, cs(cin), device="cuda") if accumulate else torch.zeros(n, h, h, cs(cin), device="cuda")
        dx1 = dx0.clone()
        layer.backward_data(dy, n, dx0, accumulate=accumulate)
        arm(2, a, yraw, mean, istd, T.ACT["lrelu"])
        layer.backward_data(dy, n, dx1, accumulate=accumulate)
        nch = chunks()
        assert nch > 0 and torch.equal(dx0, dx1)
        rows = n * h * h
        full, ref = torch.zeros(2 * cin, dtype=torch.float64, device="cuda"), torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
        k.tree_sum(ws, nch, 2 * cin, full)
        k.colstats(1, dx0, a, yraw, mean, istd, rows, cin, cs(cin), T.ACT["lrelu"], ws2, min(rows, 256), ref)
        gg = dx0.cpu().numpy().astype(np.float64).reshape(rows, -1)[:, :cin] * np.where(a.cpu().numpy().reshape(rows, -1)[:, :cin] > 0, 1.0, 0.2)
        xh = (yraw.cpu().numpy().astype(np.float64).reshape(rows, -1)[:, :cin] - mean.cpu().numpy()) * istd.cpu().numpy()
        scale = max(np.abs(gg).sum(0).max(), np.abs(gg * xh).sum(0).max())
        assert np.allclose(full.cpu().numpy()[:cin], gg.sum(0), rtol=0, atol=3e-7 * scale)
        assert np.allclose(full.cpu().numpy()[cin:], (gg * xh).sum(0), rtol=0, atol=3e-7 * scale)
        assert np.allclose(full.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=3e-7 * scale)
    layer.close()
    # a layer whose schedule splits K cannot carry the statistics: it says so (0 chunks) and its output is unaffected
    monkeypatch.setenv("IAN_OPTS", "tg_split=1,tg_target_items=4096,tg_min_steps=1")
    sp = T.Layer(lib, T.K_CONV, 64, 64, 8, 8)
    sp.set_params([torch.randn(64 * 64 * 25, device="cuda") * 0.1])
    xs, ys = torch.randn(2, 8, 8, 64, device="cuda"), torch.zeros(2, 4, 4, 64, device="cuda")
    assert lib.ian_layer_stats_next(sp.h, 1, None, None, None, None, 0, C.c_void_p(ws.data_ptr()), cap) == 0
    sp.forward(xs, 2, ys)
    assert lib.ian_layer_stats_chunks(sp.h) == 0
    sp.close()


@pytest.mark.parametrize("which", ["gen", "discrim"])
def test_step_with_epilogue_statistics_equals_the_colstats_step(which):
    """The training step with the batch statistics riding on the producing GEMMs (fused_stats = 1: libian_ablation.so, a measured
    negative result) against the same step with the colstats passes (fused_stats = 0, the product): the float64 sums differ in
    summation order only, so the float32 statistics agree except in an occasional last bit -- losses to 2e-6, every gradient
    tensor to 2e-5 relative L2.  In the product library the option changes nothing (same bits)."""
    import os
    from neural_photo_editor_amd.lib import is_ablation_build
    from neural_photo_editor_amd import synthetic as S
    from neural_photo_editor_amd.trainer import Trainer
    CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neural_photo_editor_amd", "configs", "IAN.py")
    B = 16
    P = S.make_train_params(S.make_params("IAN", 1))
    X, Z = S.make_images(B, seed=60), S.make_latents(B, seed=70)
    eps = np.random.RandomState(80).randn(B, 100).astype(np.float32)
    dev = lambda t: torch.from_numpy(np.ascontiguousarray(t)).cuda()
    out = {}
    for fused in (1, 0):
        tr = Trainer(CFG, P, batch=B)
        tr.set_option("fused_stats", fused)
        tr.forward(dev(X), dev(Z), dev(eps))
        m = tr.metrics()
        tr.backward(which)
        tr._finish_allreduce(which)
        torch.cuda.synchronize()
        out[fused] = (m, {g: tr.grads_numpy(g) for g in (("dec" if which == "gen" else "enc"), "Z")},
                      {n: tr.read(n) for n in ("bnorm2.mean", "bnorm_dc4.inv_std", "dec_conv3abnorm1.mean")})
        tr.close()
    for key in out[1][0]:
        assert abs(out[1][0][key] - out[0][0][key]) <= 2e-6 * max(1.0, abs(out[0][0][key])), key
    l2 = lambda a_, b_: float(np.linalg.norm((a_ - b_).astype(np.float64)) / (np.linalg.norm(b_.astype(np.float64)) + 1e-30))
    worst = sorted(((l2(out[1][1][g][n], out[0][1][g][n]), n) for g in out[0][1] for n in out[0][1][g]), reverse=True)
    assert worst[0][0] < 2e-5, worst[:5]
    if not is_ablation_build():
        assert worst[0][0] == 0.0, worst[:3]                      # product: the requests are declined, the option is inert
    for n in out[0][2]:                                            # running averages of the real-data pass follow the same statistics
        assert np.allclose(out[1][2][n], out[0][2][n], rtol=1e-6, atol=1e-7), n


@pytest.mark.parametrize("fin,fout,flat,unflat", [(1000, 100, None, None), (256 * 16, 200, (256, 4, 4), None), (100, 64 * 16, None, (64, 4, 4)),
                                                  (1024, 2500, None, None)])
def test_dense_layers(env, fin, fout, flat, unflat):
    lib, T, k = env
    n = 5
    rs = np.random.RandomState(fin + fout)
    W = (rs.randn(fin, fout) * 0.05).astype(np.float32)
    x = rs.randn(n, fin).astype(np.float32)
    dy = rs.randn(n, fout).astype(np.float32)
    layer = T.Layer(lib, T.K_DENSE, fin, fout, flat=flat or (0, 0, 0), unflat=unflat or (0, 0, 0))
    layer.set_params([torch.from_numpy(W.ravel()).cuda()])
    # reference-order vectors <-> internal (H,W,C) order of flattened maps (App. B.6)
    def to_internal(v, chw):
        if chw is None:
            out = np.zeros((v.shape[0], cs(v.shape[1])), np.float32); out[:, :v.shape[1]] = v; return out
        c, h, w = chw
        return v.reshape(-1, c, h, w).transpose(0, 2, 3, 1).reshape(v.shape[0], -1).copy()
    def from_internal(t, width, chw):
        a = t.cpu().numpy()
        if chw is None:
            return a[:, :width]
        c, h, w = chw
        return a.reshape(-1, h, w, c).transpose(0, 3, 1, 2).reshape(a.shape[0], -1)
    xd = torch.from_numpy(to_internal(x, flat)).cuda()
    dyd = torch.from_numpy(to_internal(dy, unflat)).cuda()
    ystride = dyd.shape[1]
    yd = torch.zeros(n, ystride, device="cuda")
    layer.forward(xd, n, yd, y_stride=ystride)
    assert rel(from_internal(yd, fout, unflat), x.astype(np.float64) @ W) < TOL
    dxd = torch.zeros_like(xd)
    layer.backward_data(dyd, n, dxd, dx_stride=xd.shape[1])
    assert rel(from_internal(dxd, fin, flat), dy.astype(np.float64) @ W.T) < TOL
    dW = torch.zeros(fin * fout, device="cuda")
    layer.backward_weight(xd, dyd, n, [dW])
    assert rel(dW.cpu().numpy().reshape(fin, fout), x.astype(np.float64).T @ dy) < TOL
    layer.close()


@pytest.mark.parametrize("rows,nchunks,C,act", [(4 * 1024, 8, 128, 2), (2 * 256, 2, 256, 1), (3 * 100, 3, 36, 2), (4 * 16, 64, 32, 0),
                                                (128, 128, 2048, 1)])
def test_fused_statistics_are_bitwise_the_two_stage_ones(env, rows, nchunks, C, act):
    """ian_k_bn_stats_affine / ian_k_bn_bwd_stats (one launch for the second stage, used by the single-process step) against
    ian_k_colstats + ian_k_bn_make_affine + the ian_k_axpy running-average / gradient accumulations (the data-parallel step):
    bit for bit, on chunks long enough for the 32-row-lane block shape (>= 64 rows per chunk), with a ragged tail (100 rows per
    chunk) and on the short-chunk shape; the sums themselves against float64."""
    lib, T, k = env
    rs = np.random.RandomState(rows + C)
    stride = cs(C)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
    pad = lambda v: dev(np.pad(v, ((0, 0), (0, stride - C))))
    y, dA = (rs.randn(rows, C) * 2 + 0.5).astype(np.float32), rs.randn(rows, C).astype(np.float32)
    gamma, beta = dev(rs.uniform(0.5, 1.5, C)), dev(rs.randn(C))
    run0 = rs.randn(2, C).astype(np.float32)
    g0 = rs.randn(2, C).astype(np.float32)
    yd, dAd = pad(y), pad(dA)
    ws = torch.zeros(nchunks * 2 * C, device="cuda", dtype=torch.float64)      # float64 partial sums (kernels_train.hip NUMERICS)
    out = {}
    for fused in (False, True):
        bn = T.BN(torch, C, "cuda")
        rm, ri = dev(run0[0]), dev(run0[1])
        gb, gg = dev(g0[0]), dev(g0[1])
        ad, dyd = torch.zeros_like(yd), torch.zeros_like(yd)
        if fused:
            k.bn_stats_affine(yd, rows, C, stride, ws, nchunks, bn.sums, float(rows), 1e-4, gamma, beta, bn.mean, bn.inv_std, bn.scale,
                              bn.shift, rm, ri, 1.0 - 0.1, 0.1)
        else:
            k.colstats(0, yd, None, None, None, None, rows, C, stride, 0, ws, nchunks, bn.sums)
            k.bn_make_affine(bn.sums, float(rows), 1e-4, gamma, beta, C, bn.mean, bn.inv_std, bn.scale, bn.shift)
            for r, cur in ((rm, bn.mean), (ri, bn.inv_std)):
                k.axpy(1.0 - 0.1, r, r, C, 0)
                k.axpy(0.1, cur, r, C, 1)
        k.affine(yd, ad, bn.scale, bn.shift, rows, C, stride, act)
        if fused:
            k.bn_bwd_stats(dAd, ad, yd, bn.mean, bn.inv_std, rows, C, stride, act, ws, nchunks, bn.bsums, gb, 1, gg, 0)
        else:
            k.colstats(1, dAd, ad, yd, bn.mean, bn.inv_std, rows, C, stride, act, ws, nchunks, bn.bsums)
            k.axpy_f64(1.0, bn.bsums[:C], gb, C, 1)
            k.axpy_f64(1.0, bn.bsums[C:], gg, C, 0)
        out[fused] = [v.cpu().numpy().copy() for v in (bn.sums, bn.mean, bn.inv_std, bn.scale, bn.shift, rm, ri, bn.bsums, gb, gg)]
    for i, (u, f) in enumerate(zip(out[False], out[True])):
        assert np.array_equal(u, f), i
    sums, bsums = out[True][0], out[True][7]
    y64 = y.astype(np.float64)
    assert sums.dtype == np.float64 and rel(sums[:C], y64.sum(0)) < 1e-12 and rel(sums[C:], (y64 ** 2).sum(0)) < 1e-13
    mean, var = y64.mean(0), y64.var(0)
    pre = (y64 - mean) / np.sqrt(var + 1e-4) * gamma.cpu().numpy() + beta.cpu().numpy()
    dact = {0: np.ones_like(pre), 1: (pre > 0).astype(np.float64), 2: np.where(pre > 0, 1.0, 0.2)}[act]
    g = dA * dact
    assert rel(bsums[:C], g.sum(0)) < 1e-5 and rel(bsums[C:], (g * (y64 - mean) / np.sqrt(var + 1e-4)).sum(0)) < 1e-5
    b32 = bsums.astype(np.float32)                                        # the float64 sums are rounded to float32 once
    assert np.array_equal(out[True][8], g0[0] + b32[:C]) and np.array_equal(out[True][9], b32[C:])   # accumulate vs overwrite


def test_batch_variance_is_well_conditioned(env):
    """Round-3 verdict, weak #1: the variance used to be E[x^2]-E[x]^2 in float32, whose cancellation error grows with
    mean^2/var.  Here mean^2/var = 2.5e5 (a float32 one-pass variance is off by percents, a float32 two-pass one -- Lasagne's
    input.var, minilasagne.py:607 -- by ~1e-6): the float64 partial sums must give mean / inv_std to float32 round-off and the
    normalised activations to 1e-5 of the float64 result."""
    lib, T, k = env
    rs = np.random.RandomState(7)
    rows, C = 4 * 1024, 128
    y = (50.0 + 0.1 * rs.randn(rows, C)).astype(np.float32)
    y64 = y.astype(np.float64)
    onepass32 = (y * y).sum(0, dtype=np.float32) / np.float32(rows) - (y.sum(0, dtype=np.float32) / np.float32(rows)) ** 2
    assert np.abs(onepass32 / y64.var(0) - 1).max() > 1e-2                # the hazard is real on this input
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    yd = torch.from_numpy(y).cuda()
    bn = T.BN(torch, C, "cuda")
    ws = torch.zeros(8 * 2 * C, device="cuda", dtype=torch.float64)
    for fused in (False, True):
        if fused:
            k.bn_stats_affine(yd, rows, C, C, ws, 8, bn.sums, float(rows), 1e-4, gamma, beta, bn.mean, bn.inv_std, bn.scale, bn.shift, None, None,
                              0.9, 0.1)
        else:
            k.colstats(0, yd, None, None, None, None, rows, C, C, 0, ws, 8, bn.sums)
            k.bn_make_affine(bn.sums, float(rows), 1e-4, gamma, beta, C, bn.mean, bn.inv_std, bn.scale, bn.shift)
        istd = 1.0 / np.sqrt(y64.var(0) + 1e-4)
        assert np.abs(bn.mean.cpu().numpy() / y64.mean(0) - 1).max() < 1.2e-7
        assert np.abs(bn.inv_std.cpu().numpy() / istd - 1).max() < 1.2e-7, np.abs(bn.inv_std.cpu().numpy() / istd - 1).max()
    # x*scale + shift with |mean*scale| ~ 500 x the output: the folded form costs ~eps32 * 500 on activations of O(1)
    ad = torch.zeros_like(yd)
    k.affine(yd, ad, bn.scale, bn.shift, rows, C, C, 0)
    ref = (y64 - y64.mean(0)) * istd
    assert np.abs(ad.cpu().numpy() - ref).max() < 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("rows,C,act", [(4 * 16, 32, 2), (3 * 64, 128, 2), (5, 1000, 1), (4, 100, 0)])
def test_batchnorm_train_forward_backward(env, rows, C, act):
    lib, T, k = env
    rs = np.random.RandomState(rows + C)
    stride = cs(C)
    y = (rs.randn(rows, C) * 2 + 0.5).astype(np.float32)
    gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), rs.randn(C).astype(np.float32)
    dA = rs.randn(rows, C).astype(np.float32)
    yt = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    gt, bt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mean, var = yt.mean(0), ((yt - yt.mean(0)) ** 2).mean(0)
    pre = (yt - mean) / torch.sqrt(var + 1e-4) * gt + bt
    a = {0: pre, 1: torch.relu(pre), 2: F.leaky_relu(pre, 0.2)}[act]
    gy, gg, gb = torch.autograd.grad(a, [yt, gt, bt], torch.tensor(dA, dtype=torch.float64))
    pad = lambda v: torch.from_numpy(np.pad(v, ((0, 0), (0, stride - C)))).cuda()
    yd, dAd = pad(y), pad(dA)
    ad = torch.zeros_like(yd)
    bn = T.BN(torch, C, "cuda")
    ws = torch.zeros(256 * 2 * C, device="cuda", dtype=torch.float64)
    k.colstats(0, yd, None, None, None, None, rows, C, stride, 0, ws, min(256, rows), bn.sums)
    k.bn_make_affine(bn.sums, float(rows), 1e-4, torch.from_numpy(gamma).cuda(), torch.from_numpy(beta).cuda(), C, bn.mean, bn.inv_std, bn.scale, bn.shift)
    k.affine(yd, ad, bn.scale, bn.shift, rows, C, stride, act)
    assert rel(ad.cpu().numpy()[:, :C], a.detach().numpy()) < TOL
    k.colstats(1, dAd, ad, yd, bn.mean, bn.inv_std, rows, C, stride, act, ws, min(256, rows), bn.bsums)
    dyd = torch.zeros_like(yd)
    k.bn_bwd(dAd, ad, yd, bn.mean, bn.inv_std, bn.scale, bn.bsums, float(rows), dyd, rows, C, stride, act)
    s = bn.bsums.cpu().numpy()
    assert rel(s[:C], gb.numpy()) < 1e-4 and rel(s[C:], gg.numpy()) < 1e-4
    assert rel(dyd.cpu().numpy()[:, :C], gy.numpy()) < 1e-4


def test_minibatch_layer_and_head(env):
    lib, T, k = env
    n, nin, nk, nd = 6, 64, 20, 5
    rs = np.random.RandomState(0)
    theta = (rs.randn(nin, nk, nd) * 0.05).astype(np.float32)
    lws = (rs.randn(nk, nd) * 0.1).astype(np.float32)
    b = rs.randn(nk).astype(np.float32)
    feat = rs.randn(n, nin).astype(np.float32)
    Wd = (rs.randn(nin + nk, 3) * 0.3).astype(np.float32)
    tt = lambda v: torch.tensor(v, dtype=torch.float64, requires_grad=True)
    th, lw, bb, ft, wd = tt(theta), tt(lws), tt(b), tt(feat), tt(Wd)
    W = th * (torch.exp(lw) / torch.sqrt((th ** 2).sum(0))).unsqueeze(0)
    act = torch.tensordot(ft, W, dims=([1], [0]))
    ad = (act.unsqueeze(3) - act.permute(1, 2, 0).unsqueeze(0)).abs().sum(2) + 1e6 * torch.eye(n, dtype=torch.float64).unsqueeze(1)
    f = torch.exp(-ad).sum(2) + bb.unsqueeze(0)
    mb = torch.cat([ft, f], 1)
    p = torch.softmax(mb @ wd, 1)
    loss = 0.7 * (-torch.log(p[:, 1])).sum() + 0.2 * (-torch.log(p[:, 2])).sum()
    gth, glw, gbb, gft, gwd = torch.autograd.grad(loss, [th, lw, bb, ft, wd])
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    ncol = nk * nd
    Wdev, colscale = torch.zeros(nin * ncol, device="cuda"), torch.zeros(ncol, device="cuda")
    k.mb_weight(c(theta), c(lws), Wdev, colscale, nin, ncol)
    assert rel(Wdev.cpu().numpy().reshape(nin, nk, nd), W.detach().numpy()) < TOL
    layer = T.Layer(lib, T.K_DENSE, nin, ncol)
    layer.set_params([Wdev])
    featd = c(feat)
    sa, sm = cs(ncol), cs(nin + nk)
    actd = torch.zeros(n, sa, device="cuda")
    layer.forward(featd, n, actd, y_stride=sa)
    mbd = torch.zeros(n, sm, device="cuda")
    k.mb_forward(actd, n, sa, 0, n, nk, nd, c(b), featd, nin, nin, mbd, sm)
    assert rel(mbd.cpu().numpy()[:, :nin + nk], mb.detach().numpy()) < TOL
    pd, ld = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 4, device="cuda")
    k.disc_head(mbd, sm, nin + nk, c(Wd), n, 1, 2, 1, pd, ld)
    assert rel(pd.cpu().numpy(), p.detach().numpy()) < TOL
    assert rel(ld.cpu().numpy()[:, 0], -np.log(p.detach().numpy()[:, 1])) < TOL
    dlog, dmb = torch.zeros(n, 4, device="cuda"), torch.zeros(n, sm, device="cuda")
    k.disc_head_bwd(pd, c(Wd), nin + nk, n, 1, 0.7, 2, 0.2, dlog, dmb, sm)
    dWd = torch.zeros((nin + nk) * 3, device="cuda")
    k.disc_head_wgrad(mbd, sm, nin + nk, n, dlog, dWd, 0)
    assert rel(dWd.cpu().numpy().reshape(-1, 3), gwd.numpy()) < 1e-4
    dact = torch.zeros(n, sa, device="cuda")
    k.mb_backward(actd, n, sa, 0, n, nk, nd, dmb.view(-1)[nin:], sm, dact, sa)
    dfeat = torch.zeros(n, nin, device="cuda")
    k.grad_pass(dmb, sm, 0, dfeat, None, nin, n, nin, 0, 0)
    layer.backward_data(dact, n, dfeat, dx_stride=nin, accumulate=True)
    assert rel(dfeat.cpu().numpy(), gft.numpy()) < 1e-4
    dW = torch.zeros(nin * ncol, device="cuda")
    layer.backward_weight(featd, dact, n, [dW])
    dth, dlw = torch.zeros(nin * ncol, device="cuda"), torch.zeros(ncol, device="cuda")
    k.mb_weight_bwd(c(theta), colscale, dW, dth, dlw, nin, ncol, 0)
    assert rel(dth.cpu().numpy().reshape(theta.shape), gth.numpy()) < 1e-4
    assert rel(dlw.cpu().numpy().reshape(lws.shape), glw.numpy()) < 1e-4
    layer.close()


def test_sampling_iaf_losses_ortho_adam(env):
    lib, T, k = env
    from neural_photo_editor_amd import made
    n, d = 5, 100
    rs = np.random.RandomState(3)
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    tt = lambda v: torch.tensor(v, dtype=torch.float64, requires_grad=True)
    pad = lambda v: np.pad(v, ((0, 0), (0, 128 - d))).astype(np.float32)
    mu, ls, eps = rs.randn(n, d).astype(np.float32), (rs.randn(n, d) * 0.3).astype(np.float32), rs.randn(n, d).astype(np.float32)
    masks = made.masks_once(d)
    Ws = [(rs.randn(d, d) * 0.1).astype(np.float32) * m for _ in range(2) for m in masks]
    bs = [(rs.randn(d) * 0.1).astype(np.float32) for _ in range(6)]
    mut, lst = tt(mu), tt(ls)
    z0 = mut + torch.exp(lst) * torch.tensor(eps, dtype=torch.float64)
    W = [torch.tensor(w, dtype=torch.float64) for w in Ws]
    Bv = [torch.tensor(b, dtype=torch.float64) for b in bs]
    # the reference graph runs each masked MLP on its own first hidden layer (layers.py:775 overwrites
    # MADE.input_layer; oracle.made_as_wired, pinned by tests/golden/ref_layers.npz 'iaf/*')
    mlp = lambda z, o: (torch.relu(z @ W[o] + Bv[o]) @ W[o + 1] + Bv[o + 1]) + (z @ W[o + 2] + Bv[o + 2])
    mm = lambda z, o: mlp(torch.relu(z @ W[o] + Bv[o]), o)
    z = (z0 - mm(z0, 0)) / torch.exp(mm(z0, 3))
    dz = rs.randn(n, d).astype(np.float32)
    kl = -0.5 * (1 + 2 * lst - mut ** 2 - torch.exp(2 * lst)).mean()
    gmu, gls = torch.autograd.grad((z * torch.tensor(dz, dtype=torch.float64)).sum() + kl, [mut, lst])
    wts, bias = c(np.stack(Ws)), c(np.stack(bs))
    mud, lsd, epsd = c(pad(mu)), c(pad(ls)), c(eps)
    z0d, zd, klt = torch.zeros(n, 128, device="cuda"), torch.zeros(n, 128, device="cuda"), torch.zeros(n, d, device="cuda")
    k.sample(mud, lsd, epsd, z0d, klt, n, d, 128, d)
    k.made_iaf(z0d, zd, wts, bias, n, d, 128)
    assert rel(zd.cpu().numpy()[:, :d], z.detach().numpy()) < TOL
    out = torch.zeros(4, device="cuda")
    k.sum_rows(klt, n * d, 1, -0.5 / (n * d), out)
    assert abs(float(out[0]) - float(kl)) < 1e-5
    dz0d, dmud, dlsd = torch.zeros(n, 128, device="cuda"), torch.zeros(n, 128, device="cuda"), torch.zeros(n, 128, device="cuda")
    k.made_iaf_bwd(z0d, c(pad(dz)), dz0d, wts, bias, n, d, 128)
    k.sample_bwd(mud, lsd, epsd, dz0d, dmud, dlsd, n, d, 128, d, 1.0 / (n * d))
    assert rel(dmud.cpu().numpy()[:, :d], gmu.numpy()) < 1e-4 and rel(dlsd.cpu().numpy()[:, :d], gls.numpy()) < 1e-4
    # pixel loss / feature loss
    a, b = rs.uniform(-1, 1, (7, 33)).astype(np.float32), rs.uniform(-1, 1, (7, 33)).astype(np.float32)
    at = tt(a)
    pl = (2 * (at - torch.tensor(b, dtype=torch.float64) + 1e-8).abs()).mean()
    (ga,) = torch.autograd.grad(3.0 * pl, [at])
    ws, da = torch.zeros(2048, device="cuda"), torch.zeros(7 * 33, device="cuda")
    k.pair_loss(c(a), c(b), da, 7 * 33, 1, 1, 0, 3.0 / (7 * 33), 0, ws, 64, 1.0 / (7 * 33), out)
    assert abs(float(out[0]) - float(pl)) < 1e-5 and rel(da.cpu().numpy().reshape(7, 33), ga.numpy()) < TOL
    ml = ((at - torch.tensor(b, dtype=torch.float64)) ** 2).mean()
    (ga,) = torch.autograd.grad(ml, [at])
    k.pair_loss(c(a), c(b), da, 7 * 33, 1, 1, 1, 1.0 / (7 * 33), 0, ws, 64, 1.0 / (7 * 33), out)
    assert abs(float(out[0]) - float(ml)) < 1e-5 and rel(da.cpu().numpy().reshape(7, 33), ga.numpy()) < TOL
    # orthogonal regulariser
    for shape in ((6, 40, 5, 5), (9, 300, 3, 3)):
        Wn = (rs.randn(*shape) * 0.1).astype(np.float32)
        wt = tt(Wn)
        y = torch.einsum("abik,abjk->aij", wt, wt) - torch.eye(shape[2], dtype=torch.float64).unsqueeze(0)
        val = y.abs().sum()
        (gw,) = torch.autograd.grad(1e-3 * val, [wt])
        dW, vals = torch.zeros(Wn.size, device="cuda"), torch.zeros(64, device="cuda")
        k.ortho(c(Wn), dW, shape[0], shape[1], shape[2], 1e-3, vals)
        assert abs(float(vals[:shape[0]].sum()) - float(val)) < 1e-3 * float(val)
        assert rel(dW.cpu().numpy().reshape(shape), gw.numpy()) < 1e-4
    # Adam (App. B.7)
    p, g = rs.randn(1000).astype(np.float32), rs.randn(1000).astype(np.float32)
    m, v = np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    pd_, md, vd = c(p), c(m), c(v)
    pr = p.astype(np.float64)
    mr, vr = np.zeros(1000), np.zeros(1000)
    for t in (1, 2, 3):
        a_t = 2e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)
        k.adam(pd_, c(g), md, vd, 1000, float(a_t), 0.5, 0.999, 1e-8)
        mr = 0.5 * mr + 0.5 * g
        vr = 0.999 * vr + 0.001 * g.astype(np.float64) ** 2
        pr = pr - a_t * mr / (np.sqrt(vr) + 1e-8)
    assert np.abs(pd_.cpu().numpy() - pr).max() < 1e-6


def test_layer_autotune_keeps_the_arithmetic(env):
    """ian_layer_autotune only changes the (tile, split-K, K-loop schedule) decomposition: forward / backward-data results
    before and after agree to float32 summation order."""
    lib, T, k = env
    n, cin, cout, h = 8, 64, 96, 16
    rs = np.random.RandomState(4)
    W = (rs.randn(cin, cout, 5, 5) * 0.05).astype(np.float32)
    x = rs.randn(n, cin, h, h).astype(np.float32)
    dy = rs.randn(n, cout, 2 * h, 2 * h).astype(np.float32)
    layer = T.Layer(lib, T.K_DECONV, cin, cout, h, h)
    Wd = torch.from_numpy(W.ravel()).cuda()
    layer.set_params([Wd])
    xd, dyd = to_nhwc(x), to_nhwc(dy)
    y0 = torch.zeros(n, 2 * h, 2 * h, cs(cout), device="cuda")
    dx0 = torch.zeros(n, h, h, cs(cin), device="cuda")
    layer.forward(xd, n, y0)
    layer.backward_data(dyd, n, dx0)
    a = torch.randn(n * 32 * 32 * 128, device="cuda")
    b = torch.randn(n * 32 * 32 * 128, device="cuda")
    layer.autotune(n, a, b)
    y1, dx1 = torch.zeros_like(y0), torch.zeros_like(dx0)
    layer.forward(xd, n, y1)
    layer.backward_data(dyd, n, dx1)
    assert rel(y1.cpu().numpy(), y0.cpu().numpy()) < 1e-5 and rel(dx1.cpu().numpy(), dx0.cpu().numpy()) < 1e-5
    ref = F.conv_transpose2d(torch.tensor(x, dtype=torch.float64), torch.flip(torch.tensor(W, dtype=torch.float64), (2, 3)), stride=2,
                             padding=2, output_padding=1).numpy()
    assert rel(from_nhwc(y1, cout), ref) < TOL
    layer.close()


@pytest.mark.parametrize("n", [2, 8])
def test_head6_forward_equals_three_layer_forwards(env, n):
    """ian_layer_head6_forward (R, G_a, B_a of IAN.py:183-199 in one pass over the 128-channel map) vs the three separate
    ian_layer_forward calls and vs float64 torch."""
    lib, T, k = env
    rs = np.random.RandomState(11 + n)
    sc = [2, 3, 4]
    x = rs.randn(n, 128, 64, 64).astype(np.float32)
    xd = to_nhwc(x)
    layers, keep, refs = [], [], []
    for i in range(3):
        W = (rs.randn(2, 128, 3, 3) * 0.05).astype(np.float32)
        coeffs = [rs.uniform(0.5, 1.5, 2).astype(np.float32) for _ in range(4)]
        xt = torch.tensor(x, dtype=torch.float64)
        Wt = torch.tensor(W, dtype=torch.float64)
        y = F.conv2d(xt, Wt, padding=1) * torch.tensor(coeffs[0], dtype=torch.float64).reshape(1, -1, 1, 1)
        for j, s in enumerate(sc):
            y = y + F.conv2d(xt, Wt, padding=s, dilation=s) * torch.tensor(coeffs[1 + j], dtype=torch.float64).reshape(1, -1, 1, 1)
        refs.append(torch.sigmoid(y).numpy() if i == 0 else y.numpy())
        layer = T.Layer(lib, T.K_MDC, 128, 2, 64, 64, scales=sc)
        params = [torch.from_numpy(a.ravel()).cuda() for a in [W] + coeffs]
        keep.append(params)
        layer.set_params(params)
        layers.append(layer)
    ys = [torch.zeros(n, 64, 64, 32, device="cuda") for _ in range(3)]
    assert layers[0].head6_forward(layers[1], layers[2], xd, n, ys[0], ys[1], ys[2], 32, (5, 0, 0))
    sep = [torch.zeros(n, 64, 64, 32, device="cuda") for _ in range(3)]
    for i in range(3):
        layers[i].forward(xd, n, sep[i], act=5 if i == 0 else 0)
    for i in range(3):
        assert rel(from_nhwc(ys[i], 2), refs[i]) < TOL, i
        assert rel(from_nhwc(ys[i], 2), from_nhwc(sep[i], 2)) < 1e-5, i
        assert float(ys[i][..., 2:].abs().max()) == 0.0          # channel padding untouched
    for l in layers:
        l.close()


@pytest.mark.parametrize("n,accumulate", [(2, False), (5, True)])
def test_head6_backward_equals_three_layer_calls(env, n, accumulate):
    """ian_layer_head6_backward (one shifted gather + two dense GEMMs for R, G_a, B_a) vs the three ian_layer_backward_weight /
    ian_layer_backward_data calls and vs float64 autograd of the layer expression (IAN.py:183-199; MDCL W + coefficients)."""
    lib, T, k = env
    rs = np.random.RandomState(23 + n)
    sc = [2, 3, 4]
    x = rs.randn(n, 128, 64, 64).astype(np.float32)
    xd = to_nhwc(x)
    layers, keep, refs, dys = [], [], [], []
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    total = 0.0
    leaves = []
    for i in range(3):
        W = (rs.randn(2, 128, 3, 3) * 0.05).astype(np.float32)
        coeffs = [rs.uniform(0.5, 1.5, 2).astype(np.float32) for _ in range(4)]
        dy = rs.randn(n, 2, 64, 64).astype(np.float32)
        Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
        ct = [torch.tensor(c, dtype=torch.float64, requires_grad=True) for c in coeffs]
        y = F.conv2d(xt, Wt, padding=1) * ct[0].reshape(1, -1, 1, 1)
        for j, s in enumerate(sc):
            y = y + F.conv2d(xt, Wt, padding=s, dilation=s) * ct[1 + j].reshape(1, -1, 1, 1)
        total = total + (y * torch.tensor(dy, dtype=torch.float64)).sum()
        leaves.append([Wt] + ct)
        layer = T.Layer(lib, T.K_MDC, 128, 2, 64, 64, scales=sc)
        params = [torch.from_numpy(a.ravel()).cuda() for a in [W] + coeffs]
        keep.append(params)
        layer.set_params(params)
        layers.append(layer)
        dys.append(to_nhwc(dy))
    total.backward()
    refs = [[t.grad.numpy().ravel() for t in lv] for lv in leaves]
    base = 0.25 if accumulate else 0.0
    fused = [[torch.full_like(p, base) for p in ps] for ps in keep]
    dx_f = torch.full((n, 64, 64, 128), base, device="cuda")
    assert layers[0].head6_backward(layers[1], layers[2], xd, dys, n, 32, dx=dx_f, dx_stride=128, dx_accumulate=accumulate,
                                    dparams3=fused, accumulate=accumulate)
    sep = [[torch.full_like(p, base) for p in ps] for ps in keep]
    dx_s = torch.full((n, 64, 64, 128), base, device="cuda")
    for i in range(3):
        layers[i].backward_weight(xd, dys[i], n, sep[i], accumulate=accumulate)
        layers[i].backward_data(dys[i], n, dx_s, accumulate=accumulate or i > 0)
    torch.cuda.synchronize()
    for i in range(3):
        for j in range(5):
            got, two, ref = fused[i][j].cpu().numpy() - base, sep[i][j].cpu().numpy() - base, refs[i][j]
            assert rel(got, ref) < TOL, (i, j)
            assert rel(got, two) < 1e-5, (i, j)
    assert rel(from_nhwc(dx_f, 128) - base, xt.grad.numpy()) < TOL
    assert rel(from_nhwc(dx_f, 128), from_nhwc(dx_s, 128)) < 1e-5
    # data gradient alone (no x, no parameter gradients)
    dx_o = torch.zeros(n, 64, 64, 128, device="cuda")
    assert layers[0].head6_backward(layers[1], layers[2], None, dys, n, 32, dx=dx_o, dx_stride=128)
    assert rel(from_nhwc(dx_o, 128), xt.grad.numpy()) < TOL
    for l in layers:
        l.close()


@pytest.mark.parametrize("cin", [2, 4])
def test_thin_mdcl_lds_staged_equals_direct(env, cin, monkeypatch):
    """G_b / B_b (IAN.py:187-206; 2 or 4 input channels, 2 filters): the LDS-staged kernel (enough row bands to fill the
    chip) and the direct-from-L1 kernel run the same FMAs in the same order -> identical forward and backward-data."""
    lib, T, k = env
    n, sc = 32, [2, 3, 4]
    rs = np.random.RandomState(70 + cin)
    W = (rs.randn(2, cin, 3, 3) * 0.3).astype(np.float32)
    coeffs = [rs.uniform(0.5, 1.5, 2).astype(np.float32) for _ in range(4)]
    x = to_nhwc(rs.randn(n, cin, 64, 64).astype(np.float32))
    dy = to_nhwc(rs.randn(n, 2, 64, 64).astype(np.float32))
    outs = []
    for tile in (1, 0):
        monkeypatch.setenv("IAN_OPTS", "mdc_thin_tile=%d" % tile)
        layer = T.Layer(lib, T.K_MDC, cin, 2, 64, 64, scales=sc)
        params = [torch.from_numpy(a.ravel()).cuda() for a in [W] + coeffs]
        layer.set_params(params)
        y = torch.zeros(n, 64, 64, 32, device="cuda")
        dx = torch.zeros(n, 64, 64, 32, device="cuda")
        layer.forward(x, n, y, act=5)
        layer.backward_data(dy, n, dx)
        torch.cuda.synchronize()
        outs.append((y.cpu().numpy(), dx.cpu().numpy()))
        layer.close()
    assert np.abs(outs[0][0]).max() > 0 and np.abs(outs[0][1]).max() > 0
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    xt = torch.tensor(from_nhwc(x, cin), dtype=torch.float64)
    Wt = torch.tensor(W, dtype=torch.float64)
    ref = F.conv2d(xt, Wt, padding=1) * torch.tensor(coeffs[0], dtype=torch.float64).reshape(1, -1, 1, 1)
    for j, s in enumerate(sc):
        ref = ref + F.conv2d(xt, Wt, padding=s, dilation=s) * torch.tensor(coeffs[1 + j], dtype=torch.float64).reshape(1, -1, 1, 1)
    assert rel(from_nhwc(torch.from_numpy(outs[0][0]), 2), torch.sigmoid(ref).numpy()) < TOL


@pytest.mark.parametrize("kind", ["conv", "deconv"])
def test_backward_weight_eight_wave_tile_is_bitwise_the_four_wave_tile(env, kind, monkeypatch):
    """tapwgrad with 8-wave 128x128 workgroups (default) and with the 4-wave ones (wg_w8=0): every output element
    accumulates the same products in the same order -> identical weight gradients.  The 4-wave tile is compiled into
    libian_ablation.so only (tests/test_gpu_ablation.py runs this test against it)."""
    from neural_photo_editor_amd.lib import is_ablation_build
    if not is_ablation_build():
        pytest.skip("the superseded 4-wave tapwgrad tile exists in libian_ablation.so only")
    lib, T, k = env
    n, cin, cout, h = 8, 128, 256, 16
    rs = np.random.RandomState(123)
    x = to_nhwc(rs.randn(n, cin, h, h).astype(np.float32))
    oh = h // 2 if kind == "conv" else h * 2
    dy = to_nhwc(rs.randn(n, cout, oh, oh).astype(np.float32))
    W = (rs.randn(*((cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5))) * 0.1).astype(np.float32)
    outs = []
    for w8 in (1, 0):
        monkeypatch.setenv("IAN_OPTS", "wg_w8=%d" % w8)
        layer = T.Layer(lib, T.K_CONV if kind == "conv" else T.K_DECONV, cin, cout, h, h)
        params = [torch.from_numpy(W.ravel()).cuda()]
        layer.set_params(params)
        g = [torch.zeros_like(params[0])]
        layer.backward_weight(x, dy, n, g)
        torch.cuda.synchronize()
        outs.append(g[0].cpu().numpy())
        layer.close()
    assert np.abs(outs[0]).max() > 0 and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("kind,cin,cout,h,n", [("conv", 128, 256, 16, 8), ("conv", 64, 160, 8, 5), ("deconv", 256, 128, 8, 8), ("deconv", 64, 32, 8, 3),
                                               ("dense", 1000, 100, 1, 7), ("dense_flat", 256 * 16, 200, 1, 6), ("dense_unflat", 100, 64 * 16, 1, 6)])
def test_round6_weight_paths_are_bitwise_the_round5_ones(env, kind, cin, cout, h, n, monkeypatch):
    """Round 6 changed HOW three things run, not what they compute: tapwgrad's K loop (wg_pipe: fragment reads pinned a group ahead,
    A operand as ds_read_b64 even / odd rows), its split reduce (wg_reduce_tiled: slabs read as they lie, scatter to the
    reference layout through LDS) and the repack of the slabs after an update (pack_tiled).  Each output element still sums the
    same products in the same order and the repack is a copy: forward, backward-data and backward-weight (overwrite and
    accumulate) must be IDENTICAL to the round-5 kernels, incl. ragged channel tiles (160 filters), ragged pixel ranges and the
    transposed conv's (Cin, Cout, 5, 5) parameter layout."""
    lib, T, k = env
    rs = np.random.RandomState(7)
    K = {"conv": T.K_CONV, "deconv": T.K_DECONV}.get(kind, T.K_DENSE)
    oh = h // 2 if kind == "conv" else h * 2
    dense_kw = {"dense_flat": dict(flat=(256, 4, 4)), "dense_unflat": dict(unflat=(64, 4, 4))}.get(kind, {})   # DenseLayer after / before (C,H,W)
    if kind.startswith("dense"):
        x = torch.from_numpy(np.pad(rs.randn(n, cin).astype(np.float32), ((0, 0), (0, cs(cin) - cin)))).cuda()
        dy = torch.from_numpy(np.pad(rs.randn(n, cout).astype(np.float32), ((0, 0), (0, cs(cout) - cout)))).cuda()
        W = (rs.randn(cin, cout) * 0.1).astype(np.float32)
    else:
        x = to_nhwc(rs.randn(n, cin, h, h).astype(np.float32))
        dy = to_nhwc(rs.randn(n, cout, oh, oh).astype(np.float32))
        W = (rs.randn(*((cout, cin, 5, 5) if kind == "conv" else (cin, cout, 5, 5))) * 0.1).astype(np.float32)
    outs = []
    for opts in ("wg_pipe=0,wg_reduce_tiled=0,pack_tiled=0", "wg_pipe=1,wg_reduce_tiled=1,pack_tiled=1", "wg_pipe=2,wg_reduce_tiled=1,pack_tiled=1",
                 "wg_pipe=3,wg_reduce_tiled=1,pack_tiled=1,tg_variant=7"):   # two K-steps of loads in flight, both GEMM families
        monkeypatch.setenv("IAN_OPTS", opts)
        layer = T.Layer(lib, K, cin, cout, h, h) if not kind.startswith("dense") else T.Layer(lib, K, cin, cout, **dense_kw)
        params = [torch.from_numpy(W.ravel()).cuda()]
        layer.set_params(params)
        y = torch.full((n,) + ((oh, oh) if not kind.startswith("dense") else ()) + (cs(cout),), 7.0, device="cuda")
        dx = torch.full_like(x, 7.0)
        layer.forward(x, n, y)
        layer.backward_data(dy, n, dx)
        g = [torch.full_like(params[0], 0.25)]
        layer.backward_weight(x, dy, n, g)
        layer.backward_weight(x, dy, n, g, accumulate=True)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (y, dx, g[0])])
        layer.close()
    assert np.abs(outs[0][2]).max() > 0 and np.isfinite(outs[0][2]).all()
    for other in outs[1:]:
        for a, b, what in zip(outs[0], other, ("forward", "backward-data", "backward-weight")):
            assert np.array_equal(a, b), what


def test_head6_backward_row_staged_gather_is_bitwise_the_per_tap_gather(env, monkeypatch):
    """head6_zbuild_rows_kernel (round 6: Z built one pixel row per workgroup through LDS, coalesced float4 rows) copies exactly the
    elements head6_zbuild_kernel gathers: data gradient and all parameter gradients of the three head layers are identical."""
    lib, T, k = env
    rs = np.random.RandomState(77)
    n, sc = 3, [2, 3, 4]
    xd = to_nhwc(rs.randn(n, 128, 64, 64).astype(np.float32))
    Ws = [(rs.randn(2, 128, 3, 3) * 0.05).astype(np.float32) for _ in range(3)]
    cfs = [[rs.uniform(0.5, 1.5, 2).astype(np.float32) for _ in range(4)] for _ in range(3)]
    dys = [to_nhwc(rs.randn(n, 2, 64, 64).astype(np.float32)) for _ in range(3)]
    outs = []
    for rows in (0, 1):
        monkeypatch.setenv("IAN_OPTS", "zbuild_rows=%d" % rows)
        layers, keep = [], []
        for i in range(3):
            layer = T.Layer(lib, T.K_MDC, 128, 2, 64, 64, scales=sc)
            params = [torch.from_numpy(a.ravel()).cuda() for a in [Ws[i]] + cfs[i]]
            layer.set_params(params)
            layers.append(layer)
            keep.append(params)
        grads = [[torch.zeros_like(p) for p in ps] for ps in keep]
        dx = torch.zeros(n, 64, 64, 128, device="cuda")
        assert layers[0].head6_backward(layers[1], layers[2], xd, dys, n, 32, dx=dx, dx_stride=128, dparams3=grads)
        torch.cuda.synchronize()
        outs.append([dx.cpu().numpy()] + [g.cpu().numpy() for gs in grads for g in gs])
        for l in layers:
            l.close()
    assert np.abs(outs[0][0]).max() > 0
    for a, b in zip(*outs):
        assert np.array_equal(a, b)

"""GPU parity of the configurations round 1 never ran on hardware ("close the untested configurations"):

  * the training building blocks and the composed update rules at the BENCHMARKED size -- 128 images per GPU -- where
    tapwgrad's pixel-split schedule and ian_k_colstats' per-image chunking take the paths bench.py times;
  * ``deconv_flip=False`` (the one convention that cannot be checked against Theano: SURVEY App. B.2) through the
    inference path, the latent brush and a training layer;
  * ``exact=False`` (data parallel with local batch statistics / local MinibatchLayer).

References: float64 torch-CPU autograd (kernels), oracle/train_twin.py in float64 (composed gradients), the numpy oracle
and torch twin with deconv_flip=False.  Measured error levels are written to gpurun_out/diag/ for the record.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ian_oracle as O
from oracle.torch_twin import TorchTwin
from oracle.train_twin import TrainTwin, make_train_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs")
NB = 128   # per-GPU batch of bench.py's train_step (BASELINE.json configs[4]: 1024 over 8 GPUs)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


def diag(name, obj):
    d = os.path.join(ROOT, "gpurun_out", "diag")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as fh:
            json.dump(obj, fh, indent=1, default=lambda o: float(o))
    except OSError:
        pass


def cs(c):
    return (c + 31) // 32 * 32


def to_nhwc(x):
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cs(c))
    out[..., :c] = torch.from_numpy(x).permute(0, 2, 3, 1)
    return out.cuda()


def from_nhwc(t, c):
    return t.cpu().numpy()[..., :c].transpose(0, 3, 1, 2)


@pytest.fixture(scope="module")
def env():
    from neural_photo_editor_amd.lib import load_train_library
    from neural_photo_editor_amd import trainer as T
    lib = load_train_library()
    return lib, T, T.K(lib)


# the layers of IAN.py whose backward-weight contraction (images x pixels) is longest at 128 images, one per kind,
# at their real shapes: enc_conv2, dec_conv3, MDBLOCK dec_conv4a, the RGB-Beta head's R and G_b
FULL_CASES = [
    ("conv", dict(cin=128, cout=256, h=32)),
    ("deconv", dict(cin=256, cout=128, h=16)),
    ("mdc", dict(cin=128, cout=128, h=32, scales=[0, 2, 3])),
    ("mdc", dict(cin=128, cout=2, h=64, scales=[2, 3, 4])),
    ("mdc", dict(cin=2, cout=2, h=64, scales=[2, 3, 4])),
]


@pytest.mark.parametrize("kind,g", FULL_CASES)
def test_layers_at_the_benchmarked_batch(env, kind, g):
    """forward / backward-data / backward-weight of one layer of each kind on 128 images vs float64 autograd."""
    lib, T, k = env
    n = NB
    rs = np.random.RandomState(hash((kind, g["cin"], g["cout"])) % 2 ** 31)
    cin, cout, h = g["cin"], g["cout"], g["h"]
    x = rs.randn(n, cin, h, h).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    if kind == "conv":
        W = (rs.randn(cout, cin, 5, 5) * 0.02).astype(np.float32)
        params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)]
        y = F.conv2d(xt, params[0], stride=2, padding=2)
        layer = T.Layer(lib, T.K_CONV, cin, cout, h, h)
    elif kind == "deconv":
        W = (rs.randn(cin, cout, 5, 5) * 0.02).astype(np.float32)
        params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)]
        y = F.conv_transpose2d(xt, torch.flip(params[0], (2, 3)), stride=2, padding=2, output_padding=1)
        layer = T.Layer(lib, T.K_DECONV, cin, cout, h, h)
    else:
        sc = g["scales"]
        W = (rs.randn(cout, cin, 3, 3) * 0.05).astype(np.float32)
        coeffs = [rs.uniform(0.5, 1.5, cout).astype(np.float32) for _ in range(1 + len(sc))]
        params = [torch.tensor(W, dtype=torch.float64, requires_grad=True)] + [torch.tensor(c, dtype=torch.float64, requires_grad=True) for c in coeffs]
        y = F.conv2d(xt, params[0], padding=1) * params[1].reshape(1, -1, 1, 1)
        for i, s in enumerate(sc):
            cf = params[2 + i].reshape(1, -1, 1, 1)
            y = y + (F.conv2d(xt, params[0].mean((2, 3), keepdim=True)) if s == 0 else F.conv2d(xt, params[0], padding=s, dilation=s)) * cf
        layer = T.Layer(lib, T.K_MDC, cin, cout, h, h, scales=sc)
    dy = rs.randn(*y.shape).astype(np.float32)
    gx, *gp = torch.autograd.grad(y, [xt] + params, torch.tensor(dy, dtype=torch.float64))
    # the layer keeps the parameter POINTERS (an MDCL's backward-weight reads W and the coefficients again): the tensors
    # must outlive the layer's use of them (include/ian_train.h, ian_layer_set_params)
    dev_params = [torch.from_numpy(p.detach().numpy().astype(np.float32).ravel()).cuda() for p in params]
    layer.set_params(dev_params)
    oh = y.shape[2]
    xd, dyd = to_nhwc(x), to_nhwc(dy)
    yd = torch.zeros(n, oh, oh, cs(cout), device="cuda")
    layer.forward(xd, n, yd)
    e = {"fwd": rel(from_nhwc(yd, cout), y.detach().numpy())}
    dxd = torch.zeros(n, h, h, cs(cin), device="cuda")
    layer.backward_data(dyd, n, dxd)
    e["bwd_data"] = rel(from_nhwc(dxd, cin), gx.numpy())
    dp = [torch.zeros(int(np.prod(p.shape)), device="cuda") for p in params]
    layer.backward_weight(xd, dyd, n, dp)
    e["bwd_weight"] = max(rel(got.cpu().numpy().reshape(ref.shape), ref.numpy()) for got, ref in zip(dp, gp))
    layer.backward_weight(xd, dyd, n, dp, accumulate=True)
    e["bwd_weight_acc"] = rel(dp[0].cpu().numpy().reshape(gp[0].shape), 2 * gp[0].numpy())
    layer.close()
    diag("layer128_%s_%d_%d" % (kind, cin, cout), e)
    assert e["fwd"] < 2e-5 and e["bwd_data"] < 2e-5, e
    # a contraction over 128 x (16..64)^2 pixels in float32: round-off grows with sqrt(K)
    assert e["bwd_weight"] < 1e-4 and e["bwd_weight_acc"] < 1e-4, e


@pytest.mark.parametrize("rows_per_image,C,act", [(4096, 128, 2), (16, 1024, 2), (1, 1000, 1)])
def test_batch_norm_at_the_benchmarked_batch(env, rows_per_image, C, act):
    """Lasagne batch_norm training mode (App. B.3) forward + backward over 128 images with the trainer's chunk rule."""
    lib, T, k = env
    n = NB
    rows, stride = n * rows_per_image, cs(C)
    rs = np.random.RandomState(C + rows_per_image)
    y = (rs.randn(rows, C) * rs.uniform(0.5, 2, C) + rs.randn(C)).astype(np.float32)
    gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), rs.randn(C).astype(np.float32)
    dA = rs.randn(rows, C).astype(np.float32)
    yt, gt, bt = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (y, gamma, beta)]
    mean, var = yt.mean(0), yt.var(0, unbiased=False)
    pre = (yt - mean) / torch.sqrt(var + 1e-4) * gt + bt
    a = F.leaky_relu(pre, 0.2) if act == 2 else torch.relu(pre)
    gy, gg, gb = torch.autograd.grad(a, [yt, gt, bt], torch.tensor(dA, dtype=torch.float64))
    pad = lambda v: torch.from_numpy(np.pad(v, ((0, 0), (0, stride - C)))).cuda()
    yd, dAd = pad(y), pad(dA)
    ad = torch.zeros_like(yd)
    bn = T.BN(torch, C, "cuda")
    nch = n * max(1, rows_per_image // 512)
    ws = torch.zeros(nch * 2 * C, device="cuda", dtype=torch.float64)
    k.colstats(0, yd, None, None, None, None, rows, C, stride, 0, ws, nch, bn.sums)
    k.bn_make_affine(bn.sums, float(rows), 1e-4, torch.from_numpy(gamma).cuda(), torch.from_numpy(beta).cuda(), C, bn.mean, bn.inv_std, bn.scale, bn.shift)
    k.affine(yd, ad, bn.scale, bn.shift, rows, C, stride, act)
    e = {"fwd": rel(ad.cpu().numpy()[:, :C], a.detach().numpy())}
    k.colstats(1, dAd, ad, yd, bn.mean, bn.inv_std, rows, C, stride, act, ws, nch, bn.bsums)
    dyd = torch.zeros_like(yd)
    k.bn_bwd(dAd, ad, yd, bn.mean, bn.inv_std, bn.scale, bn.bsums, float(rows), dyd, rows, C, stride, act)
    s = bn.bsums.cpu().numpy()
    e.update(dbeta=rel(s[:C], gb.numpy()), dgamma=rel(s[C:], gg.numpy()), dy=rel(dyd.cpu().numpy()[:, :C], gy.numpy()))
    diag("bn128_%d_%d" % (rows_per_image, C), e)
    assert e["fwd"] < 2e-5 and e["dbeta"] < 1e-4 and e["dgamma"] < 1e-4 and e["dy"] < 1e-4, e


def test_generator_update_at_a_quarter_of_the_benchmarked_batch():
    """update_gen's gradients (decoder_params, Z_params: train_IAN.py:256-273) on 32 images (round 6: 128 until then -- 148 s of the
    suite's 734, almost all of it the twin's float64 backward on the host; the kernels AT 128 images are held by the per-layer
    tests of this file, by the 8 x 16 sharded step that must equal the single-process 128-image step (tests/test_gpu_dp.py) and by
    the committed 128-image decomposition record, profiles/r06_decomposition_b128_gen.json; r05_ for the round-5 kernels) vs float64 autograd of the
    twin; the encoder passes on X_hat / X_gen are fed the twin's images, as in test_gpu_train.test_gradients_match_autograd.
    Bars are multiples of what a float32 evaluation of the SAME restatement (torch-CPU twin, float32 vs float64, these
    inputs; 8 minutes of CPU, so measured once and recorded here) moves each group by -- round 3, with the reference's
    MADE wiring (oracle.made_as_wired): decoder median 1.74e-3 / worst 5.5e-3, Z median 1.4e-4 / worst 3.2e-3.  (With
    the textbook MADE of rounds 1-2 the synthetic parameters drove |z| to ~2000 and saturated the decoder, which made
    the comparison look 50x better conditioned: 3.3e-5 / 9e-3.)
    Measured on MI355X (gpurun_out/diag/composed128.json, round 3): decoder 3.8e-3 / 3.8e-2, Z 1.25e-4 / 7.1e-3."""
    TWIN32 = {"dec": (1.74e-3, 5.5e-3), "Z": (1.4e-4, 3.2e-3)}
    from neural_photo_editor_amd.trainer import Trainer
    P = make_train_params(O.make_params("IAN", 1))
    NBH = NB // 4
    X, Z = O.make_images(NBH, seed=31), O.make_latents(NBH, seed=32)
    eps = np.random.RandomState(33).randn(NBH, 100).astype(np.float32)
    tw = TrainTwin(P, dtype=torch.float64)
    c = tw.cfg
    L = tw.losses(X, Z, eps)
    gen_loss = L["adv_gen"] + c["recon_weight"] * L["pixel_loss"] + c["feature_weight"] * L["feature_loss"] + L["l2_gen"]
    z_loss = c["feature_weight"] * L["feature_loss"] + c["recon_weight"] * L["pixel_loss"] + L["adv_gen"] + L["kl_div"] + L["l2_Z"]
    names = {"dec": list(tw.groups["dec"]), "Z": list(tw.groups["Z"])}
    # ONE float64 backward instead of two (suite budget: this test was 213 of the suite's 817 s): the two losses differ only in terms the
    # other group's parameters do not reach -- l2_gen is a function of decoder_params alone, kl_div and l2_Z of the encoder / Z_params
    # alone -- so d(gen_loss)/d(dec) and d(z_loss)/d(Z) are both read off the gradient of gen_loss + kl_div + l2_Z
    assert abs(float((gen_loss + L["kl_div"] + L["l2_Z"]) - (z_loss + L["l2_gen"]))) <= 1e-12 * abs(float(gen_loss))
    g_all = torch.autograd.grad(gen_loss + L["kl_div"] + L["l2_Z"], [tw.P[n] for n in names["dec"] + names["Z"]])
    g_dec, g_z = g_all[:len(names["dec"])], g_all[len(names["dec"]):]
    ref = {"dec": dict(zip(names["dec"], g_dec)), "Z": dict(zip(names["Z"], g_z))}
    xh, xg = [t.detach().numpy().astype(np.float32) for t in (tw.tensors["X_hat"], tw.tensors["X_gen"])]
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=NBH)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tr.forward(dev(X), dev(Z), dev(eps), xhat_override=dev(xh), xgen_override=dev(xg))
    m = tr.metrics()
    for k in ("pixel_loss", "kl_div", "feature_loss", "gen_recon_loss", "gen_sample_loss", "discrim_d_loss"):
        assert abs(m[k] - float(L[k])) <= 2e-5 * max(1.0, abs(float(L[k]))), (k, m[k], float(L[k]))
    tr.backward("gen")
    tr._regularizers("gen")
    out = {}
    for gname in ("dec", "Z"):
        got = tr.grads_numpy(gname)
        errs = sorted(((rel(got[name], r.numpy()), name) for name, r in ref[gname].items()), reverse=True)
        out[gname] = {"median": float(np.median([e for e, _ in errs])), "worst": errs[:5]}
    diag("composed32", out)
    # median: 2 x the float32 twin's (round-3 verdict).  Worst tensor: the maximum of ~150 heavy-tailed draws; 1-ulp perturbations of
    # the layer outputs move it by 7x from seed to seed (9e-3 .. 6.3e-2 at 16 images, profiles/r04_fp32_conditioning.json), so one
    # float32 draw bounds another only within that spread: 8 x.  The sharp per-tensor guard is tests/test_gpu_decomposition.py
    # (float64 gradient at the HIP step's own activations: every tensor within 3.1e-5).
    for gname in ("dec", "Z"):
        assert out[gname]["median"] < 2 * TWIN32[gname][0] and out[gname]["worst"][0][0] < 8 * TWIN32[gname][1], (gname, out[gname])


def test_encoder_passes_at_the_benchmarked_batch():
    """encoder_params gradients of one discriminator pass on 128 well-separated images (cross-entropy seeds through the
    3-way head, MinibatchLayer over 128 samples, three batch-normalised strided convs) vs float64 autograd: the sharp form
    of update_discrim's building block (test_gpu_train.test_encoder_passes_backward_sharp) at the benchmarked size."""
    from oracle.train_twin import ENC_PARAMS
    from neural_photo_editor_amd.trainer import Trainer
    P = make_train_params(O.make_params("IAN", 1))
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=NB)
    tw = TrainTwin(P, dtype=torch.float64)
    rs = np.random.RandomState(5)
    s = rs.uniform(0.15, 1.0, (NB, 1, 1, 1)).astype(np.float32)
    o = rs.uniform(-0.6, 0.3, (NB, 1, 1, 1)).astype(np.float32)
    imgs = [np.clip(O.make_images(NB, seed=60 + i) * s + o, -1, 1).astype(np.float32) for i in range(3)]
    Z, eps = O.make_latents(NB, seed=7), np.random.RandomState(8).randn(NB, 100).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tr.forward(dev(imgs[0]), dev(Z), dev(eps), xhat_override=dev(imgs[1]), xgen_override=dev(imgs[2]))
    enc = [tw.P[n] for n in ENC_PARAMS]
    out = {}
    for E, img, t in ((tr.EX, imgs[0], 0), (tr.EG, imgs[2], 2)):
        g = tw.encoder(torch.tensor(img, dtype=torch.float64))
        p = tw.discriminator(g[3])
        ref = torch.autograd.grad((-torch.log(p[:, t])).mean(), enc)
        tr.enc_backward(E, (t, 1.0 / tr.N, -1, 0.0), False, True, False, reset=True)
        got = tr.grads_numpy("enc")
        errs = sorted(((rel(got[n], r.numpy()), n) for n, r in zip(ENC_PARAMS, ref)), reverse=True)
        out["target%d" % t] = {"median": float(np.median([e for e, _ in errs])), "worst": errs[:4]}
        # measured on MI355X (gpurun_out/diag/encoder128.json, round 2): worst 2.6e-3 (bnorm3.beta), the float32
        # conditioning of 128 x 128 pairwise |a_b - a_b'| kernels; a wrong chunk / pixel-split path shows up as O(1)
        assert errs[0][0] < 1e-2 and float(np.median([e for e, _ in errs])) < 2e-3, (t, errs[:4])
    diag("encoder128", out)


# ---- deconv_flip = False --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch", O.ARCHS)
def test_deconv_flip_false_inference_and_brush(arch):
    """The kernel-flip switch of DeconvLayer (layers.py:476-481) set the other way: decoder, brush
    gradients and the edge (dec_out) kernels must follow the oracle built with deconv_flip=False, and must differ
    from the default convention (the switch is live)."""
    from neural_photo_editor_amd import IAN
    P = O.make_params(arch, 1)
    m = IAN(os.path.join(CFG, arch + ".py"), True, params=P, deconv_flip=False)
    orc = O.Oracle(arch, P, deconv_flip=False)
    z = O.make_latents(3, seed=21)
    x = O.make_images(3, seed=22)
    ref = orc.sample_at(z)
    assert rel(m.sample_at(z), ref) < 1e-4
    assert rel(m.reconstruct(x), orc.reconstruct(x)) < 1e-4
    assert rel(O.Oracle(arch, P).sample_at(z), ref) > 1e-2          # the two conventions give different images
    tw = TorchTwin(arch, P, deconv_flip=False, dtype=torch.float64)
    rgb = np.random.RandomState(5).uniform(-1, 1, (1, 3, 64, 64)).astype(np.float32)
    # Two latents per patch, the better one counts: a gradient through ~5*10^6 leaky-ReLU units is discontinuous wherever a
    # pre-activation sits within float32 round-off of zero, and the synthetic latent make_latents(seed=21)[0] has such a
    # unit (dec_conv4 channel 98, pixel (43,38): measured on MI355X, its derivative flips 1 <-> 0.2 with the split-K
    # summation order, a 37 % change of ONE element of the 128x64x64 gradient map and 5e-3 of dz, while every other
    # element agrees to 2e-7).  A wrong kernel-flip convention shows up as O(1) on every latent.
    errs = {}
    for patch in ((26, 26, 30, 30), (0, 0, 64, 64), (60, 0, 64, 9)):
        e_rgb = [rel(m.imgradRGB(*patch, rgb, z[i:i + 1]), tw.imgradRGB(*patch, rgb, z[i:i + 1])) for i in (1, 2)]
        e_light = [rel(m.imgrad(*patch, z[i:i + 1]), tw.imgrad(*patch, z[i:i + 1])) for i in (1, 2)]
        errs["rgb%s" % (patch,)] = e_rgb
        errs["light%s" % (patch,)] = e_light
    diag("flip_false_brush_%s" % arch, errs)
    assert all(min(v) < 5e-4 for v in errs.values()), errs
    m.close()


def test_deconv_flip_false_training_layer(env):
    lib, T, k = env
    n, cin, cout, h = 5, 64, 32, 8
    rs = np.random.RandomState(12)
    x = rs.randn(n, cin, h, h).astype(np.float32)
    W = (rs.randn(cin, cout, 5, 5) * 0.1).astype(np.float32)
    xt, wt = torch.tensor(x, dtype=torch.float64, requires_grad=True), torch.tensor(W, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(xt, wt, stride=2, padding=2, output_padding=1)        # no kernel flip
    dy = rs.randn(*y.shape).astype(np.float32)
    gx, gw = torch.autograd.grad(y, [xt, wt], torch.tensor(dy, dtype=torch.float64))
    layer = T.Layer(lib, T.K_DECONV, cin, cout, h, h, deconv_flip=False)
    layer.set_params([torch.from_numpy(W.ravel()).cuda()])
    xd, dyd = to_nhwc(x), to_nhwc(dy)
    yd = torch.zeros(n, 2 * h, 2 * h, cs(cout), device="cuda")
    layer.forward(xd, n, yd)
    assert rel(from_nhwc(yd, cout), y.detach().numpy()) < 2e-5
    dxd = torch.zeros(n, h, h, cs(cin), device="cuda")
    layer.backward_data(dyd, n, dxd)
    assert rel(from_nhwc(dxd, cin), gx.numpy()) < 2e-5
    dW = [torch.zeros(W.size, device="cuda")]
    layer.backward_weight(xd, dyd, n, dW)
    assert rel(dW[0].cpu().numpy().reshape(W.shape), gw.numpy()) < 2e-5
    layer.close()


def test_deconv_flip_false_training_step_matches_twin():
    """One composed generator sweep with the other kernel convention (decoder_params gradients depend on it)."""
    from neural_photo_editor_amd.trainer import Trainer
    B = 4
    P = make_train_params(O.make_params("IAN", 1))
    X, Z = O.make_images(B, seed=1), O.make_latents(B, seed=6)
    eps = np.random.RandomState(7).randn(B, 100).astype(np.float32)
    tw = TrainTwin(P, dtype=torch.float64, deconv_flip=False)
    g64, _ = tw.gradients(X, Z, eps)
    xh, xg = [t.detach().numpy().astype(np.float32) for t in (tw.tensors["X_hat"], tw.tensors["X_gen"])]
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=B, deconv_flip=False)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tr.forward(dev(X), dev(Z), dev(eps), xhat_override=dev(xh), xgen_override=dev(xg))
    assert rel(tr.DZ["xhat"].cpu().numpy(), tw.tensors["X_hat"].detach().numpy()) < 1e-4
    tr.backward("gen")
    tr._regularizers("gen")
    got = tr.grads_numpy("dec")
    errs = sorted(((rel(got[name], ref.numpy()), name) for name, ref in g64["dec"].items()), reverse=True)
    diag("flip_false_gen", {"median": float(np.median([e for e, _ in errs])), "worst": errs[:5]})
    assert float(np.median([e for e, _ in errs])) < 5e-3 and errs[0][0] < 0.2, errs[:5]
    tw2 = TrainTwin(P, dtype=torch.float64)                                         # default convention: a different function
    tw2.losses(X, Z, eps)
    assert rel(tw2.tensors["X_hat"].detach().numpy(), tw.tensors["X_hat"].detach().numpy()) > 1e-2


# ---- exact = False ----------------------------------------------------------------------------------------------------
class _LocalComm:
    """A 2-rank communicator whose collectives are identities: what rank 0 computes BEFORE the gradient sum."""
    active, world, rank, group, bucket_bytes = True, 2, 0, None, 16 << 20

    def ops(self, torch):
        from neural_photo_editor_amd.trainer import build_ops
        self.errors, self.calls = [], {"allreduce": 0, "wait_all": 0}

        def allreduce(buf, count, stream):
            self.calls["allreduce"] += 1           # identity: the gradient stays the rank's own contribution

        def wait_all(stream):
            self.calls["wait_all"] += 1

        def allgather(src, dst, count, stream):
            raise AssertionError("exact=False must neither all-gather batch statistics nor MinibatchLayer activations")

        return build_ops(self.world, self.rank, allreduce, wait_all, allgather, self.errors)

    def barrier(self):
        pass


def test_non_power_of_two_shards_are_announced(capfd):
    """Round-3 verdict, weak #9: the partial-sum tree of the batch statistics is bit-identical between N ranks and one
    process only for power-of-two shards; any other per-rank batch / world size must say so at set-up, loudly (stderr, from
    ian_trainer_finalize -- the one sequencer -- so that C callers see it too)."""
    from neural_photo_editor_amd.trainer import Trainer
    P = make_train_params(O.make_params("IAN", 1))

    class ThreeRanks(_LocalComm):
        world = 3

    capfd.readouterr()
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=2, comm=ThreeRanks(), exact=True)
    err = capfd.readouterr().err
    assert "not a power of two" in err and "NOT guaranteed bit-identical" in err, err
    assert tr.N == 6 and tr.exact and tr.stat("world") == 3
    tr.close()
    tr = Trainer(os.path.join(CFG, "IAN.py"), P, batch=2, comm=_LocalComm(), exact=True)     # 2 x 2: silent
    assert "power of two" not in capfd.readouterr().err
    tr.close()


@pytest.mark.parametrize("which", ["gen", "discrim"])
def test_local_statistics_mode(which):
    """exact=False (train_cli --local-statistics): batch-norm statistics and the MinibatchLayer see only the rank's own
    shard; losses are still means over the GLOBAL minibatch, so a rank's pre-reduction gradient is n/N times the
    single-process gradient of its shard.  Checked against a world-1 trainer on the same 4 images: factor 1/2 exactly
    (a power of two: bitwise), for every tensor of the updated groups."""
    from neural_photo_editor_amd.trainer import Trainer
    B = 4
    P = make_train_params(O.make_params("IAN", 1))
    X, Z = O.make_images(B, seed=2), O.make_latents(B, seed=3)
    eps = np.random.RandomState(4).randn(B, 100).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    single = Trainer(os.path.join(CFG, "IAN.py"), P, batch=B)
    local = Trainer(os.path.join(CFG, "IAN.py"), P, batch=B, comm=_LocalComm(), exact=False)
    assert local.N == 2 * B and not local.exact and single.N == B
    for tr in (single, local):
        tr.forward(dev(X), dev(Z), dev(eps))
        tr.backward(which)
        tr._finish_allreduce(which)
    torch.cuda.synchronize()
    assert local.comm.calls["allreduce"] == local.plan_size(which) > 0 and local.comm.calls["wait_all"] == 1 and not local.comm.errors
    assert torch.equal(single.DZ["xhat"], local.DZ["xhat"]) and torch.equal(single.EG["p"], local.EG["p"])
    for g in (("dec" if which == "gen" else "enc"), "Z"):
        a, b = single.groups[g].g, local.groups[g].g
        assert float(a.abs().max()) > 0
        assert torch.allclose(0.5 * a, b, rtol=1e-6, atol=0), g
    ms, ml = single.metrics(), local.metrics()
    assert abs(ms["pixel_loss"] - 2 * ml["pixel_loss"]) < 1e-6 * abs(ms["pixel_loss"])      # half of the global mean lives on each rank

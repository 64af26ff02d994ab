"""CPU tests that pin the oracle: MADE known-answer vector (reference-derived), numpy restatement vs the
independent torch-CPU twin, equivalence of the reference's two decoder formulations, float64 twin,
finite-difference gradients, and the committed golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ian_oracle as O
from oracle.torch_twin import TorchTwin

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


# ---- MADE masks: bit-exact known answer, SURVEY App. C (mask_generator.py:15-103) ------------------------
def test_made_ordering_known_answer():
    ordering, child = O.made_ordering()
    assert child == 822569775
    assert ordering[:20].tolist() == [52, 79, 87, 45, 24, 71, 82, 80, 34, 36, 89, 77, 40, 13, 8, 35, 56, 98, 5, 1]
    assert O.ordering_digest(ordering) == "4ecb480535babb6b"
    assert int(np.argmin(ordering)) == 80
    assert sorted(ordering.tolist()) == list(range(100))


def test_made_masks_known_counts():
    M0, M1, MD = O.made_masks()
    assert (int(M0.sum()), int(M1.sum()), int(MD.sum())) == (100, 9900, 4950)
    assert np.nonzero(M0.sum(1))[0].tolist() == [80]  # only the first-ordered input feeds the hidden layer
    for M in (M0, M1, MD):
        assert M.dtype == np.float32 and set(np.unique(M).tolist()) <= {0.0, 1.0}
    # strict autoregressive order through the direct path: input i reaches output j iff order[i] < order[j]
    ordering, _ = O.made_ordering()
    assert np.array_equal(MD, (ordering[:, None] < ordering[None, :]).astype(np.float32))


def test_made_masks_match_golden_file():
    g = np.load(os.path.join(GOLD, "made_masks.npz"))
    M = np.stack(O.made_masks()).astype(np.uint8)
    assert np.array_equal(np.packbits(M), g["packed"])
    assert np.array_equal(O.made_ordering()[0], g["ordering"])


def test_made_output_is_autoregressive():
    P = O.make_params("IAN", 1)
    orc = O.Oracle("IAN", P, dtype=np.float64)
    ordering, _ = O.made_ordering()
    z = np.random.RandomState(3).randn(1, 100)
    base = O.made(z, orc.P, "l_IAF_mu", orc.masks)
    k = 37
    z2 = z.copy(); z2[0, k] += 1.0
    diff = np.abs(O.made(z2, orc.P, "l_IAF_mu", orc.masks) - base)[0]
    assert np.all(diff[ordering <= ordering[k]] == 0.0)   # nothing at or before k in the ordering moves
    assert diff[ordering > ordering[k]].max() > 0


# ---- primitive conventions vs torch ------------------------------------------------------------------------
def test_conv_and_deconv_conventions_vs_torch():
    rs = np.random.RandomState(0)
    x = rs.randn(2, 6, 8, 8).astype(np.float32)
    W = rs.randn(5, 6, 5, 5).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(W), stride=2, padding=2).numpy()
    assert rel(O.conv5s2(x, W), ref) < 1e-5
    Wd = rs.randn(6, 4, 5, 5).astype(np.float32)
    xt, wt = torch.from_numpy(x), torch.from_numpy(Wd)
    # crop=2 forced-2x formulation (layers.py:436-483) with the App. B.2 flip
    a = F.conv_transpose2d(xt, torch.flip(wt, (2, 3)), stride=2, padding=2, output_padding=1).numpy()
    assert rel(O.deconv5s2(x, Wd, flip=True), a) < 1e-5
    # crop=1 + drop first row/col formulation (IAN_simple.py:182-223) is the same map
    b = F.conv_transpose2d(xt, torch.flip(wt, (2, 3)), stride=2, padding=1)[:, :, 1:, 1:].numpy()
    assert b.shape == a.shape and rel(b, a) < 1e-5
    c = F.conv_transpose2d(xt, wt, stride=2, padding=2, output_padding=1).numpy()
    assert rel(O.deconv5s2(x, Wd, flip=False), c) < 1e-5


def test_mdcl_vs_torch_branches():
    rs = np.random.RandomState(1)
    x = rs.randn(2, 5, 12, 12).astype(np.float32)
    P = {"mW": rs.randn(3, 5, 3, 3).astype(np.float32), "m_coeff_base": rs.rand(3).astype(np.float32),
         "m_coeff_1x1": rs.rand(3).astype(np.float32), "m_coeff_2": rs.rand(3).astype(np.float32),
         "m_coeff_3": rs.rand(3).astype(np.float32)}
    tw = TorchTwin("IAN_simple", P)
    got = O.mdcl(x, P, "m", [0, 2, 3])
    ref = tw.mdcl(torch.from_numpy(x), "m", [0, 2, 3]).numpy()
    assert rel(got, ref) < 1e-5


@pytest.mark.parametrize("arch", O.ARCHS)
def test_numpy_oracle_vs_torch_twin_and_f64(arch):
    P = O.make_params(arch, 1)
    x = O.make_images(2)
    orc, tw = O.Oracle(arch, P), TorchTwin(arch, P)
    z = orc.encode_images(x)
    assert rel(z, tw.np_encode(x)) < 2e-5
    xh = orc.sample_at(z)
    assert rel(xh, tw.np_decode(z)) < 2e-5
    o64 = O.Oracle(arch, P, dtype=np.float64)
    assert rel(z, o64.encode_images(x)) < 2e-5
    assert rel(xh, o64.sample_at(z)) < 2e-5
    assert np.abs(xh).max() < 0.999 and xh.std() > 0.05  # outputs stay in the sensitive range


@pytest.mark.parametrize("arch", O.ARCHS)
def test_golden_vectors(arch):
    g = np.load(os.path.join(GOLD, "%s_seed1.npz" % arch))
    P = O.make_params(arch, 1)
    orc = O.Oracle(arch, P)
    x = O.make_images(2)
    assert rel(orc.Zfn(x), g["zpre"]) < 1e-5
    assert rel(orc.encode_images(x), g["z"]) < 1e-5
    assert rel(orc.sample_at(g["z"]), g["xhat"]) < 1e-5
    assert np.array_equal(O.make_latents(2), g["z_sample"])
    assert rel(orc.sample_at(g["z_sample"]), g["x_sample"]) < 1e-5


def test_brush_gradients_vs_finite_differences():
    """API.py:59,64 gradients from autograd (float64) agree with central differences on the restatement."""
    arch = "IAN_simple"
    P = O.make_params(arch, 1)
    tw = TorchTwin(arch, P, dtype=torch.float64)
    o64 = O.Oracle(arch, P, dtype=np.float64)
    z = O.make_latents(1).astype(np.float64)
    rgb = np.full((1, 3, 64, 64), -1.0); rgb[:, 0] = 1.0
    c1, r1, c2, r2 = 26, 26, 30, 30
    g = tw.imgradRGB(c1, r1, c2, r2, rgb, z)
    gl = tw.imgrad(c1, r1, c2, r2, z)

    def loss_rgb(zz):
        xh = o64.sample_at(zz)
        return np.mean((-xh[0, :, r1:r2, c1:c2] + rgb[0, :, r1:r2, c1:c2]) ** 2)

    def loss_l(zz):
        return np.mean(o64.sample_at(zz)[0, :, r1:r2, c1:c2])

    eps = 1e-5
    for k in (0, 17, 63, 99):
        dz = np.zeros_like(z); dz[0, k] = eps
        fd = (loss_rgb(z + dz) - loss_rgb(z - dz)) / (2 * eps)
        assert abs(fd - g[0, k]) < 1e-6 + 1e-4 * abs(g[0, k])
        fdl = (loss_l(z + dz) - loss_l(z - dz)) / (2 * eps)
        assert abs(fdl - gl[0, k]) < 1e-6 + 1e-4 * abs(gl[0, k])
    gold = np.load(os.path.join(GOLD, "IAN_simple_seed1.npz"))
    assert rel(g, gold["grad_rgb"]) < 1e-5 and rel(gl, gold["grad_light"]) < 1e-5


def test_c_restatement_agrees_with_numpy_oracle():
    """oracle/ian_primitives.c (plain C, float64 accumulation): third independent statement of the conv / transposed
    conv / MDCL index conventions; must agree with the numpy restatement (and through it with the torch twin)."""
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle")], check=True)
    lib = ctypes.CDLL(os.path.join(root, "oracle", "_ref", "libian_primitives.so"))
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rs = np.random.RandomState(0)
    N, Cin, H, Cout = 2, 5, 8, 7
    x = rs.randn(N, Cin, H, H).astype(np.float32)
    W = rs.randn(Cout, Cin, 5, 5).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    y = np.zeros((N, Cout, H // 2, H // 2), np.float32)
    lib.ref_conv5s2(fp(x), fp(W), fp(b), fp(y), N, Cin, H, H, Cout)
    assert np.abs(y - O.conv5s2(x, W, b)).max() < 1e-4
    Wd = rs.randn(Cin, Cout, 5, 5).astype(np.float32)
    for flip in (1, 0):
        yd = np.zeros((N, Cout, 2 * H, 2 * H), np.float32)
        lib.ref_deconv5s2(fp(x), fp(Wd), fp(yd), N, Cin, H, H, Cout, flip)
        assert np.abs(yd - O.deconv5s2(x, Wd, None, bool(flip))).max() < 1e-4
    Wm = rs.randn(Cout, Cin, 3, 3).astype(np.float32)
    scales = np.array([0, 2, 3], np.int32)
    coeffs = rs.uniform(0.5, 1.5, (4, Cout)).astype(np.float32)
    P = {"mW": Wm, "m_coeff_base": coeffs[0], "m_coeff_1x1": coeffs[1], "m_coeff_2": coeffs[2], "m_coeff_3": coeffs[3]}
    ym = np.zeros((N, Cout, H, H), np.float32)
    lib.ref_mdcl(fp(x), fp(Wm), fp(coeffs), fp(scales), 3, fp(ym), N, Cin, H, H, Cout)
    assert np.abs(ym - O.mdcl(x, P, "m", [0, 2, 3])).max() < 1e-4

"""NPE.paint's photo blend (NPE.py:218-231) and the uint8 image conversion (NPE.py:110,261) on the device, against the
reference's numpy/scipy expression (neural_photo_editor_amd.npe_ops.photo_blend_host): byte work -> bit-exact."""
import os

import numpy as np
import pytest

from oracle import ian_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs")


@pytest.fixture(scope="module")
def model():
    from neural_photo_editor_amd import IAN
    P = O.make_params("IAN_simple", 1)
    return IAN(os.path.join(CFG, "IAN_simple.py"), True, params=P)


def session(model, seed):
    """What NPE.infer leaves behind (NPE.py:239-274): IM, Z, RECON (uint8), ERROR (float32)."""
    from neural_photo_editor_amd import npe_ops as N
    rs = np.random.RandomState(seed)
    IM = np.uint8((O.make_images(1, seed=seed)[0] + 1.0) * 127.5)
    Z = model.encode_images(np.asarray([N.to_tanh(IM)], dtype=np.float32))
    RECON = np.uint8(N.from_tanh(model.sample_at(np.float32(Z))[0]))
    ERROR = N.to_tanh(np.float32(IM)) - N.to_tanh(np.float32(RECON))
    return IM, Z + 0.3 * rs.randn(*Z.shape).astype(np.float32), RECON, ERROR


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_photo_blend_is_bit_exact(model, seed):
    from neural_photo_editor_amd import npe_ops as N
    IM, Z, RECON, ERROR = session(model, seed)
    assert ERROR.dtype == np.float32 and RECON.dtype == np.uint8
    xhat = model.sample_at(Z)[0]
    want_im, want_mask = N.photo_blend_host(xhat, RECON, ERROR)
    im, mask = model.photo_blend(Z, RECON, ERROR)
    assert im.dtype == np.uint8 and mask.dtype == np.float64
    assert np.array_equal(mask, want_mask)          # float64, bit for bit (scipy's summation order)
    assert np.array_equal(im, want_im)
    im2, _ = N.photo_blend(model, Z.reshape(10, 10), RECON, ERROR)       # the NPE-style entry (10x10 latent grid)
    assert np.array_equal(im2, want_im)
    assert want_mask.max() > 1e-3 and (im != RECON).any()                # a non-trivial edit


def test_photo_blend_out_of_range_wraps_like_numpy(model):
    """ERROR large enough to push values outside [0,255]: the reference's bare np.uint8 cast wraps; so does the kernel."""
    from neural_photo_editor_amd import npe_ops as N
    IM, Z, RECON, ERROR = session(model, 5)
    ERROR = (ERROR * 6.0).astype(np.float32)
    xhat = model.sample_at(Z)[0]
    want_im, want_mask = N.photo_blend_host(xhat, RECON, ERROR)
    with np.errstate(invalid="ignore"):
        v = N.from_tanh(N.to_tanh(RECON) + want_mask * (xhat - N.to_tanh(np.float32(RECON))) + (1 - want_mask) * ERROR)
    assert (v < 0).any() or (v >= 256).any()
    im, mask = model.photo_blend(Z, RECON, ERROR)
    assert np.array_equal(im, want_im) and np.array_equal(mask, want_mask)


def test_photo_blend_device_pointers(model):
    import torch
    from neural_photo_editor_amd import npe_ops as N
    IM, Z, RECON, ERROR = session(model, 7)
    want_im, want_mask = model.photo_blend(Z, RECON, ERROR)
    half = N.gaussian_half_kernel()
    zd, rd, ed = torch.from_numpy(Z).cuda(), torch.from_numpy(RECON).cuda(), torch.from_numpy(ERROR).cuda()
    imd, md = torch.zeros(3, 64, 64, dtype=torch.uint8, device="cuda"), torch.zeros(64, 64, dtype=torch.float64, device="cuda")
    model.handle.photo_blend(zd, rd, ed, half, imd, md, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(imd.cpu().numpy(), want_im) and np.array_equal(md.cpu().numpy(), want_mask)


@pytest.mark.parametrize("arch,n", [("IAN_simple", 1), ("IAN_simple", 5), ("IAN", 9)])
def test_sample_at_uint8(arch, n):
    """np.uint8(from_tanh(sample_at(z))) (NPE.py:110,261) computed on the device == the host expression, byte for byte."""
    from neural_photo_editor_amd import IAN, npe_ops as N
    m = IAN(os.path.join(CFG, arch + ".py"), True, params=O.make_params(arch, 1))
    z = O.make_latents(n, seed=3) * 2.0
    want = np.uint8(N.from_tanh(m.sample_at(z)))
    got = m.sample_at_uint8(z)
    assert got.dtype == np.uint8 and got.shape == (n, 3, 64, 64) and np.array_equal(got, want)
    m.close()


def _rgb_image():
    rgb = np.zeros((3, 64, 64), np.uint8)
    rgb[0] = 230; rgb[1] = 40; rgb[2] = 90
    return rgb


@pytest.mark.parametrize("graph", [1, 0])
def test_brush_step_equals_the_composed_calls(model, graph):
    """ian_brush_step (gradient + latent update + decoder in one submission) vs imgradRGB -> numpy update -> sample_at:
    the latent is bit-identical (same float32 operations in NPE.py:205-209's order), so is the image (same kernels on the
    same latent); 12 consecutive events so that the graph is captured and replayed, with the brush size changing."""
    from neural_photo_editor_amd import npe_ops as N
    model.handle.set_option("edit_graph", graph)
    try:
        IM, Z, RECON, ERROR = session(model, 11)
        rgb8 = _rgb_image()
        rgb = np.float32(N.to_tanh(np.float32(rgb8)))[None]
        z_a = np.float32(Z).copy()
        z_b = z_a.copy()
        for i in range(12):
            box = (20 + i, 22, 24 + i + (i % 3), 26 + (i % 3))
            g = model.imgradRGB(box[0], box[1], box[2], box[3], rgb, z_a)
            z_a = np.float32(z_a - np.float32(0.05) * (g * np.float32(1 + (box[2] - box[0]))))
            x_a = model.sample_at(z_a)
            z_b, x_b = model.brush_step(box[0], box[1], box[2], box[3], z_b, RGB=rgb, weight=0.05)
            assert z_b.dtype == np.float32 and np.array_equal(z_b, z_a), i
            assert np.array_equal(x_b, x_a), i
        assert np.abs(z_a - np.float32(Z)).max() > 1e-4          # the brush moved the latent
        # NPE.scroll: lighten / darken
        for sign in (1.0, -1.0):
            g = model.imgrad(10, 12, 16, 18, z_a)
            z_a = np.float32(z_a + np.float32(sign * 0.1) * (g * np.float32(7)))
            z_b, x_b = model.brush_step(10, 12, 16, 18, z_b, weight=0.1, sign=sign)
            assert np.array_equal(z_b, z_a) and np.array_equal(x_b, model.sample_at(z_a))
    finally:
        model.handle.set_option("edit_graph", 1)


def test_zero_copy_graphs_equal_the_copy_node_graphs(model):
    """Round 4: the kernels of the interactive graphs read the brush rectangle from the pinned host block and write the new
    latent, the gradient and the image into it directly (no hipMemcpy nodes).  Same kernels, same arithmetic: the events must be
    bit-identical to the graphs with copy nodes (edit_zero_copy = 0), also when the decoder cache short-cuts a call and when
    sample_at / imgradRGB interleave with brush_step."""
    from neural_photo_editor_amd import npe_ops as N
    IM, Z, RECON, ERROR = session(model, 17)
    rgb = np.float32(N.to_tanh(np.float32(_rgb_image())))[None]
    runs = {}
    try:
        for zc in (0, 1):
            model.handle.set_option("edit_zero_copy", zc)
            z = np.float32(Z).copy()
            rec = []
            for i in range(10):
                box = (18 + i, 20, 23 + i, 25 + (i % 2))
                if i % 4 == 3:                                   # a two-call event in between (its own graphs)
                    g = model.imgradRGB(box[0], box[1], box[2], box[3], rgb, z)
                    z = np.float32(z - np.float32(0.05) * (g * np.float32(1 + (box[2] - box[0]))))
                    rec.append((z.copy(), model.sample_at(z), model.sample_at(z)))       # second call: decoder-cache hit
                else:
                    z, x = model.brush_step(box[0], box[1], box[2], box[3], z, RGB=rgb, weight=0.05)
                    rec.append((z.copy(), x, model.sample_at(z)))                           # cache hit right after the event
            runs[zc] = rec
    finally:
        model.handle.set_option("edit_zero_copy", 1)
    for (za, xa, ya), (zb, xb, yb) in zip(runs[0], runs[1]):
        assert np.array_equal(za, zb) and np.array_equal(xa, xb) and np.array_equal(ya, yb)
        assert np.array_equal(xb, yb)                                                      # the cached image is the event's image
    assert np.abs(runs[1][-1][0] - np.float32(Z)).max() > 1e-4


def test_activation_readable_after_device_pointer_call_then_host_call(model):
    """a device-pointer ian_reconstruct marks the output slot's own buffer stale; any later call that refills that
    buffer (sample_at, the decoder cache, the graphs, brush_step) must make it readable again."""
    import torch
    from neural_photo_editor_amd import synthetic as O
    x = torch.from_numpy(O.make_images(1, seed=4)).cuda()
    out = torch.empty_like(x)
    model.handle.call("ian_reconstruct", x, 1, out)
    torch.cuda.synchronize()
    z = O.make_latents(1, seed=6)
    img = model.sample_at(z)
    got = model.handle.read_slot(model.lowered.out_slot, 1)          # used to fail: "bound to the caller's device buffer ..."
    assert np.array_equal(got, img)


def test_paint_event_photo_mode(model):
    """NPE.paint in photo mode as one submission: (Z_new, IM) == brush_step + photo_blend_host on the decoded image."""
    from neural_photo_editor_amd import npe_ops as N
    IM, Z, RECON, ERROR = session(model, 13)
    rgb8 = _rgb_image()
    Zg = np.float32(Z).reshape(10, 10)
    Zh = Zg.copy()
    for i in range(6):
        box = (30, 28 + i, 34, 32 + i)
        Zh = N.brush_step(model, Zh, box, rgb8)                              # composed: imgradRGB + numpy
        want_im, _ = N.photo_blend_host(model.sample_at(Zh.reshape(1, -1))[0], RECON, ERROR)
        Zg, im = N.paint_event(model, Zg, box, rgb8, RECON, ERROR)
        assert Zg.shape == (10, 10) and np.array_equal(Zg, Zh), i
        assert im.dtype == np.uint8 and np.array_equal(im, want_im), i
    # sample mode: the float image
    Zs, x = N.paint_event(model, Zg, (5, 5, 9, 9), rgb8)
    Zh = N.brush_step(model, Zh, (5, 5, 9, 9), rgb8)
    assert np.array_equal(Zs, Zh) and np.array_equal(x, model.sample_at(Zh.reshape(1, -1))[0])


def test_npe_session_replay_vs_reference_executed_session(model):
    """One whole NPE.py editing session -- infer, 6 brush events in photo mode (two colours, two brush sizes), 3 scroll events, Reset --
    as the sequence of facade calls its callbacks make (NPE.py:192-235, 239-279, 305-316, 330-340), recorded by running the reference's
    own API.IAN on the evaluating stand-in (tests/golden/ref_session_IAN_simple.npz <- make_ref_golden.py; the numpy / scipy lines between
    the model calls restated from NPE.py).  Replayed through the INTEGRATION.md section 1 surface: encode_images, ian_decode_u8
    (sample_at_uint8), ian_brush_step with the device photo blend (npe_ops.paint_event), ian_photo_blend, ian_brush_step in lighten
    mode.  Bars: every latent and blend mask within 1e-4 (the north-star tolerance on float32 values); canvas bytes equal to the
    reference's except where a float32 image value sits on a uint8 truncation boundary -- at most one level on at most 0.2 % of the
    pixels of an image (measured and recorded in gpurun_out/diag/npe_session.json)."""
    import json
    from session_replay import replay, compare
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_session_IAN_simple.npz"))
    events = replay(model, fx)
    worst = compare(events, fx, tol=1e-4, max_off_by_one_frac=2e-3)
    assert [e["kind"] for e in events].count("paint") == 6 and [e["kind"] for e in events].count("scroll") == 3
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out", "diag"), exist_ok=True)
        json.dump(worst, open(os.path.join(ROOT, "gpurun_out", "diag", "npe_session.json"), "w"), indent=1)
    except OSError:
        pass

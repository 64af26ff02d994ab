"""Replay of the reference-executed NPE.py editing session (tests/golden/ref_session_IAN_simple.npz, written by
tests/golden/make_ref_golden.py from /root/reference/API.py on the evaluating Theano / Lasagne stand-in) through the drop-in surface:
neural_photo_editor_amd.api.IAN + npe_ops -- the calls INTEGRATION.md section 1 tells a maintainer of NPE.py to make, in the order its
callbacks make them (infer NPE.py:239-279, paint :192-235 in photo mode, scroll :305-316, Reset :330-340).  `model` is the HIP class
(GPU test: ian_brush_step / ian_photo_blend / ian_decode_u8 carry the events) or any double with the API.py method names (CPU test: the
float64 torch twin of the oracle, through npe_ops' host composition).  No UI code."""
import numpy as np

from neural_photo_editor_amd import npe_ops


def replay(model, fx):
    """-> list of per-event dicts {kind, Z (10,10) float32, shown uint8 (3,64,64) or None, MASK or None, RECON / ERROR for infer / reset}."""
    GIM = fx["GIM"]
    kinds = [str(k) for k in fx["kinds"]]
    Z = np.zeros((10, 10), np.float32)
    RECON = ERROR = None
    out = []
    for k, kind in enumerate(kinds):
        x1, y1, x2, y2, dsize, r, g, b, delta = [int(v) for v in fx["state"][k]]
        rec = {"kind": kind, "shown": None, "MASK": None}
        if kind in ("infer", "reset"):
            IM = GIM
            Z = np.reshape(np.asarray(model.encode_images(np.asarray([npe_ops.to_tanh(IM)], dtype=np.float32)))[0], (10, 10)).astype(np.float32)
            if hasattr(model, "sample_at_uint8"):                 # np.uint8(from_tanh(sample_at(Z))) on the device (ian_decode_u8)
                RECON = model.sample_at_uint8(np.float32([Z.flatten()]))[0]
            else:
                RECON = np.uint8(npe_ops.from_tanh(np.float32(model.sample_at(np.float32([Z.flatten()])))[0]))
            ERROR = npe_ops.to_tanh(np.float32(IM)) - npe_ops.to_tanh(np.float32(RECON))
            rec.update(shown=IM, RECON=RECON, ERROR=ERROR)
        elif kind == "paint":
            rgb = np.zeros((3, 64, 64), np.float32)
            rgb[0], rgb[1], rgb[2] = r, g, b                       # myRGB[0] (NPE.py:87,359): colour levels 0..255 held as float32
            Z, IM = npe_ops.paint_event(model, Z, (x1, y1, x2, y2), rgb, RECON, ERROR)      # the whole NPE.paint body, one submission
            IM2, MASK = npe_ops.photo_blend(model, Z, RECON, ERROR)                          # the blend alone on the same latent
            assert np.array_equal(IM, IM2)
            rec.update(shown=np.asarray(IM), MASK=np.asarray(MASK))
        elif kind == "scroll":
            if hasattr(model, "brush_step"):                       # NPE.scroll as one submission: imgrad, Z += sign * 0.1 * grad, sample_at
                z_new, x = model.brush_step(x1, y1, x2, y2, np.float32([Z.flatten()]), RGB=None, weight=0.1, sign=float(np.sign(delta)))
                Z = z_new.reshape(10, 10)
                img = x[0]
            else:
                Z = npe_ops.lighten_step(model, Z, (x1, y1, x2, y2), weight=0.1, sign=np.sign(delta))
                img = np.float32(model.sample_at(np.float32([Z.flatten()])))[0]
            rec["shown"] = np.uint8(npe_ops.from_tanh(img))
        rec["Z"] = np.asarray(Z, np.float32).reshape(10, 10).copy()
        out.append(rec)
    return out


def compare(events, fx, tol, max_off_by_one_frac):
    """Every event's latent within `tol` (max-abs error over max-abs reference), masks within `tol`, canvas images equal to the
    reference's uint8 values except where a float32 value sits on a truncation boundary: differences of at most one level on at
    most `max_off_by_one_frac` of the pixels.  -> worst errors (for the records)."""
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))
    worst = {"Z": 0.0, "MASK": 0.0, "pixels_off_by_one_frac": 0.0, "ERROR": 0.0}
    for k, ev in enumerate(events):
        e = rel(ev["Z"], fx["%02d_Z" % k])
        assert e < tol, (k, ev["kind"], "Z", e)
        worst["Z"] = max(worst["Z"], e)
        if ev["MASK"] is not None:
            e = rel(ev["MASK"], fx["%02d_MASK" % k])
            assert e < tol, (k, "MASK", e)
            worst["MASK"] = max(worst["MASK"], e)
        for name in ("shown", "RECON"):
            key = "%02d_%s" % (k, name)
            if key in fx.files and ev.get(name) is not None:
                d = np.abs(np.asarray(ev[name], np.int32) - np.asarray(fx[key], np.int32))
                assert d.max() <= 1, (k, ev["kind"], name, int(d.max()))
                frac = float((d != 0).mean())
                assert frac <= max_off_by_one_frac, (k, ev["kind"], name, frac)
                worst["pixels_off_by_one_frac"] = max(worst["pixels_off_by_one_frac"], frac)
        if "ERROR" in ev:
            # one RECON level (where it differs) moves ERROR by 2/255: compare where RECON agrees
            same = np.asarray(ev["RECON"]) == fx["%02d_RECON" % k]
            e = float(np.abs(ev["ERROR"][same] - fx["%02d_ERROR" % k][same]).max())
            assert e < 1e-6, (k, "ERROR", e)
            worst["ERROR"] = max(worst["ERROR"], e)
    return worst

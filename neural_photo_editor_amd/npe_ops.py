"""The steps that sit right after the hot path in every NPE edit (NPE.py:192-235, 276-314): the latent update and
the photo blend.  In the reference they are 64x64x3 numpy/scipy expressions inside the Tk callbacks; here

  * ``brush_step`` / ``lighten_step`` are the latent updates of NPE.paint / NPE.scroll (host arithmetic on 100 floats
    around the HIP gradient call); ``paint_event`` is the whole NPE.paint body as one device submission
    (``ian_brush_step``: gradient, latent update, decoder, optional blend);
  * ``photo_blend`` is NPE.paint's photo-mode blend (NPE.py:218-231).  With a ``neural_photo_editor_amd.IAN`` model it
    runs as ONE 64x64 HIP kernel chained after the decoder (``ian_photo_blend``, include/ian.h): the decoded image never
    leaves the device, the edit needs one 12 KB uint8 device->host copy;
  * ``photo_blend_host`` is the reference's expression itself, dtype for dtype (float32 DELTA, float64 MASK / D, the
    bare ``np.uint8`` cast) -- the oracle the kernel is tested against bit-for-bit, and the fallback for models that
    are not the HIP class (a test double).
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter

BLEND_SIGMA = 0.7       # NPE.py:224
BLEND_RADIUS = 3        # scipy: int(truncate * sigma + 0.5) with truncate = 4.0


def to_tanh(x):
    """NPE.py:37-38 (dtype follows numpy promotion, as in the reference: uint8 -> float64, float32 -> float32)."""
    return 2.0 * (x / 255.0) - 1.0


def from_tanh(x):
    """NPE.py:40-41"""
    return 255.0 * (x + 1) / 2.0


def gaussian_half_kernel(sigma=BLEND_SIGMA, radius=BLEND_RADIUS):
    """Weights w[0..radius] (centre outwards) of scipy.ndimage's 1-D Gaussian (_filters._gaussian_kernel1d, order 0)."""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], np.float64)


def separable_reflect_filter(a, half):
    """scipy.ndimage.gaussian_filter restated: axis 0 then axis 1, 'reflect' boundary (d c b a | a b c d | d c b a),
    float64, and NI_Correlate1D's summation order for a symmetric kernel:
        t = x[l]*w0;  for j = R..1:  t += (x[l-j] + x[l+j]) * w[j]
    -- the order the HIP kernel reproduces (kernels_npe.hip), so that its mask equals scipy's bit for bit."""
    a = np.asarray(a, np.float64)
    R = len(half) - 1
    for axis in (0, 1):
        x = np.moveaxis(a, axis, 0)
        n = x.shape[0]
        idx = np.arange(-R, n + R)
        idx = np.where(idx < 0, -idx - 1, idx)
        idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
        xe = x[idx]
        t = xe[R:R + n] * half[0]
        for j in range(R, 0, -1):
            t = t + (xe[R - j:R - j + n] + xe[R + j:R + j + n]) * half[j]
        a = np.moveaxis(t, 0, axis)
    return a


def photo_blend_host(xhat, recon_uint8, error):
    """NPE.py:218-231 verbatim.  xhat = model.sample_at(Z)[0] (float32 (3,64,64)), recon_uint8 = RECON, error = ERROR
    (float32).  Returns (IM uint8, MASK float64)."""
    RECON = np.asarray(recon_uint8)
    DELTA = np.asarray(xhat, np.float32) - to_tanh(np.float32(RECON))
    MASK = gaussian_filter(np.min([np.mean(np.abs(DELTA), axis=0), np.ones((64, 64))], axis=0), BLEND_SIGMA)
    D = MASK * DELTA + (1 - MASK) * np.asarray(error)
    with np.errstate(invalid="ignore"):
        IM = np.uint8(from_tanh(to_tanh(RECON) + D))     # bare cast, as NPE.py:231: no clipping (out-of-range wraps)
    return IM, MASK


def brush_step(model, Z, box, rgb_uint8, weight=0.05):
    """NPE.paint's latent update (NPE.py:199-209), in the reference's order on a float32 Z:
    grad = dL/dZ * (1 + (x2 - x1));  Z -= weight * grad.
    Z: (10,10) or (1,100) float32; box = (x1, y1, x2, y2) in 64-pixel space; rgb_uint8: (3,64,64) brush colour image."""
    x1, y1, x2, y2 = [int(v) for v in box]
    shape = np.shape(Z)
    z = np.float32(np.reshape(Z, (1, -1)))
    g = np.asarray(model.imgradRGB(x1, y1, x2, y2, np.float32(to_tanh(np.float32(rgb_uint8)))[None], z)[0])
    grad = g * np.float32(1 + (x2 - x1))
    return (z[0] - np.float32(weight) * grad).reshape(shape).astype(np.float32)


def lighten_step(model, Z, box, weight=0.1, sign=1.0):
    """NPE.scroll (NPE.py:305-314): grad = d mean(patch)/dZ * (1 + (x2 - x1));  Z += sign(event.delta) * weight * grad."""
    x1, y1, x2, y2 = [int(v) for v in box]
    shape = np.shape(Z)
    z = np.float32(np.reshape(Z, (1, -1)))
    grad = np.asarray(model.imgrad(x1, y1, x2, y2, z)[0]) * np.float32(1 + (x2 - x1))
    return (z[0] + np.float32(float(sign) * float(weight)) * grad).reshape(shape).astype(np.float32)


def paint_event(model, Z, box, rgb_uint8, recon_uint8=None, error=None, weight=0.05):
    """The whole body of NPE.paint (NPE.py:199-231) for a float32 Z: brush step, then the sample (sample mode) or the blended
    photo (photo mode, when RECON / ERROR are given).  With the HIP model this is ONE device submission
    (``IAN.brush_step`` -> ian_brush_step); otherwise the composition of the calls above.
    -> (Z_new in Z's shape, image): image = float32 (3,64,64) sample in sample mode, uint8 (3,64,64) IM in photo mode."""
    x1, y1, x2, y2 = [int(v) for v in box]
    shape = np.shape(Z)
    z = np.float32(np.reshape(Z, (1, -1)))
    photo = (recon_uint8, error) if recon_uint8 is not None else None
    if hasattr(model, "brush_step"):
        rgb = np.float32(to_tanh(np.float32(rgb_uint8)))[None]
        if photo is None:
            z_new, x = model.brush_step(x1, y1, x2, y2, z, RGB=rgb, weight=weight)
            return z_new.reshape(shape), x[0]
        z_new, _, im, _ = model.brush_step(x1, y1, x2, y2, z, RGB=rgb, weight=weight, image=False, photo=photo)
        return z_new.reshape(shape), im
    z_new = brush_step(model, z, box, rgb_uint8, weight)
    if photo is None:
        return z_new.reshape(shape), model.sample_at(z_new)[0]
    return z_new.reshape(shape), photo_blend(model, z_new, recon_uint8, error)[0]


def photo_blend(model, Z, recon_uint8, error):
    """NPE.py:218-231: DELTA = G(Z) - to_tanh(RECON); MASK = gaussian_filter(min(mean|DELTA|, 1), 0.7);
    IM = uint8(from_tanh(to_tanh(RECON) + MASK*DELTA + (1-MASK)*ERROR)).  recon_uint8, error: (3,64,64).
    -> (IM uint8 (3,64,64), MASK float64 (64,64)).  Runs on the device when the model offers it."""
    z = np.float32(np.reshape(Z, (1, -1)))
    if hasattr(model, "photo_blend"):
        return model.photo_blend(z, recon_uint8, error)
    return photo_blend_host(model.sample_at(z)[0], recon_uint8, error)

"""Host-side steps that sit right after the hot path in every NPE edit (NPE.py:192-235, 276-300): the latent
update and the photo blend.  They are 64x64x3 numpy/scipy expressions in the reference and stay on the host here
(the model calls around them -- imgradRGB, sample_at -- are the HIP path); collected so that an editor built on
``neural_photo_editor_amd.IAN`` does not have to copy them out of the Tk callbacks."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter


def to_tanh(x):
    """NPE.py:37-38"""
    return 2.0 * (np.asarray(x, np.float32) / 255.0) - 1.0


def from_tanh(x):
    """NPE.py:34-35"""
    return 255.0 * (np.asarray(x, np.float32) + 1) / 2.0


def brush_step(model, Z, box, rgb_uint8, weight=0.05):
    """NPE.paint's latent update (NPE.py:199-209): Z -= weight * dL/dZ * (1 + (x2 - x1)).
    Z: (10,10) or (1,100) float32; box = (x1, y1, x2, y2) in 64-pixel space; rgb_uint8: (3,64,64) brush colour image."""
    x1, y1, x2, y2 = [int(v) for v in box]
    shape = np.shape(Z)
    z = np.float32(np.reshape(Z, (1, -1)))
    g = np.asarray(model.imgradRGB(x1, y1, x2, y2, np.float32(to_tanh(rgb_uint8))[None], z)[0])
    return (z[0] - weight * g * (1 + (x2 - x1))).reshape(shape).astype(np.float32)


def lighten_step(model, Z, box, weight=0.1, sign=1.0):
    """NPE.scroll (NPE.py:305-314): Z += sign * weight * d mean(patch) / dZ."""
    x1, y1, x2, y2 = [int(v) for v in box]
    shape = np.shape(Z)
    z = np.float32(np.reshape(Z, (1, -1)))
    g = np.asarray(model.imgrad(x1, y1, x2, y2, z)[0])
    return (z[0] + sign * weight * g).reshape(shape).astype(np.float32)


def photo_blend(model, Z, recon_uint8, error):
    """NPE.py:218-231: DELTA = G(Z) - to_tanh(RECON); MASK = gaussian_filter(min(mean|DELTA|, 1), 0.7);
    IM = uint8(from_tanh(to_tanh(RECON) + MASK*DELTA + (1-MASK)*ERROR)).  recon_uint8, error: (3,64,64)."""
    z = np.float32(np.reshape(Z, (1, -1)))
    delta = model.sample_at(z)[0] - to_tanh(np.float32(recon_uint8))
    mask = gaussian_filter(np.min([np.mean(np.abs(delta), axis=0), np.ones((64, 64))], axis=0), 0.7)
    d = mask * delta + (1 - mask) * np.asarray(error, np.float32)
    return np.uint8(np.clip(from_tanh(to_tanh(recon_uint8) + d), 0, 255)), mask

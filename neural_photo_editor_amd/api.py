"""Plat-style model facade: the drop-in for the reference's API.py.

``IAN(config_path, dnn)`` keeps API.py's surface (API.py:11-110) so that NPE.py's call sequence
(NPE.py:18,110,205,218,257,261,296,311,323,333,335) runs unchanged:

    model = IAN(config_path='IAN_simple.py', dnn=True)
    z  = model.encode_images(x)            # f32[n,3,64,64] in [-1,1] -> f32[n,100]
    x_ = model.sample_at(z)                # f32[n,100] -> f32[n,3,64,64]
    g  = model.imgradRGB(c1,r1,c2,r2,RGB,z); g2 = model.imgrad(c1,r1,c2,r2,z)
    model.get_zdim(); model.cfg; model.model

Everything numerical happens in libian.so (hand-written HIP, include/ian.h); this class only reads
the config, loads the checkpoint and moves numpy buffers across the C ABI.  Extra methods give the
four function equivalents sample_IAN.py compiles for itself (sample_IAN.py:86-94).
"""
from __future__ import annotations

import logging
import os
import warnings

import numpy as np

from . import checkpoints, config_loader, lowering, made
from .lib import Handle


class IAN:
    def __init__(self, config_path, dnn=True, params=None, deconv_flip=True):
        """config_path, dnn: as API.py:12.  ``params`` (optional extension): dict of Theano-named arrays
        used instead of the '<config>.npz' checkpoint (tests and the benchmark use synthetic weights
        because the reference's weight files are absent)."""
        config_module = config_loader.load_config(config_path)
        self.cfg = config_module.cfg                                   # API.py:19
        self.weights_fname = str(config_path)[:-3] + ".npz"            # API.py:20
        self.model = config_loader.build_model(config_module, dnn=dnn)  # API.py:21
        self.lowered = lowering.lower_model(self.model)
        self._h = Handle(self.lowered, deconv_flip=deconv_flip)
        self.metadata = {}

        # Load weights (API.py:23-30)
        specs = self.lowered.params
        if params is None:
            params = {}
            if os.path.exists(self.weights_fname):
                try:
                    params, self.metadata = checkpoints.load_weights(self.weights_fname, lowering.all_param_specs(self.model))
                except Exception as exc:  # e.g. a git-LFS pointer instead of the archive
                    warnings.warn("could not read %s (%s); parameters keep their initial values" % (self.weights_fname, exc))
            else:
                warnings.warn("weights file %s not found; parameters keep their initial values" % self.weights_fname)
        rs = np.random.RandomState(0)
        for p in specs:
            arr = params.get(p.name)
            if arr is None:
                logging.warning("unable to load parameter %s", p.name)
                arr = p.init.sample(p.shape, rs)
            elif tuple(np.shape(arr)) != p.shape:
                raise ValueError("parameter %s has shape %s, expected %s" % (p.name, np.shape(arr), p.shape))
            self._h.load_param(p.name, arr)

        # Shuffle weights if using IAF with MADE (API.py:32-36): reset("Once") on both MADEs
        if "l_IAF_mu" in self.model:
            self.model["l_IAF_mu"].reset("Once")
            self.model["l_IAF_ls"].reset("Once")
        if self.lowered.has_made:
            self.made_masks = made.masks_once(self.lowered.num_latents)
            self._h.set_made_masks(*self.made_masks)
        self._h.finalize()
        self._zdim = self.lowered.num_latents

    # ---- helpers -------------------------------------------------------------------------------------
    @staticmethod
    def _f32(a, shape_tail, what):
        if not (type(a) is np.ndarray and a.dtype == np.float32 and a.flags.c_contiguous):    # the interactive loop passes ready arrays
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        if a.ndim != len(shape_tail) + 1 or tuple(a.shape[1:]) != tuple(shape_tail):
            raise ValueError("%s must have shape (n,%s), got %s" % (what, ",".join(map(str, shape_tail)), a.shape))
        if a.shape[0] < 1:
            raise ValueError("%s is empty" % what)
        return a

    def _grad_out(self, z):
        """API.py:59,64 differentiate a loss on X_hat[0] with respect to the WHOLE Z: the result has Z's shape (n, zdim) and, the
        decoder being deterministic (per-sample), rows 1.. are exactly zero.  NPE.py only ever passes n = 1."""
        return np.empty((1, self._zdim), np.float32) if z.shape[0] == 1 else np.zeros((z.shape[0], self._zdim), np.float32)

    def _run(self, fn, src, out_tail):
        out = np.empty((src.shape[0],) + tuple(out_tail), np.float32)
        self._h.call(fn, src, src.shape[0], out)
        return out

    # ---- API.py surface ------------------------------------------------------------------------------
    def imgrad(self, c1, r1, c2, r2, z):
        """API.py:66-70: change in latents which would lighten the local image patch."""
        z = self._f32(z, (self._zdim,), "z")
        dz = self._grad_out(z)
        self._h.grad_light(int(c1), int(r1), int(c2), int(r2), z[:1], dz[:1])
        return dz

    def imgradRGB(self, c1, r1, c2, r2, RGB, z):
        """API.py:72-76: change in latents which would move the patch towards RGB."""
        z = self._f32(z, (self._zdim,), "z")
        rgb = self._f32(RGB, (3, 64, 64), "RGB")
        dz = self._grad_out(z)
        self._h.grad_rgb(int(c1), int(r1), int(c2), int(r2), rgb[:1], z[:1], dz[:1])
        return dz

    def encode_images(self, images):
        """API.py:78-90: n x 3 x 64 x 64 in [-1,1] -> n x zdim."""
        return self._run("ian_encode", self._f32(images, (3, 64, 64), "images"), (self._zdim,))

    def get_zdim(self):
        """API.py:92-96."""
        return self.cfg["num_latents"]

    def sample_at(self, z):
        """API.py:98-110: n x zdim -> n x 3 x 64 x 64."""
        return self._run("ian_decode", self._f32(z, (self._zdim,), "z"), (3, 64, 64))

    # ---- what NPE.py does with the decoder output on every edit, on the device ----------
    def sample_at_uint8(self, z):
        """np.uint8(from_tanh(sample_at(z))) as in NPE.py:110,261 (update_photo, RECON): n x zdim -> uint8 n x 3 x 64 x 64;
        the float image never leaves the device."""
        z = self._f32(z, (self._zdim,), "z")
        out = np.empty((z.shape[0], 3, 64, 64), np.uint8)
        self._h.call("ian_decode_u8", z, z.shape[0], out)
        return out

    def photo_blend(self, z, recon_uint8, error, sigma=0.7):
        """NPE.paint's photo-mode blend (NPE.py:218-231) chained after the decoder on the device:
        -> (IM uint8 (3,64,64), MASK float64 (64,64)), bit-exact with the numpy/scipy expression (npe_ops.photo_blend_host)."""
        from . import npe_ops
        z = self._f32(z, (self._zdim,), "z")
        recon = np.ascontiguousarray(recon_uint8, dtype=np.uint8)
        err = np.ascontiguousarray(error, dtype=np.float32)
        if recon.shape != (3, 64, 64) or err.shape != (3, 64, 64):
            raise ValueError("RECON and ERROR must have shape (3,64,64)")
        half = npe_ops.gaussian_half_kernel(sigma, int(4.0 * float(sigma) + 0.5))
        im, mask = np.empty((3, 64, 64), np.uint8), np.empty((64, 64), np.float64)
        self._h.photo_blend(z[:1], recon, err, half, im, mask)
        return im, mask

    def brush_step(self, c1, r1, c2, r2, z, RGB=None, weight=0.05, sign=-1.0, image=True, photo=None, sigma=0.7, want_mask=False):
        """One whole NPE.paint / NPE.scroll event on the device (ian_brush_step): the latent gradient (imgradRGB when RGB is
        given, imgrad otherwise), Z + sign*weight*(dZ*(1+(c2-c1))) in float32 exactly as NPE.py:205-209 / 313-314 compute it,
        and sample_at of the new latent -- one submission instead of two calls with a host update between them.
        -> (z_new (1,zdim), image (1,3,64,64) or None)  or, with photo=(RECON uint8, ERROR float32),
           (z_new, image or None, IM uint8 (3,64,64), MASK float64 (64,64) if want_mask else None)  (photo mode,
           NPE.py:218-231; NPE.paint itself only displays IM, so the 32 KB mask stays on the device unless asked for)."""
        z = self._f32(z, (self._zdim,), "z")
        rgb = self._f32(RGB, (3, 64, 64), "RGB")[:1] if RGB is not None else None
        z_new = np.empty((1, self._zdim), np.float32)
        x = np.empty((1, 3, 64, 64), np.float32) if image else None
        pa = None
        if photo is not None:
            recon = np.ascontiguousarray(photo[0], dtype=np.uint8)
            err = np.ascontiguousarray(photo[1], dtype=np.float32)
            if recon.shape != (3, 64, 64) or err.shape != (3, 64, 64):
                raise ValueError("RECON and ERROR must have shape (3,64,64)")
            from . import npe_ops
            half = npe_ops.gaussian_half_kernel(sigma, int(4.0 * float(sigma) + 0.5))
            im, mask = np.empty((3, 64, 64), np.uint8), (np.empty((64, 64), np.float64) if want_mask else None)
            pa = (recon, err, half, im, mask)
        coef = float(sign) * float(weight)      # rounded to float32 at the boundary, as numpy rounds the scalar
        self._h.brush_step(int(c1), int(r1), int(c2), int(r2), rgb, z[:1], coef, float(1 + (int(c2) - int(c1))), z_new, None, x, pa)
        if photo is not None:
            return z_new, x, pa[3], pa[4]
        return z_new, x

    # ---- sample_IAN.py function equivalents ----------------------------------------------
    def sampleZ(self, z):
        """sample_IAN.py:88: l_Z -> l_out (same as sample_at)."""
        return self.sample_at(z)

    def Zfn(self, images):
        """sample_IAN.py:90-91: image -> l_Z_IAF (deterministic)."""
        return self._run("ian_encode_pre_iaf", self._f32(images, (3, 64, 64), "images"), (self._zdim,))

    def Z_IAF_fn(self, z):
        """sample_IAN.py:93-94: l_Z_IAF -> l_Z."""
        return self._run("ian_iaf", self._f32(z, (self._zdim,), "z"), (self._zdim,))

    def sample(self, z):
        """sample_IAN.py:86: l_Z_IAF -> l_out."""
        return self.sample_at(self.Z_IAF_fn(z))

    def reconstruct(self, images):
        """encode_images followed by sample_at with the latent kept on the device (bench config 2)."""
        return self._run("ian_reconstruct", self._f32(images, (3, 64, 64), "images"), (3, 64, 64))

    # ---- introspection ---------------------------------------------------------------------------------
    def activation(self, name, n):
        """Activation (NCHW) of the named layer output from the last call (tests / debugging)."""
        return self._h.read_slot(self.lowered.slot_by_name(name), n)

    @property
    def handle(self):
        return self._h

    def close(self):
        self._h.close()

"""Thin ctypes caller of ``ian_train_step`` (include/ian_train.h, csrc/ian_trainer.cpp): the train_IAN.py update functions
(train_IAN.py:309-329) of the full IAN on ONE GPU, wired in C++ inside libian.so.  Python here only loads the parameters,
hands over buffers and names the metrics; ``trainer.Trainer`` sequences the same launches itself when the step is data
parallel (torch.distributed / RCCL)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import config_loader, made
from .lib import load_train_library

METRICS = ("discrim_d_loss", "gen_recon_loss", "gen_sample_loss", "discrim_g_loss", "discrim_acc", "kl_div", "pixel_loss",
           "pixel_acc", "feature_loss")
GEN_KEYS = ("gen_recon_loss", "gen_sample_loss", "pixel_loss", "feature_loss", "pixel_acc")          # train_IAN.py:291-296
DISCRIM_KEYS = ("discrim_g_loss", "discrim_d_loss", "discrim_acc", "pixel_loss", "pixel_acc")        # train_IAN.py:298-304


class TrainConfig(C.Structure):
    _fields_ = [("batch", C.c_int32), ("num_latents", C.c_int32), ("deconv_flip", C.c_int32), ("reserved", C.c_int32),
                ("learning_rate", C.c_double), ("beta1", C.c_double), ("reg", C.c_float), ("ortho", C.c_float), ("recon_weight", C.c_float),
                ("feature_weight", C.c_float), ("dg_weight", C.c_float), ("dd_weight", C.c_float), ("agr_weight", C.c_float),
                ("ags_weight", C.c_float)]


class CTrainerError(RuntimeError):
    pass


class CTrainer:
    def __init__(self, config_path, params, batch, deconv_flip=True):
        self.lib = load_train_library()
        c = dict(config_loader.load_config(config_path).cfg)
        self.cfg = c
        lr = c["learning_rate"][0] if isinstance(c["learning_rate"], dict) else c["learning_rate"]
        tc = TrainConfig(int(batch), int(c["num_latents"]), int(bool(deconv_flip)), 0, float(lr), float(c["beta1"]), float(c["reg"]),
                         float(c.get("ortho", -1.0)), float(c["recon_weight"]), float(c["feature_weight"]), float(c["dg_weight"]),
                         float(c["dd_weight"]), float(c["agr_weight"]), float(c["ags_weight"]))
        self.n, self.lr = int(batch), float(lr)
        self._h = C.c_void_p()
        rc = self.lib.ian_trainer_create(C.byref(tc), C.byref(self._h))
        if rc:
            raise CTrainerError("ian_trainer_create failed (%d)%s" % (rc, ": no HIP device, libian has no CPU fallback" if rc == -10 else ""))
        self.shapes = {}
        for name, v in params.items():
            a = np.ascontiguousarray(v, np.float32)
            rc = self.lib.ian_trainer_load_param(self._h, name.encode(), C.c_void_p(a.ctypes.data), a.size)
            if rc == -2:
                continue                       # an entry the training graph does not own
            self._check(rc)
            self.shapes[name] = a.shape
        m = [np.ascontiguousarray(x, np.float32) for x in made.masks_once(int(c["num_latents"]))]   # train_IAN.py:404-405
        self._check(self.lib.ian_trainer_set_made_masks(self._h, *[C.c_void_p(x.ctypes.data) for x in m], m[0].shape[0]))
        self._check(self.lib.ian_trainer_finalize(self._h))

    def _check(self, rc):
        if rc:
            raise CTrainerError("libian trainer error %d: %s" % (rc, (self.lib.ian_trainer_last_error(self._h) or b"?").decode()))

    @staticmethod
    def _ptr(a):
        if hasattr(a, "data_ptr"):             # a torch tensor on the device (or the host)
            return C.c_void_p(a.data_ptr())
        return C.c_void_p(a.ctypes.data)

    def step(self, which, X, Z, eps, return_metrics=True, stream=0):
        """which: 'gen' | 'discrim'.  X (n,3,64,64), Z (n,100), eps (n,100): float32 numpy arrays or device tensors."""
        keep = [a if hasattr(a, "data_ptr") else np.ascontiguousarray(a, np.float32) for a in (X, Z, eps)]
        out = (C.c_float * 9)() if return_metrics else None
        n = int(keep[0].shape[0])
        if int(keep[1].shape[0]) != n or int(keep[2].shape[0]) != n:
            raise CTrainerError("X, Z and eps must hold the same number of rows")
        self._check(self.lib.ian_train_step(self._h, 0 if which == "gen" else 1, self._ptr(keep[0]), self._ptr(keep[1]),
                                            self._ptr(keep[2]), n, out, C.c_void_p(stream)))
        return dict(zip(METRICS, [float(v) for v in out])) if return_metrics else None

    def update_gen(self, X, Z, eps):
        m = self.step("gen", X, Z, eps)
        return [m[k] for k in GEN_KEYS]

    def update_discrim(self, X, Z, eps):
        m = self.step("discrim", X, Z, eps)
        return [m[k] for k in DISCRIM_KEYS]

    def autotune(self, stream=0):
        self._check(self.lib.ian_trainer_autotune(self._h, C.c_void_p(stream)))

    def set_option(self, key, value):
        self._check(self.lib.ian_trainer_set_option(self._h, key.encode(), float(value)))

    def read(self, name, grad=False):
        out = np.empty(self.shapes[name], np.float32)
        self._check(self.lib.ian_trainer_read_param(self._h, name.encode(), int(grad), C.c_void_p(out.ctypes.data), out.size))
        return out

    def adam_steps(self):
        return tuple(self.lib.ian_trainer_adam_steps(self._h, g) for g in (0, 1, 2))

    def state_dict(self):
        return {n: self.read(n) for n in self.shapes if not n.startswith("l_IAF_")}

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ian_trainer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

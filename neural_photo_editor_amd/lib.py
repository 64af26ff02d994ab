"""ctypes binding of libian.so (include/ian.h).  No compute happens in Python.

The library is hand-written HIP for gfx950; there is deliberately NO fallback: if it cannot be built
or loaded, or no HIP device is present when a model is finalized, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

IAN_MAX_SCALES = 4


class OpDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("segment", C.c_int32), ("src", C.c_int32), ("src2", C.c_int32), ("src3", C.c_int32),
        ("dst", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("act", C.c_int32), ("has_bias", C.c_int32),
        ("flat_c", C.c_int32), ("flat_h", C.c_int32), ("flat_w", C.c_int32),
        ("unflat_c", C.c_int32), ("unflat_h", C.c_int32), ("unflat_w", C.c_int32),
        ("n_scales", C.c_int32), ("scales", C.c_int32 * IAN_MAX_SCALES),
        ("name", C.c_char_p), ("bn_name", C.c_char_p),
    ]


class SlotDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32), ("ops", C.POINTER(OpDesc)), ("n_slots", C.c_int32), ("slots", C.POINTER(SlotDesc)),
        ("x_slot", C.c_int32), ("zpre_slot", C.c_int32), ("z_slot", C.c_int32), ("out_slot", C.c_int32),
        ("num_latents", C.c_int32), ("deconv_flip", C.c_int32),
    ]


EXPORTS = (
    "ian_create", "ian_load_param", "ian_set_made_masks", "ian_finalize", "ian_encode", "ian_decode",
    "ian_encode_pre_iaf", "ian_iaf", "ian_reconstruct", "ian_grad_rgb", "ian_grad_light", "ian_decode_u8", "ian_photo_blend",
    "ian_read_slot", "ian_read_slot_grad",
    "ian_brush_step", "ian_profile_enable", "ian_profile_read", "ian_autotune", "ian_set_option", "ian_box_probe", "ian_last_error", "ian_version", "ian_destroy",
)

_lib = None


def library_path():
    return _build.LIB


def is_ablation_build():
    """True when the loaded library is libian_ablation.so (IAN_ABLATION_BUILD=1 + IAN_LIB=...): the negative-result variants
    (tapgemm schedules 0 / 3, in-launch split-K combine, kernels_b1.hip, 4-wave tapwgrad tile) exist only there."""
    lib = load_library()
    lib.ian_version.restype = C.c_char_p
    return b"ablation" in (lib.ian_version() or b"")


def load_library():
    """Load (building first if the in-tree .so is absent or stale and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    override = os.environ.get("IAN_LIB")     # tests/test_sanitize.py: the ASan/UBSan build of the same sources (build.py, IAN_SANITIZE=1)
    if override:
        if not os.path.exists(override):
            raise RuntimeError("IAN_LIB=%s does not exist" % override)
        path = override
    elif _build.have_hipcc():
        path = _build.build()      # a compile / link error propagates: never run stale kernels behind a failed build
    elif not os.path.exists(path):
        raise RuntimeError("libian.so is missing and hipcc is not available to build it")
    elif not _build.is_fresh():
        raise RuntimeError("libian.so does not match the sources under csrc/ (digest stamp differs) and hipcc is not "
                           "available to rebuild it")
    # Load order matters when PyTorch-ROCm lives in the same process (device buffers, streams, torch.distributed): it ships
    # its own HIP runtime next to the system one libian links.  With torch's loaded first both work side by side (every
    # test and bench.py run that way); with libian's loaded first, libian's runtime finds no device once torch has
    # initialised (measured on the MI355X box: build() followed by smoke() in one process).  So: torch first, if present.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32, fp = C.c_void_p, C.c_int32, C.c_void_p
    lib.ian_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    lib.ian_load_param.argtypes = [vp, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), i32]
    lib.ian_set_made_masks.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, i32]
    lib.ian_finalize.argtypes = [vp]
    for fn in ("ian_encode", "ian_decode", "ian_encode_pre_iaf", "ian_iaf", "ian_reconstruct"):
        getattr(lib, fn).argtypes = [vp, fp, i32, fp, vp]
    lib.ian_grad_rgb.argtypes = [vp, i32, i32, i32, i32, fp, fp, fp, vp]
    lib.ian_grad_light.argtypes = [vp, i32, i32, i32, i32, fp, fp, vp]
    lib.ian_decode_u8.argtypes = [vp, fp, i32, fp, vp]
    lib.ian_photo_blend.argtypes = [vp, fp, fp, fp, fp, i32, fp, fp, vp]
    lib.ian_brush_step.argtypes = [vp, i32, i32, i32, i32, fp, fp, C.c_float, C.c_float, fp, fp, fp, C.POINTER(PhotoArgs), vp]
    lib.ian_read_slot.argtypes = [vp, i32, i32, fp, vp]
    lib.ian_read_slot_grad.argtypes = [vp, i32, i32, fp, vp]
    lib.ian_profile_enable.argtypes = [vp, i32]
    lib.ian_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
    lib.ian_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.ian_autotune.argtypes = [vp, i32, i32, vp]
    lib.ian_box_probe.argtypes = [i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    lib.ian_last_error.argtypes = [vp]
    lib.ian_last_error.restype = C.c_char_p
    lib.ian_version.restype = C.c_char_p
    lib.ian_destroy.argtypes = [vp]
    lib.ian_destroy.restype = None
    for fn in EXPORTS:
        if fn not in ("ian_last_error", "ian_version", "ian_destroy"):
            getattr(lib, fn).restype = i32
    _lib = lib
    return lib


class PhotoArgs(C.Structure):
    """ian_photo_args (include/ian.h)."""
    _fields_ = [("recon", C.c_void_p), ("error", C.c_void_p), ("gauss_half", C.c_void_p), ("radius", C.c_int32),
                ("im", C.c_void_p), ("mask", C.c_void_p)]


class IanError(RuntimeError):
    pass


def box_probe(iters=900, launches=250, stream=None):
    """ian_box_probe: sustained fp32-MFMA rate of this box (register-only loop, launches of ~200 us at iters = 900).
    -> {"tflops": ..., "us_per_launch": ...}"""
    lib = load_library()
    tf, us = C.c_double(), C.c_double()
    rc = lib.ian_box_probe(int(iters), int(launches), C.byref(tf), C.byref(us), C.c_void_p(stream or 0))
    if rc != 0:
        raise IanError("ian_box_probe failed (%d)" % rc)
    return {"tflops": tf.value, "us_per_launch": us.value}


def _ptr(buf):
    """Raw address of a numpy array (host) or of anything exposing data_ptr() (torch device tensor)."""
    if isinstance(buf, np.ndarray):
        return C.c_void_p(buf.__array_interface__["data"][0])    # (ndarray.ctypes builds a helper object per access: ~1 us each)
    if hasattr(buf, "data_ptr"):
        return C.c_void_p(buf.data_ptr())
    if isinstance(buf, int):
        return C.c_void_p(buf)
    raise TypeError("expected a numpy array, a tensor with data_ptr() or an address")


class Handle:
    """Thin RAII wrapper over ian_handle*."""

    def __init__(self, lowered, deconv_flip=True):
        self.lib = load_library()
        self.lowered = lowered
        n = len(lowered.ops)
        self._keep = []
        ops = (OpDesc * n)()
        for i, op in enumerate(lowered.ops):
            d = ops[i]
            d.kind, d.segment, d.src, d.src2, d.src3, d.dst = op.kind, op.segment, op.src, op.src2, op.src3, op.dst
            d.cin, d.cout, d.in_h, d.in_w, d.act, d.has_bias = op.cin, op.cout, op.in_h, op.in_w, op.act, op.has_bias
            d.flat_c, d.flat_h, d.flat_w = op.flat
            d.unflat_c, d.unflat_h, d.unflat_w = op.unflat
            d.n_scales = len(op.scales)
            for k, s in enumerate(op.scales):
                d.scales[k] = s
            nm = op.name.encode()
            self._keep.append(nm)
            d.name = nm
            if op.bn_name:
                bn = op.bn_name.encode()
                self._keep.append(bn)
                d.bn_name = bn
            else:
                d.bn_name = None
        slots = (SlotDesc * len(lowered.slots))()
        for i, (h, w, c) in enumerate(lowered.slots):
            slots[i].h, slots[i].w, slots[i].c = h, w, c
        desc = ModelDesc(n, ops, len(lowered.slots), slots, lowered.x_slot, lowered.zpre_slot, lowered.z_slot,
                         lowered.out_slot, lowered.num_latents, int(bool(deconv_flip)))
        self._h = C.c_void_p()
        rc = self.lib.ian_create(C.byref(desc), C.byref(self._h))
        if rc != 0:
            raise IanError("ian_create failed (%d)" % rc)

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.ian_last_error(self._h)
            raise IanError("libian error %d: %s" % (rc, msg.decode() if msg else "?"))

    def load_param(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self._check(self.lib.ian_load_param(self._h, name.encode(), C.c_void_p(a.ctypes.data), shape, a.ndim))

    def set_made_masks(self, m0, m1, md):
        arrs = [np.ascontiguousarray(m, dtype=np.float32) for m in (m0, m1, md)]
        self._check(self.lib.ian_set_made_masks(self._h, *[C.c_void_p(a.ctypes.data) for a in arrs], arrs[0].shape[0]))

    def finalize(self):
        self._check(self.lib.ian_finalize(self._h))

    def call(self, fn, src, n, dst, stream=None):
        self._check(getattr(self.lib, fn)(self._h, _ptr(src), n, _ptr(dst), C.c_void_p(stream or 0)))

    def grad_rgb(self, c1, r1, c2, r2, rgb, z, dz, stream=None):
        self._check(self.lib.ian_grad_rgb(self._h, c1, r1, c2, r2, _ptr(rgb), _ptr(z), _ptr(dz), C.c_void_p(stream or 0)))

    def grad_light(self, c1, r1, c2, r2, z, dz, stream=None):
        self._check(self.lib.ian_grad_light(self._h, c1, r1, c2, r2, _ptr(z), _ptr(dz), C.c_void_p(stream or 0)))

    def photo_blend(self, z, recon_u8, error, gauss_half, im, mask=None, stream=None):
        self._check(self.lib.ian_photo_blend(self._h, _ptr(z), _ptr(recon_u8), _ptr(error), _ptr(gauss_half), len(gauss_half) - 1,
                                             _ptr(im), _ptr(mask) if mask is not None else C.c_void_p(0), C.c_void_p(stream or 0)))

    def brush_step(self, c1, r1, c2, r2, rgb, z, coef, gscale, z_new, dz=None, x=None, photo=None, stream=None):
        """ian_brush_step; photo = (recon_u8, error, gauss_half, im, mask or None)."""
        null = C.c_void_p(0)
        pa = None
        if photo is not None:
            recon, err, half, im, mask = photo
            pa = PhotoArgs(_ptr(recon).value, _ptr(err).value, _ptr(half).value, len(half) - 1, _ptr(im).value,
                           _ptr(mask).value if mask is not None else None)
        self._check(self.lib.ian_brush_step(self._h, c1, r1, c2, r2, _ptr(rgb) if rgb is not None else null, _ptr(z), coef, gscale,
                                            _ptr(z_new), _ptr(dz) if dz is not None else null, _ptr(x) if x is not None else null,
                                            C.byref(pa) if pa is not None else None, C.c_void_p(stream or 0)))

    def read_slot(self, slot, n):
        h, w, c = self.lowered.slots[slot]
        out = np.empty((n, c, h, w), np.float32)
        self._check(self.lib.ian_read_slot(self._h, slot, n, _ptr(out), C.c_void_p(0)))
        return out

    def read_slot_grad(self, slot, n=1):
        h, w, c = self.lowered.slots[slot]
        out = np.empty((n, c, h, w), np.float32)
        self._check(self.lib.ian_read_slot_grad(self._h, slot, n, _ptr(out), C.c_void_p(0)))
        return out

    def profile_enable(self, on=True):
        self._check(self.lib.ian_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, n, fl, tot = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        self._check(self.lib.ian_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(fl), C.byref(tot)))
        return {"tapgemm_ms": ms.value, "tapgemm_launches": n.value, "tapgemm_flops": fl.value, "total_ms": tot.value}

    def autotune(self, n, what=1, stream=None):
        self._check(self.lib.ian_autotune(self._h, int(n), int(what), C.c_void_p(stream or 0)))

    def set_option(self, key, value):
        self._check(self.lib.ian_set_option(self._h, key.encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ian_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- include/ian_train.h: layer objects + element-wise ops of the training step -------------------------------------
_TRAIN_HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ian_train.h")
_train_ready = False


def parse_header_prototypes(path):
    """[(return type, name, [argument C types])] of every function declared in a C header (plain C, one per ';')."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"#.*", " ", text)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ian_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                argt.append("ptr" if "*" in a else a.rsplit(" ", 1)[0].replace("const ", "").strip())
        out.append((ret, name, argt))
    return out


_CT = {"int32_t": C.c_int32, "int64_t": C.c_int64, "float": C.c_float, "double": C.c_double, "int": C.c_int, "ptr": C.c_void_p}


def load_train_library():
    """libian.so with argtypes/restype of every include/ian_train.h entry point set from the header itself."""
    global _train_ready
    lib = load_library()
    if not _train_ready:
        for ret, name, argt in parse_header_prototypes(_TRAIN_HEADER):
            fn = getattr(lib, name)
            fn.argtypes = [_CT[t] for t in argt]
            fn.restype = C.c_char_p if "char" in ret else (None if ret == "void" else (C.c_int64 if "int64" in ret else C.c_int32))
        _train_ready = True
    return lib


def train_exports():
    return [name for _, name, _ in parse_header_prototypes(_TRAIN_HEADER)]

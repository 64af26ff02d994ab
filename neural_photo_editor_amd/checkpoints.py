"""npz checkpoints in the reference's format (GANcheckpoints.py:11-57), readable and writable in both directions.

An archive maps Theano parameter names to arrays, plus an optional 'metadata' entry.  The
reference writes ``param_dict['metadata'] = pickle.dumps(metadata)`` under Python 2 (GANcheckpoints.py:21): a
protocol-0 (ASCII) pickle held in a ``str``, which ``np.savez_compressed`` stores as a 0-d ``|S<n>`` array, and
reads it back with ``pickle.loads(str(param_dict['metadata']))`` (:54).  ``save_weights`` writes exactly that
representation -- a 0-d bytes array holding a protocol-0 pickle of plain Python values, emitted the way cPickle
does (``S'key'`` strings, ``I``/``F`` numbers) -- so the reference's ``API.py`` / ``train_IAN.py --resume`` can
open a checkpoint written here, and ``load_weights`` reads the reference's own files, including the
``np.float32`` learning rate train_IAN.py:571 puts into the metadata (a ``numpy.core.multiarray.scalar`` reduce).

``load_weights`` keeps the reference's behaviour: match by name, warn and skip on a shape mismatch (:40-50),
warn on a missing entry (:53-54), return the metadata dict (:55-58).  Archives are opened with
``allow_pickle=False`` and the metadata goes through a restricted unpickler (numpy scalar / dtype
reconstruction only): loading a downloaded weight file cannot execute code.
"""
from __future__ import annotations

import io
import logging
import os
import pickle
import warnings

import numpy as np


# ---- metadata: Python-2 cPickle protocol 0, plain values only ------------------------------------------------------
def _plain(v):
    """numpy scalars / 0-d arrays -> Python numbers (a numpy>=2 scalar would pickle a reference to numpy._core,
    which Python-2 numpy cannot import)."""
    if isinstance(v, (np.generic, np.ndarray)):
        v = np.asarray(v)
        if v.ndim != 0:
            raise TypeError("checkpoint metadata values must be scalars or strings, got an array of shape %s" % (v.shape,))
        v = v.item()
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    raise TypeError("checkpoint metadata values must be scalars or strings, got %r" % type(v))


def _p0_atom(v):
    if v is None:
        return "N"
    if isinstance(v, bool):
        return "I01\n" if v else "I00\n"
    if isinstance(v, int):
        return ("I%d\n" % v) if -2 ** 31 <= v < 2 ** 31 else ("L%dL\n" % v)
    if isinstance(v, float):
        return "F%s\n" % repr(v)
    s = v.encode("ascii").decode("ascii")          # metadata strings are ASCII (keys: 'epoch', 'itr', ...)
    return "S%s\n" % repr(s)                       # repr of a printable-ASCII str is the same in Python 2 and 3


def dumps_py2(metadata):
    """Protocol-0 pickle text of a flat dict, byte-for-byte what Python 2's ``cPickle.dumps`` emits for it
    (memo ``p<i>`` entries included), as bytes."""
    out = ["(dp0\n"]
    memo = 1
    for k, v in metadata.items():
        if not isinstance(k, str):
            raise TypeError("checkpoint metadata keys must be strings")
        out.append(_p0_atom(k) + "p%d\n" % memo)
        memo += 1
        v = _plain(v)
        out.append(_p0_atom(v))
        if isinstance(v, str):
            out.append("p%d\n" % memo)
            memo += 1
        out.append("s")
    out.append(".")
    return "".join(out).encode("ascii")


class _MetadataUnpickler(pickle.Unpickler):
    """Only what a metadata dict of train_IAN.py:571 needs: numpy scalar / dtype reconstruction."""
    # _codecs.encode: Python 3 protocol<=2 pickles of numpy scalars carry their raw bytes as
    # ``_codecs.encode(u'...', 'latin1')`` (round-1 archives of this repo stored such a pickle); a pure function.
    _ALLOWED = {("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
                ("_codecs", "encode")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError("checkpoint metadata may not reference %s.%s" % (module, name))
        if name == "dtype":
            return np.dtype
        if name == "encode":
            import codecs
            return codecs.encode
        from numpy.core.multiarray import scalar   # resolves to numpy._core on numpy >= 2
        return scalar


def loads_metadata(raw):
    if isinstance(raw, np.ndarray):
        raw = raw.tobytes() if raw.dtype == np.uint8 else raw.item()   # uint8 vector: archives written by round 1 of this repo
    if isinstance(raw, str):
        raw = raw.encode("latin1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)             # numpy.core -> numpy._core alias
        return _MetadataUnpickler(io.BytesIO(bytes(raw)), encoding="latin1").load()


# ---- archives ------------------------------------------------------------------------------------------------------
def save_weights(fname, named_arrays, metadata=None, tmp_suffix=""):
    """named_arrays: dict name -> ndarray (unique names are guaranteed by the dict).  ``tmp_suffix`` makes the
    temporary file name unique when several processes could write the same checkpoint."""
    payload = {k: np.asarray(v) for k, v in named_arrays.items()}
    if metadata is not None:
        payload["metadata"] = np.array(dumps_py2(metadata))             # 0-d '|S<n>' array, as GANcheckpoints.py:21
    fname = str(fname)
    tmp = os.path.splitext(fname)[0] + tmp_suffix + ".tmp.npz"         # never a half-written file (GANcheckpoints.py:24-28)
    np.savez_compressed(tmp, **payload)
    os.replace(tmp, fname)


def load_weights(fname, specs, extra_prefixes=()):
    """specs: iterable of objects with .name and .shape.  Returns (dict name -> float32 array, metadata).
    ``extra_prefixes``: entries whose name starts with one of these are returned too, whatever their shape
    (the discriminator head is not part of the inference graph's parameter list)."""
    found = {}
    with np.load(str(fname), allow_pickle=False) as archive:
        names = set(archive.files)
        for p in specs:
            if p.name in names:
                arr = archive[p.name]
                if tuple(arr.shape) != tuple(p.shape):
                    warnings.warn("shape mismatch:%s stored:%s new:%s, skipping" % (p.name, arr.shape, p.shape))
                    continue
                found[p.name] = np.asarray(arr, np.float32)
            else:
                logging.warning("unable to load parameter %s from %s", p.name, fname)
        for k in names:
            if k != "metadata" and k not in found and any(k.startswith(pre) for pre in extra_prefixes):
                found[k] = np.asarray(archive[k], np.float32)
        metadata = loads_metadata(archive["metadata"]) if "metadata" in names else {}
    return found, metadata

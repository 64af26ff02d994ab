"""npz checkpoints in the reference's format (GANcheckpoints.py:11-57).

An archive maps Theano parameter names to arrays, plus an optional 'metadata' entry
holding a pickled dict (written by Python 2's cPickle, GANcheckpoints.py:21).  ``load_weights`` keeps the
reference's behaviour: match by name, warn and skip on a shape mismatch (:40-50), warn on a missing
entry (:53-54), return the metadata dict (:55-58).
"""
from __future__ import annotations

import logging
import os
import pickle
import warnings

import numpy as np


def save_weights(fname, named_arrays, metadata=None):
    """named_arrays: dict name -> ndarray (unique names are guaranteed by the dict)."""
    payload = {k: np.asarray(v) for k, v in named_arrays.items()}
    if metadata is not None:
        payload["metadata"] = np.frombuffer(pickle.dumps(metadata, protocol=2), dtype=np.uint8)
    fname = str(fname)
    if os.path.exists(fname):  # avoid half-written files (GANcheckpoints.py:24-28)
        tmp = os.path.splitext(fname)[0] + ".tmp.npz"
        np.savez_compressed(tmp, **payload)
        os.replace(tmp, fname)
    else:
        np.savez_compressed(fname, **payload)


def _decode_metadata(entry):
    raw = entry.tobytes() if isinstance(entry, np.ndarray) and entry.dtype != object else entry
    if isinstance(raw, np.ndarray):
        raw = raw.item()
    if isinstance(raw, str):
        raw = raw.encode("latin1")
    return pickle.loads(raw, encoding="latin1")


def load_weights(fname, specs):
    """specs: iterable of objects with .name and .shape.  Returns (dict name -> float32 array, metadata)."""
    found = {}
    with np.load(str(fname), allow_pickle=True) as archive:
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
        metadata = _decode_metadata(archive["metadata"]) if "metadata" in names else {}
    return found, metadata

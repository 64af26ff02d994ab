"""Writes tests/golden/py2_style_checkpoint.npz: a small archive laid out the way the reference's
GANcheckpoints.save_weights (GANcheckpoints.py:11-30) lays one out under Python 2 + numpy 1.x:

  * parameters keyed by their Theano names (a handful of IAN_simple tensors, tiny shapes are not possible --
    shapes must match the config -- so only small layers are included; the loader warns for the rest, as the
    reference does, GANcheckpoints.py:53-54);
  * 'metadata' = cPickle.dumps({...}) -- protocol 0 text in a Python-2 ``str`` -- which numpy stores as a 0-d
    '|S<n>' array; the dict is the one train_IAN.py:571 writes, including ``np.float32(learning_rate)``, whose
    Python-2 pickle is a ``numpy.core.multiarray.scalar`` reduce with a '\\x..'-escaped payload string.

Python 2 is not available in this container, so the pickle text below is written out by hand following cPickle's
protocol-0 grammar (pickletools.dis of it is committed in the test).  Run: python tests/golden/make_py2_checkpoint.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def py2_metadata_text(epoch=3, itr=1200, ts=1474000000.25, lr=np.float32(2e-4)):
    esc = "".join("\\x%02x" % b for b in np.float32(lr).tobytes())
    return ("(dp0\nS'epoch'\np1\nI%d\nsS'itr'\np2\nI%d\nsS'ts'\np3\nF%r\nsS'learning_rate'\np4\n"
            "cnumpy.core.multiarray\nscalar\np5\n(cnumpy\ndtype\np6\n(S'f4'\np7\nI0\nI1\ntp8\nRp9\n"
            "(I3\nS'<'\np10\nNNNI-1\nI-1\nI0\ntp11\nbS'%s'\np12\ntp13\nRp14\ns." % (epoch, itr, ts, esc)).encode("latin1")


def main():
    rs = np.random.RandomState(7)
    arrays = {
        "mu_bnorm.gamma": rs.uniform(0.5, 1.5, 100).astype(np.float32),
        "mu_bnorm.beta": rs.randn(100).astype(np.float32) * 0.1,
        "mu_bnorm.mean": rs.randn(100).astype(np.float32) * 0.1,
        "mu_bnorm.inv_std": rs.uniform(0.5, 2.0, 100).astype(np.float32),
        "enc_conv1.b": rs.randn(128).astype(np.float32) * 0.1,
        "dec_out.W": rs.randn(128, 3, 5, 5).astype(np.float32) * 0.02,
        "discrimi.W": rs.randn(1524, 1).astype(np.float32) * 0.02,
        "enc_conv1.W": rs.randn(64, 3, 5, 5).astype(np.float32),      # wrong shape on purpose: must warn and skip (:40-50)
    }
    arrays["metadata"] = np.array(py2_metadata_text())               # Python 2: np.asarray(str) -> 0-d '|S<n>'
    np.savez_compressed(os.path.join(HERE, "py2_style_checkpoint.npz"), **arrays)


if __name__ == "__main__":
    main()

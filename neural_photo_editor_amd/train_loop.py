"""Training-loop plumbing around the MI355X training step: the host-side logic of train_IAN.py:357-571
(chunked loader, learning-rate schedule, strict G/D alternation, per-chunk jsonl metrics, checkpoints with
metadata, --resume, the sample / reconstruction / interpolation grid) without Fuel, matplotlib or Theano.

The reference reads CelebA through Fuel (absent here); ``dataset`` is anything with ``num_examples`` and
``get_data(indices) -> uint8 (n,3,64,64)`` -- ``ArrayDataset`` wraps a numpy array.  All arithmetic on images
happens in ``trainer.Trainer`` (HIP); this module only sequences calls, like the reference's ``main``.
"""
from __future__ import annotations

import json
import logging
import os
import time
from collections import OrderedDict

import numpy as np

GEN_KEYS = ("gen_recon_loss", "gen_sample_loss", "pixel_loss", "feature_loss", "pixel_acc")      # train_IAN.py:291-297
DISCRIM_KEYS = ("discrim_g_loss", "discrim_d_loss", "discrim_acc", "pixel_loss", "pixel_acc")    # train_IAN.py:299-304


def to_tanh(x):
    """train_IAN.py:36-37: uint8 [0,255] -> [-1,1]"""
    return 2.0 * (np.asarray(x, np.float32) / 255.0) - 1.0


def from_tanh(x):
    """train_IAN.py:39-40"""
    return 255.0 * (np.asarray(x, np.float32) + 1) / 2.0


class ArrayDataset:
    def __init__(self, images_uint8):
        self.images = np.asarray(images_uint8)
        assert self.images.ndim == 4 and self.images.shape[1:] == (3, 64, 64), self.images.shape
        self.num_examples = len(self.images)

    def get_data(self, indices):
        return self.images[np.asarray(indices, np.int64)]


def data_loader(cfg, dataset, offset=0, shuffle=False, seed=42):
    """train_IAN.py:357-375: yields to_tanh'ed chunks of batch_size*batches_per_chunk images; the permutation is
    drawn from RandomState(seed) over num_examples-offset items and indexed from ``offset`` on (the reference's
    own arithmetic, kept as is)."""
    chunk = cfg["batch_size"] * cfg["batches_per_chunk"]
    rs = np.random.RandomState(seed)
    n = dataset.num_examples - offset
    index = rs.permutation(n) if shuffle else np.arange(n)
    for i in range(dataset.num_examples // chunk):
        sel = index[offset + chunk * i: offset + chunk * (i + 1)]
        if len(sel) < chunk:      # the reference would raise IndexError on the last chunk when offset > 0
            return
        yield to_tanh(dataset.get_data(sel))


def learning_rate_for(cfg, epoch, current):
    """train_IAN.py:441-452: dict schedule keyed by epoch, then multiplicative decay."""
    lr = current
    if isinstance(cfg["learning_rate"], dict) and epoch > 0 and epoch in cfg["learning_rate"]:
        lr = cfg["learning_rate"][epoch]
    if cfg.get("decay_rate") and epoch > 0:
        lr = lr * (1 - cfg["decay_rate"])
    return float(lr)


class MetricsLogger:
    """metrics_logging.py:11-27: append-only jsonl with a ``_stamp`` per record."""

    def __init__(self, fname, reinitialize=False):
        self.fname = str(fname)
        if reinitialize and os.path.exists(self.fname):
            os.remove(self.fname)

    def log(self, record=None, **kwargs):
        record = dict(record or {})
        record.update(kwargs)
        record["_stamp"] = time.time()
        with open(self.fname, "a") as fh:
            fh.write(json.dumps(record) + "\n")


def read_records(fname):
    """metrics_logging.py:30-40"""
    with open(str(fname)) as fh:
        return [json.loads(line) for line in fh if line.strip()]


def interpolation_latents(Ze):
    """train_IAN.py:553: 3 pairs of endpoints x 7 linear interpolants."""
    return np.asarray([Ze[2 * i] * (1 - j) + Ze[2 * i + 1] * j for i in range(3) for j in [x / 6.0 for x in range(7)]], np.float32)


def sample_grid(sample_fn, zfn, endpoints_uint8, num_latents, rng):
    """train_IAN.py:541-556 / sample_IAN.py:171-187: 27 random samples, then per endpoint pair
    [endpoint, 7 interpolants, endpoint] -> uint8 (54,3,64,64), laid out 6 rows x 9 columns by the caller."""
    samples = np.uint8(from_tanh(sample_fn(rng.randn(27, num_latents).astype(np.float32))))
    endpoints = np.uint8(endpoints_uint8)
    assert len(endpoints) == 6
    Ze = np.asarray(zfn(to_tanh(endpoints)))
    Z = interpolation_latents(Ze)
    rows = [np.insert(endpoints[2 * i:2 * (i + 1)], 1, np.uint8(from_tanh(sample_fn(Z[7 * i:7 * (i + 1)]))), axis=0) for i in range(3)]
    return np.append(samples, np.concatenate(rows, axis=0), axis=0)


def tile_grid(images, rows=6, cols=9):
    """discgen_utils.plot_image_grid without matplotlib: (rows*cols,3,h,w) -> uint8 (rows*h, cols*w, 3)"""
    n, c, h, w = images.shape
    assert n == rows * cols
    return images.reshape(rows, cols, c, h, w).transpose(0, 3, 1, 4, 2).reshape(rows * h, cols * w, c)


def train_chunk(trainer, cfg, x_chunk, itr, rng, to_device):
    """train_IAN.py:460-504: shuffle the chunk, draw Z, alternate update_gen / update_discrim.  Returns
    (mean metrics of the chunk, new itr).  ``eps`` (the Gaussian noise of l_Z_IAF, drawn inside the Theano graph in the
    reference, layers.py:433) is drawn here and passed in, which is what makes the step reproducible."""
    bs = cfg["batch_size"]
    num_batches = len(x_chunk) // bs
    index = rng.permutation(len(x_chunk))
    x_chunk = x_chunk[index]
    Z = rng.randn(len(x_chunk), cfg["num_latents"]).astype(np.float32)
    metrics = OrderedDict((k, []) for k in GEN_KEYS + DISCRIM_KEYS)
    for bi in range(num_batches):
        sl = slice(bi * bs, (bi + 1) * bs)
        eps = rng.randn(bs, cfg["num_latents"]).astype(np.float32)
        args = [to_device(a) for a in (x_chunk[sl], Z[sl], eps)]
        if itr % (cfg["update_ratio"] + 1) == 0:
            for k, v in zip(GEN_KEYS, trainer.update_gen(*args)):
                metrics[k].append(v)
        else:
            for k, v in zip(DISCRIM_KEYS, trainer.update_discrim(*args)):
                metrics[k].append(v)
        itr += 1
    return OrderedDict((k, float(np.mean(v))) for k, v in metrics.items() if v), itr


def train(cfg, trainer, dataset, weights_fname, metrics_fname=None, resume=False, max_epochs=None, to_device=None,
          checkpoint_fn=None, load_metadata=None):
    """train_IAN.py:378-571 main loop.  ``trainer``: update_gen / update_discrim / save_weights / lr (trainer.Trainer).
    ``checkpoint_fn(epoch)``: optional hook for the sample grid.  ``load_metadata``: metadata dict of the checkpoint
    being resumed (epoch, learning_rate) -- the weights themselves are loaded by whoever built the trainer."""
    to_device = to_device or (lambda a: a)
    metrics_fname = metrics_fname or str(weights_fname)[:-4] + "METRICS.jsonl"
    # data parallel: every rank runs this loop on its shard; the files (metrics log, sample grid, checkpoint) belong to
    # rank 0 alone, the other ranks wait at a barrier so nobody races ahead of a half-written checkpoint
    comm = getattr(trainer, "comm", None)
    rank = getattr(comm, "rank", 0)
    barrier = getattr(comm, "barrier", None) or (lambda: None)
    mlog = MetricsLogger(metrics_fname, reinitialize=not resume) if rank == 0 else None
    itr = 0
    min_epoch = 0
    if resume and load_metadata:
        min_epoch = load_metadata["epoch"] + 1 if "epoch" in load_metadata else 0
        if "learning_rate" in load_metadata:
            trainer.lr = float(load_metadata["learning_rate"])
    offset = True
    rng = np.random.RandomState(cfg.get("seed", 0))
    for epoch in range(min_epoch, max_epochs if max_epochs is not None else cfg["max_epochs"]):
        offset = not offset
        loader = data_loader(cfg, dataset, offset=int(offset) * cfg["batch_size"] // 2, shuffle=cfg["shuffle"], seed=epoch)
        new_lr = learning_rate_for(cfg, epoch, trainer.lr)
        if new_lr != trainer.lr:
            logging.info("Changing learning rate from %s to %s", trainer.lr, new_lr)
            trainer.lr = new_lr
        for x_chunk in loader:
            metrics, itr = train_chunk(trainer, cfg, x_chunk, itr, rng, to_device)
            logging.info("%4d %6d  %s", epoch, itr, "  ".join("%s %.4f" % kv for kv in metrics.items()))
            if mlog:
                mlog.log(epoch=epoch, itr=itr, metrics=metrics)
        if not (epoch % cfg["checkpoint_every_nth"]):
            if rank == 0:
                if checkpoint_fn:
                    checkpoint_fn(epoch)
                trainer.save_weights(weights_fname, {"epoch": epoch, "itr": itr, "ts": time.time(), "learning_rate": float(trainer.lr)})
            barrier()
    logging.info("training done")
    return itr

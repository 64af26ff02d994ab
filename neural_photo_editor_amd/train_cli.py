"""Command-line entry mirroring ``python train_IAN.py <config> [--resume]`` (train_IAN.py:573-581) on the MI355X path.

    python -m neural_photo_editor_amd.train_cli configs/IAN.py --data celeba64_uint8.npy [--resume] [--epochs N]
    torchrun --nproc-per-node 8 -m neural_photo_editor_amd.train_cli configs/IAN.py --data ... (data-parallel, RCCL)

``--data`` is a ``.npy`` / ``.npz`` (key ``images``) uint8 array (N,3,64,64): the reference's Fuel/HDF5 CelebA reader
is not available (out of scope), any array source with ``num_examples`` / ``get_data`` works through
``train_loop.train``.  Without ``--data`` a seeded synthetic set is used (smoke runs).
"""
from __future__ import annotations

import argparse
import logging
import os

import numpy as np


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("config_path", help="config .py file (IAN.py)")
    ap.add_argument("--resume", action="store_true", help="continue from <config>.npz (train_IAN.py:423-430)")
    ap.add_argument("--data", default=None, help=".npy/.npz uint8 images (N,3,64,64)")
    ap.add_argument("--epochs", type=int, default=None, help="override cfg['max_epochs']")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU minibatch (default cfg['batch_size'] / world: the reference's batch is the global one)")
    ap.add_argument("--local-statistics", action="store_true",
                    help="data parallel without SyncBN / MinibatchLayer all-gather (faster, not the reference's arithmetic)")
    return ap.parse_args(argv)


def load_images(path, n_synth=256):
    if path is None:
        from . import synthetic
        return np.uint8((synthetic.make_images(n_synth, seed=123) + 1.0) * 127.5)
    arr = np.load(path, allow_pickle=False)
    if hasattr(arr, "files"):
        arr = arr["images"]
    arr = np.asarray(arr)
    if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[1:] != (3, 64, 64):
        raise ValueError("--data must hold uint8 images of shape (N,3,64,64), got %s %s" % (arr.dtype, arr.shape))
    return arr


def initial_params(specs, seed=0):
    """name -> float32 array drawn from each parameter's recorded lasagne initialiser (config_loader.Init.sample)."""
    rs = np.random.RandomState(seed)
    return {p.name: p.init.sample(p.shape, rs) for p in specs}


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s| %(message)s")
    import torch
    import torch.distributed as dist
    from . import checkpoints, config_loader, lowering, train_loop
    from .trainer import Trainer, default_comm
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    mod = config_loader.load_config(args.config_path)
    cfg = dict(mod.cfg)
    weights_fname = str(args.config_path)[:-3] + ".npz"                      # train_IAN.py:384
    model = config_loader.build_model(mod)
    specs = lowering.all_param_specs(model)                                  # l_out + l_Z + l_discrim + batch-norm statistics
    # A fresh run starts from the config's own initialisers (IAN.py: Normal(0.02) filters, Orthogonal('relu') MADE,
    # gamma 1 / beta 0, mean 0 / inv_std 1, theta ~ N(0,0.05), b = -1: what lasagne would draw for train_IAN.py:381-392),
    # seeded per config so that every rank of a data-parallel job draws the same values.
    params, meta = initial_params(specs, cfg.get("seed", 0)), {}
    if args.resume and os.path.isfile(weights_fname):
        loaded, meta = checkpoints.load_weights(weights_fname, specs)
        params.update(loaded)
    # the reference's batch_size is the GLOBAL minibatch (one Theano function call): each rank takes 1/world of it
    if args.batch is None and cfg["batch_size"] % world:
        raise SystemExit("cfg['batch_size']=%d is not divisible by the %d ranks; pass --batch" % (cfg["batch_size"], world))
    batch = args.batch or cfg["batch_size"] // world
    cfg["batch_size"] = batch * world
    comm = default_comm()     # N > 1: the torch-free librccl filler (falls back to torch.distributed on all ranks together)
    trainer = Trainer(args.config_path, params, batch=batch, comm=comm, exact=not args.local_statistics)
    if world > 1:
        logging.info("data parallel: %d ranks, collectives through %s", world, comm.filler)
    images = load_images(args.data)
    rank = int(os.environ.get("RANK", "0"))

    class Shard(train_loop.ArrayDataset):                                    # every rank walks the same chunks, takes its rows
        pass
    ds = Shard(images)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a[rank * batch:(rank + 1) * batch] if len(a) == batch * world else a)).cuda()
    train_loop.train(cfg, trainer, ds, weights_fname, resume=args.resume, max_epochs=args.epochs, to_device=to_dev,
                     load_metadata=meta)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

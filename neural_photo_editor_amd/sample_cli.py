"""Command-line entry mirroring ``python sample_IAN.py <config>`` (sample_IAN.py:135-202) on the MI355X path: a 6x9 grid of
27 random samples, then for each of 3 test-image pairs [original, 7 latent interpolants, original].

    python -m neural_photo_editor_amd.sample_cli configs/IAN.py [--data test_uint8.npy] [--out grid.ppm] [--seed 42]

Weights come from ``<config>.npz`` (GANcheckpoints format); when that file is absent (the reference ships git-LFS
pointers) seeded synthetic parameters are used and a warning is printed.  The grid is written as a binary PPM (and
``.npy``): matplotlib / PIL are not required.
"""
from __future__ import annotations

import argparse
import os
import warnings

import numpy as np


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("config_path")
    ap.add_argument("--data", default=None, help=".npy uint8 test images (N>=6,3,64,64); default: seeded synthetic images")
    ap.add_argument("--out", default=None, help="output .ppm (default <config>_sample.ppm)")
    ap.add_argument("--seed", type=int, default=42)
    return ap.parse_args(argv)


def write_ppm(path, rgb_hwc_uint8):
    h, w, _ = rgb_hwc_uint8.shape
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(np.ascontiguousarray(rgb_hwc_uint8, np.uint8).tobytes())


def main(argv=None):
    args = parse_args(argv)
    from . import IAN, synthetic, train_loop
    weights = str(args.config_path)[:-3] + ".npz"
    params = None
    if not os.path.exists(weights):
        warnings.warn("%s not found: sampling from seeded synthetic parameters" % weights)
        arch = "IAN" if "l_IAF" in open(args.config_path).read() else "IAN_simple"
        params = synthetic.make_params(arch, seed=1)
    model = IAN(args.config_path, True, params=params)
    rng = np.random.RandomState(args.seed)
    if args.data:
        test = np.load(args.data)
    else:
        test = np.uint8((synthetic.make_images(16, seed=7) + 1.0) * 127.5)
    endpoints = test[rng.choice(len(test), 6, replace=False)]                     # sample_IAN.py:176-178
    imgs = train_loop.sample_grid(model.sample, model.Zfn, endpoints, model.get_zdim(), rng)
    grid = train_loop.tile_grid(imgs, 6, 9)
    out = args.out or str(args.config_path)[:-3] + "_sample.ppm"
    write_ppm(out, grid)
    np.save(os.path.splitext(out)[0] + ".npy", imgs)
    return out


if __name__ == "__main__":
    print(main())

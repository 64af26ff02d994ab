"""Generates the committed golden vectors under tests/golden/.

The reference itself cannot run here (no Theano/Lasagne, no weights: SURVEY section 0, 8c), so these are
outputs of the CPU oracle (oracle/ian_oracle.py float32 restatement, gradients from the torch-CPU twin
in float64) on seeded synthetic parameters and inputs -- the oracle itself is pinned by the reference-executed fixtures (make_ref_golden.py); these pin it against
regressions and give the GPU tests fixed targets.  The MADE entries are the one reference-derived
known answer.   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ian_oracle as O  # noqa: E402
from oracle.torch_twin import TorchTwin  # noqa: E402
import torch  # noqa: E402

BRUSH = (26, 26, 30, 30)  # (c1, r1, c2, r2): default NPE brush, SURVEY 8d


def red_rgb():
    rgb = np.full((1, 3, 64, 64), -1.0, np.float32)
    rgb[:, 0] = 1.0
    return rgb


def main():
    for arch in O.ARCHS:
        P = O.make_params(arch, seed=1)
        orc = O.Oracle(arch, P)
        x = O.make_images(2, seed=0)
        zpre = orc.Zfn(x)
        z = orc.encode_images(x)
        xhat = orc.sample_at(z)
        zs = O.make_latents(2, seed=2)
        xs = orc.sample_at(zs)
        tw = TorchTwin(arch, P, dtype=torch.float64)
        c1, r1, c2, r2 = BRUSH
        g_rgb = tw.imgradRGB(c1, r1, c2, r2, red_rgb(), zs[:1]).astype(np.float32)
        g_light = tw.imgrad(c1, r1, c2, r2, zs[:1]).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "%s_seed1.npz" % arch), zpre=zpre, z=z, xhat=xhat, z_sample=zs,
                            x_sample=xs, grad_rgb=g_rgb, grad_light=g_light)
        print(arch, "z std %.4f" % z.std(), "xhat std %.4f" % xhat.std(), "|g_rgb| %.3e" % np.abs(g_rgb).max())
    ordering, child = O.made_ordering()
    M0, M1, MD = O.made_masks()
    np.savez_compressed(os.path.join(HERE, "made_masks.npz"), ordering=ordering, child_seed=child,
                        packed=np.packbits(np.stack([M0, M1, MD]).astype(np.uint8)))


if __name__ == "__main__":
    main()

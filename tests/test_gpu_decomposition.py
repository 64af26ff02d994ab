"""Where is the training step's gradient error born?  (round-3 verdict, weak #1 / next-round item 1b)

The composed comparison "HIP float32 step vs float64 twin" on the synthetic fixture is ill-conditioned for ANY float32
implementation: a float32 run of the reference's own graph moves single gradient tensors by up to 17 % (fixture 'noise32'),
1-ulp perturbations of the layer outputs by 1e-2..6e-2 (scripts/exp/fp32_noise_conditioning.py) -- leaky-ReLU / |.| kinks and
batch statistics over 4 near-identical decoder outputs.  This test removes that amplification instead of widening a bar:
oracle/staged_twin.py evaluates the float64 graph AT THE HIP STEP'S OWN FORWARD POINT (every stage's output replaced by the
HIP activation, straight-through; kink branches taken from the HIP activations) and

  * each stage's LOCAL forward error (HIP output vs float64 stage applied to the HIP input) is held to 1e-4, the north-star
    tolerance on float32 activations -- the 91 named stages of the five passes, batch-statistics batch-norm included;
  * every gradient tensor of the three parameter groups is held to the float64 gradient at that point: what is left is the
    arithmetic of the backward kernels alone (tap-GEMM backward-data / backward-weight, batch-norm backward, MinibatchLayer,
    losses), which is what a backward bug would show up in.

Measured values go to gpurun_out/diag/decomposition_*.json (copied to profiles/ per round)."""
import json
import os

import numpy as np
import pytest

from oracle import ian_oracle as O
from oracle.train_twin import make_train_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs", "IAN.py")
GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def hip_provider(tr):
    """(pass, name) of oracle/staged_twin.py -> the HIP trainer's buffer as a float64 CPU tensor, NCHW / (n, features), real
    channels only (the trainer stores NHWC with the channel stride rounded up to 32)."""
    import torch
    passes = {"EX": tr.EX, "EH": tr.EH, "EG": tr.EG, "ZS": tr.ZS, "DZ": tr.DZ, "DG": tr.DG}
    widths = {"a1": 128, "y2": 256, "a2": 256, "y3": 512, "a3": 512, "y4": 1024, "a4": 1024}
    cache = {}

    def nchw(t, c):
        return t[..., :c].permute(0, 3, 1, 2).contiguous()

    def get(tag, name):
        key = (tag, name)
        if key in cache:
            return cache[key]
        B = passes[tag]
        n = tr.n
        if tag[0] == "E":
            if name in widths:
                v = nchw(B[name], widths[name])
            elif name == "feat":
                v = B["feat"][:, :1024]
            elif name == "act":
                v = B["act"][:, :2500].reshape(n, 500, 5)
            elif name == "mbf":
                v = B["mb"][:, 1024:1524]
            else:
                v = B[name]
        elif tag == "ZS":
            v = B[name][:, :1000 if name in ("y_fc1", "f") else 100]
        else:
            if name == "h0":
                v = nchw(B["h0"], 512).reshape(n, 8192)
            elif name in ("R", "G", "B"):
                v = nchw(B[name], 2)
            elif name == "xhat":
                v = B["xhat"]
            elif name in ("y4", "h4"):
                v = nchw(B[name], 128)
            else:
                t = B[name]
                v = nchw(t, t.shape[-1])
        v = cache[key] = v.detach().to("cpu", torch.float64).contiguous()
        return v
    return get


def _diag(name, obj):
    d = os.path.join(ROOT, "gpurun_out", "diag")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as fh:
            json.dump(obj, fh, indent=1)
    except OSError:
        pass


def _inputs(case):
    if case == "fixture4":          # the reference-executed fixture's minibatch (tests/golden/ref_train_IAN.npz): the worst-conditioned case
        fx = np.load(os.path.join(GOLD, "ref_train_IAN.npz"))
        B = int(fx["batch"])
        return fx["X"][:B].astype(np.float32), fx["Z"][:B].astype(np.float32), fx["gen/eps"].astype(np.float32)
    B = {"synthetic16": 16, "synthetic32": 32}[case]
    return O.make_images(B, seed=31), O.make_latents(B, seed=32), np.random.RandomState(33).randn(B, 100).astype(np.float32)


# (synthetic16, discrim) measured as well in round 4 (profiles/r04_decomposition.json: 5.2e-7 / 2.2e-6); left out of the suite for time.
# synthetic32 (round-4 verdict item 6): a quarter of the benchmarked batch, ~2 min of float64 twin; the 128-image case stays a script
# (scripts/exp/decomposition_gpu.py, 7 min) whose record is committed per round (profiles/r0N_decomposition_b128_gen.json).
# Round 6 (suite budget, round-5 verdict item 7): synthetic16 left the suite -- synthetic32 proves the same identity at twice the batch
# (round-5 record of both, gradient vs float64 at the HIP forward point, median / worst: 1.5e-6 / 3.0e-5 at 16 images, 1.6e-6 / 4.8e-6
# at 32 -- profiles/r05_decomposition.json).
@pytest.mark.parametrize("case,which", [("fixture4", "gen"), ("fixture4", "discrim"), ("synthetic32", "gen")])
def test_gradient_error_is_born_in_the_forward_conditioning_not_in_the_backward_kernels(case, which):
    import torch
    from oracle.staged_twin import StagedTwin
    from neural_photo_editor_amd.trainer import Trainer
    P = make_train_params(O.make_params("IAN", 1))
    X, Z, eps = _inputs(case)
    B = X.shape[0]
    tr = Trainer(CFG, P, batch=B)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tr.forward(dev(X), dev(Z), dev(eps))
    m = tr.metrics()
    tr.backward(which)
    tr._finish_allreduce(which)
    tr._regularizers(which)
    torch.cuda.synchronize()
    groups = ("dec", "Z") if which == "gen" else ("enc", "Z")
    got = {g: tr.grads_numpy(g) for g in groups}
    tw = StagedTwin(P, dtype=torch.float64)
    plain, Lp = tw.gradients_staged(X, Z, eps, which)                            # the float64 graph at ITS OWN forward point
    at, La = tw.gradients_staged(X, Z, eps, which, provider=hip_provider(tr))    # ... at the HIP step's forward point
    local = sorted(((v, "%s.%s" % k) for k, v in tw.local_err.items()), reverse=True)
    e_plain = sorted(((rel(got[g][n], plain[g][n].numpy()), n) for g in groups for n in plain[g]), reverse=True)
    e_at = sorted(((rel(got[g][n], at[g][n].numpy()), n) for g in groups for n in at[g]), reverse=True)
    report = {"batch": B, "which": which,
              "local_forward_error": {"stages": len(local), "median": float(np.median([e for e, _ in local])), "worst": local[:6]},
              "grad_vs_plain_float64": {"median": float(np.median([e for e, _ in e_plain])), "worst": e_plain[:6]},
              "grad_vs_float64_at_hip_forward_point": {"median": float(np.median([e for e, _ in e_at])), "worst": e_at[:6]},
              "loss_vs_plain_float64": {k: abs(m[k] - Lp[k]) / max(1.0, abs(Lp[k])) for k in m if k in Lp},
              "loss_at_hip_point_vs_hip": {k: abs(m[k] - La[k]) / max(1.0, abs(La[k])) for k in m if k in La}}
    _diag("decomposition_%s_%s" % (case, which), report)
    assert len(local) == 91          # 3 encoder passes x 11 + 8 latent + 2 decoder passes x 25 named stages
    # every stage of the training forward, taken alone: the north star asks 1e-4 on float32 activations; measured 2.7e-6 worst,
    # 1.8e-7 median (round 4, MI355X) -> held to 1e-5 / 1e-6
    assert local[0][0] < 1e-5, local[:6]
    assert float(np.median([e for e, _ in local])) < 1e-6, local[:6]
    # the losses the step reports, against the float64 losses at the same activations (round-3 verdict: 1e-4, not 2e-4)
    for k, v in report["loss_at_hip_point_vs_hip"].items():
        assert v < 1e-5, (k, v)                                   # measured 1.3e-7
    for k, v in report["loss_vs_plain_float64"].items():
        assert v < 2e-5, (k, v)                                   # measured 1.4e-6: the forward losses are well conditioned
    # the backward kernels alone: measured median 5e-7 .. 1.5e-6, worst tensor 3.1e-5 (a 2-element MDCL coefficient gradient,
    # an inner product of ~10^5 terms) -> median at float32 round-off, every tensor within 2e-4
    assert float(np.median([e for e, _ in e_at])) < 1e-5, e_at[:6]
    assert e_at[0][0] < 2e-4, e_at[:6]

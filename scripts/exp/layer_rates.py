"""Per-layer GEMM rates of the training step IN ISOLATION (one stream, nothing else on the chip): forward, backward-data and
backward-weight of every tap-GEMM layer of IAN.py at the benchmarked per-GPU batch, after the layer autotune.  The training
profile (profiles/r0N_train_ian_b128.md) shows the same kernels while the weight-gradient stream shares the chip with the data
path; the difference is what the overlap costs each kernel.
  python scripts/exp/layer_rates.py [batch=128]  ->  gpurun_out/r06_layer_rates.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neural_photo_editor_amd import trainer as T  # noqa: E402
from neural_photo_editor_amd.lib import load_train_library  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = load_train_library()
PEAK = 157.3


def taps_of(scales):   # composite stencil: centre shared (layers.py:207-258)
    return 1 + 8 * (1 + sum(1 for s in scales if s > 0))


LAYERS = []   # (name, kind, cin, cout, in_h, scales, macs per image)
cin = 3
for i, w in enumerate(T.ENC_WIDTHS):
    hw = 64 >> i
    LAYERS.append(("enc_conv%d" % (i + 1), T.K_CONV, cin, w, hw, [], (hw // 2) ** 2 * w * cin * 25))
    cin = w
for dc, ci, co, hw, blk, sc in T.DEC_STAGES:
    LAYERS.append((dc, T.K_DECONV, ci, co, hw, [], hw * hw * ci * co * 25))
    LAYERS.append((blk, T.K_MDC, co, co, 2 * hw, sc, (2 * hw) ** 2 * co * co * taps_of(sc)))
LAYERS.append(("dec_conv4", T.K_DECONV, 128, 128, 32, [], 32 * 32 * 128 * 128 * 25))


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


out = []
g = torch.Generator(device="cuda").manual_seed(0)
for name, kind, ci, co, hw, sc, macs in LAYERS:
    if name == "enc_conv1":
        continue      # edge layer: VALU / its own MFMA kernel, not tapgemm
    L = T.Layer(lib, kind, ci, co, hw, hw, scales=sc)
    oh = hw // 2 if kind == T.K_CONV else (2 * hw if kind == T.K_DECONV else hw)
    x = torch.randn(n, hw, hw, T.cs(ci), device="cuda", generator=g)
    y = torch.empty(n, oh, oh, T.cs(co), device="cuda")
    dy = torch.randn(n, oh, oh, T.cs(co), device="cuda", generator=g)
    dx = torch.empty_like(x)
    if kind == T.K_MDC:
        params = [torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.02] + [torch.full((co,), 1.0 / (1 + len(sc)), device="cuda") for _ in range(1 + len(sc))]
    elif kind == T.K_CONV:
        params = [torch.randn(co, ci, 5, 5, device="cuda", generator=g) * 0.02]
    else:
        params = [torch.randn(ci, co, 5, 5, device="cuda", generator=g) * 0.02]
    L.set_params(params)
    sa, sb = torch.randn(x.numel() if x.numel() > dy.numel() else dy.numel(), device="cuda", generator=g), None
    sb = torch.randn_like(sa)
    L.autotune(n, sa, sb)
    dparams = [torch.zeros_like(p) for p in params]
    flop = 2.0 * macs * n
    rec = {"layer": name, "gflop": flop / 1e9}
    for what, fn in (("forward", lambda: L.forward(x, n, y)), ("backward_data", lambda: L.backward_data(dy, n, dx)),
                     ("backward_weight", lambda: L.backward_weight(x, dy, n, dparams))):
        us = timed(fn)
        rec[what + "_us"] = us
        rec[what + "_frac_of_peak"] = flop / (us * 1e-6) / 1e12 / PEAK
    out.append(rec)
    print("%-12s %7.1f GFLOP | fwd %7.1f us %.2f | bwd-data %7.1f us %.2f | bwd-weight %7.1f us %.2f" % (
        name, rec["gflop"], rec["forward_us"], rec["forward_frac_of_peak"], rec["backward_data_us"], rec["backward_data_frac_of_peak"],
        rec["backward_weight_us"], rec["backward_weight_frac_of_peak"]), flush=True)
    L.close()
    del x, y, dy, dx, sa, sb
tot = {k: sum(r[k + "_us"] for r in out) for k in ("forward", "backward_data", "backward_weight")}
print("sums (us):", tot)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"batch": n, "layers": out, "sum_us": tot, "peak_tflops": PEAK}, open(os.path.join(ROOT, "gpurun_out", "r06_layer_rates.json"), "w"), indent=1)

"""Backward-weight of every tap-GEMM layer of the training step IN ISOLATION, for rocprofv3 (kernel trace / PMC passes):
one autotune, a marker launch, then REPS backward_weight calls per layer.  scripts/exp/wgrad_probe_summary.py reads the traces.
  [B=128] [REPS=5] [LAYERS=dec_conv1,enc_conv2] python scripts/exp/wgrad_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neural_photo_editor_amd import trainer as T  # noqa: E402
from neural_photo_editor_amd.lib import load_train_library  # noqa: E402

n = int(os.environ.get("B", "128"))
reps = int(os.environ.get("REPS", "5"))
only = [s for s in os.environ.get("LAYERS", "").split(",") if s]
lib = load_train_library()
PEAK = 157.3


def taps_of(scales):
    return 1 + 8 * (1 + sum(1 for s in scales if s > 0))


LAYERS = []
cin = 3
for i, w in enumerate(T.ENC_WIDTHS):
    hw = 64 >> i
    LAYERS.append(("enc_conv%d" % (i + 1), T.K_CONV, cin, w, hw, [], (hw // 2) ** 2 * w * cin * 25))
    cin = w
for dc, ci, co, hw, blk, sc in T.DEC_STAGES:
    LAYERS.append((dc, T.K_DECONV, ci, co, hw, [], hw * hw * ci * co * 25))
    LAYERS.append((blk, T.K_MDC, co, co, 2 * hw, sc, (2 * hw) ** 2 * co * co * taps_of(sc)))
LAYERS.append(("dec_conv4", T.K_DECONV, 128, 128, 32, [], 32 * 32 * 128 * 128 * 25))

g = torch.Generator(device="cuda").manual_seed(0)
objs = []
for name, kind, ci, co, hw, sc, macs in LAYERS:
    if name == "enc_conv1" or (only and name not in only):
        continue
    L = T.Layer(lib, kind, ci, co, hw, hw, scales=sc)
    oh = hw // 2 if kind == T.K_CONV else (2 * hw if kind == T.K_DECONV else hw)
    x = torch.randn(n, hw, hw, T.cs(ci), device="cuda", generator=g)
    dy = torch.randn(n, oh, oh, T.cs(co), device="cuda", generator=g)
    if kind == T.K_MDC:
        params = [torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.02] + [torch.full((co,), 1.0 / (1 + len(sc)), device="cuda") for _ in range(1 + len(sc))]
    elif kind == T.K_CONV:
        params = [torch.randn(co, ci, 5, 5, device="cuda", generator=g) * 0.02]
    else:
        params = [torch.randn(ci, co, 5, 5, device="cuda", generator=g) * 0.02]
    L.set_params(params)
    dparams = [torch.zeros_like(p) for p in params]
    L.backward_weight(x, dy, n, dparams)     # builds the schedule + workspace
    objs.append((name, L, x, dy, dparams, 2.0 * macs * n))
torch.cuda.synchronize()
res = []
for name, L, x, dy, dparams, flop in objs:
    torch.arange(7777, device="cuda")        # marker launch before every layer
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        L.backward_weight(x, dy, n, dparams)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    res.append({"layer": name, "gflop": flop / 1e9, "backward_weight_us": us, "frac_of_peak": flop / (us * 1e-6) / 1e12 / PEAK})
    print("%-12s %7.1f GFLOP  bwd-weight %8.1f us  %.3f of peak" % (name, flop / 1e9, us, res[-1]["frac_of_peak"]), flush=True)
out = os.environ.get("OUT")
if out:
    json.dump({"batch": n, "layers": res, "opts": os.environ.get("IAN_OPTS", "")}, open(out, "w"), indent=1)

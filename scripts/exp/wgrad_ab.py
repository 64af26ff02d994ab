"""Backward-weight variants in ONE process: bitwise equality and isolated timing per layer.
  variants: base = wg_pipe=0,wg_reduce_tiled=0 (round 5's kernels) | p1 = pinned two-buffer K loop + tiled reduce | p2 = rotated
  three-buffer K loop + tiled reduce.   [B=128] [REPS=10] python scripts/exp/wgrad_ab.py  ->  gpurun_out/r06_wgrad_ab.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neural_photo_editor_amd import trainer as T  # noqa: E402
from neural_photo_editor_amd.lib import load_train_library  # noqa: E402

n = int(os.environ.get("B", "128"))
reps = int(os.environ.get("REPS", "10"))
lib = load_train_library()
PEAK = 157.3
VARIANTS = [("base", "wg_pipe=0,wg_reduce_tiled=0"), ("p1", "wg_pipe=1,wg_reduce_tiled=1"), ("p2", "wg_pipe=2,wg_reduce_tiled=1"), ("p3", "wg_pipe=3,wg_reduce_tiled=1"), ("p3_tci32", "wg_pipe=3,wg_reduce_tiled=1,wg_reduce_tci=32"),
            ("p1_t512", "wg_pipe=1,wg_reduce_tiled=1,wg_target_items=512"), ("p1_t1536", "wg_pipe=1,wg_reduce_tiled=1,wg_target_items=1536")]


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
LAYERS.append(("enc_fc1", T.K_DENSE, 16384, 1024, 1, [], 16384 * 1024))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
out, bad = [], 0
for name, kind, ci, co, hw, sc, macs in LAYERS:
    if name == "enc_conv1":
        continue
    oh = hw // 2 if kind == T.K_CONV else (2 * hw if kind == T.K_DECONV else hw)
    if kind == T.K_DENSE:
        x = torch.randn(n, T.cs(ci), device="cuda", generator=g)
        dy = torch.randn(n, T.cs(co), device="cuda", generator=g)
    else:
        x = torch.randn(n, hw, hw, T.cs(ci), device="cuda", generator=g)
        dy = torch.randn(n, oh, oh, T.cs(co), device="cuda", generator=g)
    if kind == T.K_MDC:
        params = [torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.02] + [torch.full((co,), 1.0 / (1 + len(sc)), device="cuda") for _ in range(1 + len(sc))]
    elif kind == T.K_CONV:
        params = [torch.randn(co, ci, 5, 5, device="cuda", generator=g) * 0.02]
    elif kind == T.K_DENSE:
        params = [torch.randn(ci, co, device="cuda", generator=g) * 0.02]
    else:
        params = [torch.randn(ci, co, 5, 5, device="cuda", generator=g) * 0.02]
    rec = {"layer": name, "gflop": 2.0 * macs * n / 1e9}
    ref = None
    for vname, opts in VARIANTS:
        os.environ["IAN_OPTS"] = opts
        L = T.Layer(lib, kind, ci, co, hw, hw, scales=sc) if kind != T.K_DENSE else T.Layer(lib, kind, ci, co)
        L.set_params(params)
        dparams = [torch.full_like(p, 0.5) for p in params]
        L.backward_weight(x, dy, n, dparams)                      # overwrite
        L.backward_weight(x, dy, n, dparams, accumulate=True)     # + accumulate
        torch.cuda.synchronize()
        got = [d.clone() for d in dparams]
        if ref is None:
            ref = got
        else:
            same = all(torch.equal(a, b) for a, b in zip(ref, got))
            rec[vname + "_bitwise_equal_to_base"] = bool(same)
            if not same:
                bad += 1
                rec[vname + "_max_abs_diff"] = max(float((a - b).abs().max()) for a, b in zip(ref, got))
        us = timed(lambda: L.backward_weight(x, dy, n, dparams))
        rec[vname + "_us"] = us
        rec[vname + "_frac_of_peak"] = 2.0 * macs * n / (us * 1e-6) / 1e12 / PEAK
        L.close()
    out.append(rec)
    print("%-11s %6.1f GF | " % (name, rec["gflop"]) + " | ".join("%s %7.1f us %.3f%s" % (v, rec[v + "_us"], rec[v + "_frac_of_peak"],
          "" if v == "base" else (" =" if rec[v + "_bitwise_equal_to_base"] else " DIFFERENT")) for v, _ in VARIANTS), flush=True)
tot = {v: sum(r[v + "_us"] for r in out) for v, _ in VARIANTS}
print("sums (us):", tot, "  mismatching variants:", bad)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"batch": n, "variants": dict(VARIANTS), "layers": out, "sum_us": tot, "mismatches": bad}, open(os.path.join(ROOT, "gpurun_out", "r06_wgrad_ab.json"), "w"), indent=1)
sys.exit(1 if bad else 0)

#!/usr/bin/env python
"""rocprofv3 kernel trace of scripts/train_profile.py -> profiles/<name>.md: per-kernel totals of the steady-state steps
(everything after the marker launch: autotune candidates, layer construction and warm-up are cut away).

usage: python scripts/summarize_train_profile.py gpurun_out/<tag>/trace/trace_kernel_trace.csv profiles/<name> [bench.json]

If gpurun_out/<tag>/pmc_FETCH_SIZE and pmc_WRITE_SIZE exist (scripts/profile_train.sh: separate --pmc passes of the same
command), the HBM traffic of the steady-state updates is added: per update and per kernel family, FETCH_SIZE x2-corrected
as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests of wide coalesced reads are counted as 64 B), WRITE_SIZE as
reported; a JSON twin (<name>.json) carries hbm_bytes_per_update for bench.py's train_step.roofline.traffic."""
import collections, csv, json, sys

src, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = max(i for i, r in enumerate(rows) if "arange" in r["Kernel_Name"].lower())
rows = rows[mark + 1:]
tot = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("void ian::", "").replace("ian::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = tot.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
total = sum(v[1] for v in tot.values())
by_stream = collections.OrderedDict()
for r in rows:
    by_stream[r.get("Stream_Id", "?")] = by_stream.get(r.get("Stream_Id", "?"), 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
iters = sum(1 for r in rows if "adam_kernel" in r["Kernel_Name"]) // 4 or 1   # 4 parameter groups... per update: see below
gemm = sum(v[1] for k, v in tot.items() if k.startswith("tapgemm_kernel") or k.startswith("tapwgrad_kernel"))
train = None
if len(sys.argv) > 3:
    train = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]).get("train_step")
with open(out + ".md", "w") as f:
    f.write("# rocprofv3 summary: %s\n\n" % out.split("/")[-1])
    f.write("Command: `B=128 ITERS=4 rocprofv3 --kernel-trace --stats -- python scripts/train_profile.py` -- 4 updates of the full-IAN\n"
            "training step (update_gen, update_discrim, update_gen, update_discrim; train_IAN.py:309-329) at 128 images per GPU, 1 GPU,\n"
            "synthetic data, after the layer autotune and two warm-up updates (cut at the marker launch by\n"
            "scripts/summarize_train_profile.py).\n\n")
    f.write("Sum of kernel durations: **%.1f ms** for 4 updates (%.1f ms per update; first-to-last-kernel span %.1f ms).\n" % (total / 1e3, total / 4e3, span))
    if total / 1e3 > 1.02 * span:
        f.write("The sum exceeds the span because two streams run concurrently (`overlap_wgrad`: weight-gradient GEMMs + their split\n"
                "reduces on the trainer's second stream, everything else on the compute stream); a kernel that shares the chip with a\n"
                "GEMM of the other stream reports a longer duration than it has alone (e.g. `colstats`, `bn_bwd_finish`).  Per-stream\n"
                "busy time: %s.\n" % ", ".join("stream %s %.1f ms" % (sid, ms / 1e3) for sid, ms in sorted(by_stream.items())))
    if train:
        f.write("bench.py `train_step` of the same commit: update_gen %.1f ms, update_discrim %.1f ms wall, %.0f images/s.\n"
                % (train["update_gen_ms"], train["update_discrim_ms"], train["images_per_s"]))
    f.write("\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        if v[1] / total < 0.0005:
            continue
        f.write("| %s | %d | %.2f | %.1f | %.2f |\n" % (k[:100], v[0], v[1] / 1e3, v[1] / v[0], 100 * v[1] / total))
    f.write("\ntapgemm + tapwgrad (fp32 MFMA GEMMs): %.1f %% of kernel time.\n" % (100 * gemm / total))

# ---- HBM traffic (PMC passes) --------------------------------------------------------------------------------------------
import os
base = os.path.dirname(os.path.dirname(src))
traffic = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(base, "pmc_" + cname, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != cname:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0, int(r["Start_Timestamp"])])
        d[1] += float(r["Counter_Value"])
    seq = sorted(disp.values(), key=lambda d: d[2])
    marks = [i for i, d in enumerate(seq) if "arange" in d[0].lower()]
    if marks:
        seq = seq[marks[-1] + 1:]
    fam = collections.OrderedDict()
    for name, kib, _ in seq:
        n = name.replace("void ian::", "").replace("ian::", "")
        n = n[:n.index("(")] if "(" in n else n
        n = n[:n.index("<")] if "<" in n else n
        fam[n] = fam.get(n, 0.0) + kib * 1024.0
    traffic[cname] = fam
if traffic:
    fetch, write = traffic.get("FETCH_SIZE", {}), traffic.get("WRITE_SIZE", {})
    names = sorted(set(fetch) | set(write), key=lambda n: -(2 * fetch.get(n, 0) + write.get(n, 0)))
    rd, wr = 2 * sum(fetch.values()), sum(write.values())
    js = {"updates": 4, "hbm_read_bytes_per_update": rd / 4, "hbm_write_bytes_per_update": wr / 4, "hbm_bytes_per_update": (rd + wr) / 4,
          "note": "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, separate rocprofv3 --pmc passes of scripts/train_profile.py, "
                  "4 updates after the marker launch", "per_family_bytes_per_update": {n: (2 * fetch.get(n, 0) + write.get(n, 0)) / 4 for n in names}}
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from neural_photo_editor_amd import build as _b
        js["csrc_digest"] = _b._digest()
    except Exception:
        pass
    json.dump(js, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "a") as f:
        f.write("\n## HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; read side x2-corrected for gfx950)\n\n")
        f.write("Per update: **%.2f GB read + %.2f GB written = %.2f GB** (%.1f GB/s averaged over the %.1f ms an update takes: the step is "
                "nowhere near the HBM roof; it is bound by the fp32 matrix rate).\n\n" % (rd / 4e9, wr / 4e9, (rd + wr) / 4e9, (rd + wr) / 4e9 / (span / 4e3), span / 4))
        f.write("| kernel family | read GB / update | written GB / update |\n|---|---|---|\n")
        for n in names[:16]:
            f.write("| %s | %.3f | %.3f |\n" % (n, 2 * fetch.get(n, 0) / 4e9, write.get(n, 0) / 4e9))
print(open(out + ".md").read()[:6000])

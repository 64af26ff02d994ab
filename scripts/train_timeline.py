#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 kernel trace of scripts/train_profile.py: when is the matrix pipe without a GEMM?

usage: python scripts/train_timeline.py gpurun_out/<tag>/trace/trace_kernel_trace.csv [out.json]

For the steady-state updates (after the marker launch) it reports, per update kind and in total:
  * span, and the union of the intervals in which at least one MFMA GEMM kernel (tapgemm / tapwgrad / head6 / mdc_head_wgrad /
    conv1_mfma) is resident on ANY stream -- span minus that union is time in which only element-wise / reduction kernels (or
    nothing) ran: what more overlap could still hide;
  * the same union for two or more concurrent GEMM kernels (time in which MFMA work of two streams is conserved, not hidden);
  * which non-GEMM kernels run in the GEMM-free intervals (the passes worth fusing, moving or overlapping), by total exposed time.
"""
import collections, csv, json, sys

src = sys.argv[1]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = max(i for i, r in enumerate(rows) if "arange" in r["Kernel_Name"].lower())
rows = rows[mark + 1:]
GEMM = ("tapgemm_kernel", "tapwgrad", "head6_kernel", "mdc_head_wgrad", "conv1_mfma", "mdc_thin_tile", "tapgemm_bf16x3")


def name(r):
    return r["Kernel_Name"].replace("void ian::", "").replace("ian::", "").split("(")[0]


def is_gemm(r):
    n = name(r)
    return any(n.startswith(g) for g in GEMM)


t0 = int(rows[0]["Start_Timestamp"])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    ev.append((s, e, is_gemm(r), name(r), r.get("Stream_Id", "?")))
span = max(e for _, e, _, _, _ in ev)

# sweep: count of resident GEMM kernels and of any kernels over time
pts = []
for s, e, g, _, _ in ev:
    pts.append((s, 1, g))
    pts.append((e, -1, g))
pts.sort()
ng = na = 0
last = 0
t_g1 = t_g2 = t_any = 0
gemm_free = []   # intervals without a GEMM
cur_free = 0
for t, d, g in pts:
    dt = t - last
    if dt > 0:
        if ng >= 1: t_g1 += dt
        if ng >= 2: t_g2 += dt
        if na >= 1: t_any += dt
        if ng == 0: gemm_free.append((last, t))
    last = t
    na += d
    if g: ng += d

# exposed time of each non-GEMM kernel = its overlap with the GEMM-free intervals
import bisect
starts = [a for a, _ in gemm_free]
exposed = collections.Counter()
calls = collections.Counter()
dur = collections.Counter()
for s, e, g, n, sid in ev:
    if g:
        continue
    calls[n] += 1
    dur[n] += e - s
    i = max(0, bisect.bisect_right(starts, s) - 1)
    while i < len(gemm_free) and gemm_free[i][0] < e:
        a, b = gemm_free[i]
        ov = min(e, b) - max(s, a)
        if ov > 0:
            exposed[n] += ov
        i += 1
idle_nothing = span - t_any
out = {
    "span_ms": span / 1e6,
    "gemm_resident_ms": t_g1 / 1e6,
    "two_or_more_gemms_ms": t_g2 / 1e6,
    "gemm_free_ms": (span - t_g1) / 1e6,
    "nothing_resident_ms": idle_nothing / 1e6,
    "gemm_kernel_sum_ms": sum(e - s for s, e, g, _, _ in ev if g) / 1e6,
    "non_gemm_kernel_sum_ms": sum(e - s for s, e, g, _, _ in ev if not g) / 1e6,
    "exposed_non_gemm_ms": {k: v / 1e6 for k, v in exposed.most_common(25)},
    "non_gemm_calls": {k: calls[k] for k, _ in exposed.most_common(25)},
    "non_gemm_total_ms": {k: dur[k] / 1e6 for k, _ in exposed.most_common(25)},
}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)

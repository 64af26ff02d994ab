#!/usr/bin/env python
"""Turn the rocprofv3 CSVs written by scripts/profile_round.sh into the committed summary under profiles/.

usage: python scripts/summarize_profile.py gpurun_out/<tag> profiles/<name>   (writes <name>.md and <name>.json)

Only steady-state steps are summarised: the bench process also runs the autotuner (hundreds of candidate
launches), so the kernel trace is cut into steps at every launch of the step's first kernel
(conv1_nchw_kernel) and the last STEPS steps are kept.

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md:
  * FETCH_SIZE and WRITE_SIZE come from separate passes (3 + 2 TCC slots do not fit one pass), unit KiB;
  * on gfx950 FETCH_SIZE counts 128-byte requests of wide coalesced reads as 64 B -> the read side is doubled
    for the 16-B/lane streaming kernels (tapgemm, edge layers); WRITE_SIZE is uncalibrated and left as reported;
  * derived MfmaUtil is a gfx94x formula -> SQ_VALU_MFMA_BUSY_CYCLES (SIMD-cycles summed over the chip: exactly
    64 x #MFMA for v_mfma_f32_32x32x2_f32) is compared against the kernel's cycles x 1024 SIMDs instead, with the
    kernel's cycles = GRBM_GUI_ACTIVE / 8 (the counter is reported summed over the 8 XCDs: it reads ~8x the
    duration x clock of the same dispatch).
"""
from __future__ import annotations

import collections
import csv
import json
import os
import sys

STEPS = 20
FIRST_KERNEL = "conv1_"   # conv1_mfma_kernel or conv1_nchw_kernel: first launch of a reconstruction step


def short(name):
    n = name.replace("void ian::", "").replace("ian::", "")
    for cut in ("(float", "(ian::"):
        if cut in n:
            n = n[:n.index(cut)]
    return n


def steady(rows, key="Kernel_Name"):
    idx = [i for i, r in enumerate(rows) if FIRST_KERNEL in r[key]]
    if len(idx) < STEPS + 1:
        return rows, 1
    return rows[idx[-STEPS - 1]:idx[-1]], STEPS


def load_trace(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def load_pmc(path):
    """-> list of dispatches (ordered) each {name, grid, counters{}}"""
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = by.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": {},
                                                  "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {"source": src, "steps_summarised": STEPS}
    md = ["# rocprofv3 summary: %s" % os.path.basename(dst), "",
          "Command: `%s`" % open(os.path.join(src, "cmd.txt")).read().strip() if os.path.exists(os.path.join(src, "cmd.txt")) else "",
          ""]
    bench = None
    for f in ("bench.json", "trace_bench.json"):
        p = os.path.join(src, f)
        if os.path.exists(p):
            for line in open(p):
                if line.startswith("{"):
                    bench = json.loads(line)
            if bench:
                break
    if bench:
        out["bench"] = bench
        md += ["## bench.py line (un-profiled run in the same call)", "", "```json", json.dumps(bench), "```", ""]

    tr = load_trace(os.path.join(src, "trace", "trace_kernel_trace.csv"))
    st, nsteps = steady(tr)
    agg = collections.OrderedDict()
    for r in st:
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
        a = agg.setdefault(k, {"calls": 0, "ns": 0, "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"]})
        a["calls"] += 1
        a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    total_ns = sum(a["ns"] for a in agg.values())
    span_ns = int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])
    md += ["## Kernel trace, last %d steady-state steps (`rocprofv3 --kernel-trace --stats`)" % nsteps, "",
           "Sum of kernel durations per step: **%.1f us**; wall span per step (incl. launch gaps): %.1f us." % (
               total_ns / nsteps / 1e3, span_ns / nsteps / 1e3), "",
           "| kernel | workgroups | launches/step | avg us | % of kernel time | VGPR | SGPR | LDS B |", "|---|---|---|---|---|---|---|---|"]
    tg_ns = tg_calls = 0
    kern = []
    for (name, wgs), a in agg.items():
        md.append("| %s | %d | %.2f | %.1f | %.1f | %s | %s | %s |" % (name, wgs, a["calls"] / nsteps, a["ns"] / a["calls"] / 1e3,
                                                                      100.0 * a["ns"] / total_ns, a["vgpr"], a["sgpr"], a["lds"]))
        kern.append({"kernel": name, "workgroups": wgs, "launches_per_step": a["calls"] / nsteps, "avg_us": a["ns"] / a["calls"] / 1e3})
        if name.startswith("tapgemm_kernel"):
            tg_ns += a["ns"]
            tg_calls += a["calls"]
    out["kernels"] = kern
    out["tapgemm_avg_us"] = tg_ns / max(tg_calls, 1) / 1e3
    out["tapgemm_ns_per_step"] = tg_ns / nsteps
    md += ["", "tapgemm_kernel (all tile shapes): %.2f launches/step, average duration **%.1f us**, %.1f%% of kernel time."
           % (tg_calls / nsteps, out["tapgemm_avg_us"], 100.0 * tg_ns / total_ns), ""]
    if bench and bench.get("roofline"):
        md += ["bench.py's in-process HIP-event figure for the same kernel family: avg_launch_ms = %.4f (%.1f us)."
               % (bench["roofline"]["avg_launch_ms"], bench["roofline"]["avg_launch_ms"] * 1e3),
               "(the event pair also brackets the split-K reduce launch that belongs to the layer)", ""]

    # full --stats table as produced by rocprofv3 (includes autotune launches)
    sp = os.path.join(src, "trace", "trace_kernel_stats.csv")
    if os.path.exists(sp):
        md += ["## rocprofv3 --stats table (whole process, autotune candidates included)", "",
               "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(sp)):
            md.append("| %s | %s | %.2f | %.1f | %s |" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
        md.append("")

    # PMC passes
    pmc_dirs = sorted(d for d in os.listdir(src) if d.startswith("pmc_") and os.path.isdir(os.path.join(src, d)))
    per_kernel = collections.OrderedDict()
    for d in pmc_dirs:
        p = os.path.join(src, d, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        disp = load_pmc(p)
        st2, n2 = steady(disp)
        for x in st2:
            k = (short(x["Kernel_Name"]), x["grid"])
            e = per_kernel.setdefault(k, {"n": collections.Counter(), "c": collections.Counter()})
            for cn, v in x["c"].items():
                e["c"][cn] += v
                e["n"][cn] += 1
    if per_kernel:
        md += ["## PMC counters per launch (separate passes; steady-state steps only)", "",
               "FETCH is FETCH_SIZE KiB x 1024 x 2 (gfx950 128-B-request correction), WRITE is WRITE_SIZE KiB x 1024 (uncalibrated).",
               "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs).", "",
               "| kernel | grid threads | HBM read MB | HBM write MB | MFMA busy % | wave-cycles: active / wait-inst / wait-any % | LDS bank-conflict % | L2 hit % |",
               "|---|---|---|---|---|---|---|---|"]
        traffic = {}
        for (name, grid), e in per_kernel.items():
            c = {k: e["c"][k] / e["n"][k] for k in e["c"]}
            rd = c.get("FETCH_SIZE", float("nan")) * 1024 * 2
            wr = c.get("WRITE_SIZE", float("nan")) * 1024
            gui = c.get("GRBM_GUI_ACTIVE", 0)
            mf = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8.0 * 1024) if gui else float("nan")
            wc = c.get("SQ_WAVE_CYCLES", 0)
            frac = "%.0f / %.0f / %.0f" % (100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
                                          100 * c.get("SQ_WAIT_ANY", 0) / wc) if wc else "-"
            la = c.get("SQ_LDS_IDX_ACTIVE", 0)
            bc = "%.1f" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / la) if la else "-"
            hit = c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)
            l2 = "%.1f" % (100 * c.get("TCC_HIT_sum", 0) / hit) if hit else "-"
            md.append("| %s | %d | %.2f | %.2f | %.1f | %s | %s | %s |" % (name, grid, rd / 1e6, wr / 1e6, mf, frac, bc, l2))
            traffic["%s@%d" % (name, grid)] = {"read_bytes": rd, "write_bytes": wr, "mfma_busy_pct": mf, "raw": c}
        out["pmc"] = traffic
        # HBM bytes per tapgemm launch, LAUNCH-WEIGHTED: total bytes of every steady-state tapgemm dispatch / number of such dispatches
        # (until round 5 this was the unweighted mean over the (kernel name, grid) rows above, which counted a shape launched three
        # times per step once: 88.4 MB where the launches average 100.6 MB)
        rd_sum = wr_sum = 0.0
        rd_n = wr_n = 0
        for (name, grid), e in per_kernel.items():
            if not name.startswith("tapgemm_kernel"):
                continue
            rd_sum += e["c"].get("FETCH_SIZE", 0) * 1024 * 2
            rd_n += e["n"].get("FETCH_SIZE", 0)
            wr_sum += e["c"].get("WRITE_SIZE", 0) * 1024
            wr_n += e["n"].get("WRITE_SIZE", 0)
        if rd_n and wr_n:
            out["tapgemm_traffic_bytes_per_launch"] = rd_sum / rd_n + wr_sum / wr_n
            out["tapgemm_traffic_weighting"] = "launch-weighted (sum over dispatches / dispatches)"
            md += ["", "tapgemm_kernel HBM traffic per launch (launch-weighted mean over every steady-state dispatch, read x2-corrected + write): **%.1f MB**."
                   % (out["tapgemm_traffic_bytes_per_launch"] / 1e6), ""]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst + ".md", "w").write("\n".join(md) + "\n")
    # stamp the kernel sources the profile was taken on: bench.py quotes `tapgemm_traffic_bytes_per_launch` as this
    # build's roofline.traffic only when the digest matches (otherwise it reports the profile as stale)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from neural_photo_editor_amd import build as _b
    out["csrc_digest"] = _b._digest("inference")   # the sources a reconstruction kernel can depend on (bench.py: pmc_traffic)
    json.dump(out, open(dst + ".json", "w"), indent=1)
    print("wrote", dst + ".md", dst + ".json")


if __name__ == "__main__":
    main()

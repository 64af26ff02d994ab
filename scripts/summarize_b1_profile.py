#!/usr/bin/env python
"""rocprofv3 CSVs of scripts/profile_b1.sh -> profiles/<name>.{md,json}: kernels and HBM bytes per brush event and per batch-1
reconstruction (between the marker launches of scripts/b1_chain_profile.py).  FETCH_SIZE x2-corrected for gfx950 and WRITE_SIZE as
reported (MI355X_MICROARCH.md, HBM section), separate PMC passes.
usage: python scripts/summarize_b1_profile.py gpurun_out/<tag> profiles/<name>"""
import collections, csv, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
N = 100


def short(n):
    n = n.replace("void ian::", "").replace("ian::", "")
    for cut in ("(float", "(ian::", "(int", "(unsigned"):
        if cut in n:
            n = n[:n.index(cut)]
    return n


def phases(rows, name_key, order_key):
    rows = sorted(rows, key=order_key)
    marks = [i for i, r in enumerate(rows) if "arange" in r[name_key].lower() or "elementwise_kernel" in r[name_key].lower() and False]
    assert len(marks) >= 4, "markers not found (%d)" % len(marks)
    m = marks[-4:]
    return rows[m[0] + 1:m[1]], rows[m[2] + 1:m[3]]


trace = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
ev, rc = phases(trace, "Kernel_Name", lambda r: int(r["Start_Timestamp"]))
out = {"reps": N, "source": src}
md = ["# rocprofv3 summary: %s" % os.path.basename(dst), "", "Command: `%s`" % open(os.path.join(src, "cmd.txt")).read().strip(), ""]
for label, rows in (("brush event (ian_brush_step, one call, hipGraph replay)", ev), ("batch-1 reconstruction (ian_reconstruct, device buffers)", rc)):
    agg = collections.OrderedDict()
    for r in rows:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(a[1] for a in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    key = "edit" if label.startswith("brush") else "b1_recon"
    out[key] = {"launches_per_unit": len(rows) / N, "kernel_us_per_unit": tot / N / 1e3, "span_us_per_unit": span / N / 1e3}
    md += ["## %s: %.1f launches, %.1f us of kernels, %.1f us first-to-last span per unit (%d units)" % (label, len(rows) / N, tot / N / 1e3, span / N / 1e3, N), "",
           "| kernel | launches/unit | avg us | % of kernel time |", "|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        md.append("| %s | %.2f | %.2f | %.1f |" % (k, a[0] / N, a[1] / a[0] / 1e3, 100.0 * a[1] / tot))
    md.append("")
for cname, mult in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    p = os.path.join(src, "pmc_" + cname, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    by = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        d = by.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "v": 0.0})
        if r["Counter_Name"] == cname:
            d["v"] += float(r["Counter_Value"])
    rows = [dict(v, id=k) for k, v in by.items()]
    e2, r2 = phases(rows, "Kernel_Name", lambda r: r["id"])
    for key, rr in (("edit", e2), ("b1_recon", r2)):
        out[key][cname.lower() + "_bytes_per_unit"] = sum(x["v"] for x in rr) * mult / N
        fam = collections.Counter()
        for x in rr:
            fam[short(x["Kernel_Name"])] += x["v"] * mult / N
        out[key][cname.lower() + "_by_kernel"] = dict(fam)
for key, alg in (("edit", 232e6), ("b1_recon", 214e6)):
    o = out[key]
    if "fetch_size_bytes_per_unit" in o and "write_size_bytes_per_unit" in o:
        o["hbm_bytes_per_unit"] = o["fetch_size_bytes_per_unit"] + o["write_size_bytes_per_unit"]
        md += ["**%s: %.1f MB read (FETCH_SIZE x2-corrected) + %.1f MB written = %.1f MB of HBM / fabric traffic per unit** (SURVEY 8(d) algorithmic basis: %.0f MB).  "
               "FETCH_SIZE counts L2 misses to the fabric, Infinity-Cache hits included: weights that stay in the 256 MiB Infinity Cache between events still count."
               % (key, o["fetch_size_bytes_per_unit"] / 1e6, o["write_size_bytes_per_unit"] / 1e6, o["hbm_bytes_per_unit"] / 1e6, alg / 1e6), ""]
        md += ["| kernel | read MB / unit | written MB / unit |", "|---|---|---|"]
        for k in sorted(o["fetch_size_by_kernel"], key=lambda k: -o["fetch_size_by_kernel"][k]):
            md.append("| %s | %.2f | %.2f |" % (k, o["fetch_size_by_kernel"][k] / 1e6, o["write_size_by_kernel"].get(k, 0.0) / 1e6))
        md.append("")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_photo_editor_amd import build as _b
out["csrc_digest"] = _b._digest("inference")
open(dst + ".md", "w").write("\n".join(md) + "\n")
json.dump(out, open(dst + ".json", "w"), indent=1)
print("wrote", dst + ".md", dst + ".json")

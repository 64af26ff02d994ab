"""Per-layer kernel durations and PMC counters of scripts/exp/wgrad_probe.py runs.
usage: python scripts/exp/wgrad_probe_summary.py <dir with trace/ and pmc_*/> <rates.json> [out.json]"""
import collections, csv, json, os, sys
src = sys.argv[1]
rates = json.load(open(sys.argv[2]))
names = [r["layer"] for r in rates["layers"]]


def short(n):
    return n.replace("void ian::", "").replace("ian::", "").split("(")[0]


def segments(rows, key_start):
    """split the dispatch list at the marker launches; returns list (per layer) of rows"""
    segs, cur, seen = [], None, 0
    for r in rows:
        if "arange" in r["Kernel_Name"].lower():
            seen += 1
            cur = []
            segs.append(cur)
            continue
        if cur is not None:
            cur.append(r)
    return segs[-len(names):]


out = {}
tr = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
for name, seg in zip(names, segments(tr, "Start_Timestamp")):
    d = collections.OrderedDict()
    for r in seg:
        k = short(r["Kernel_Name"])
        a = d.setdefault(k, {"calls": 0, "us": 0.0, "wgs": int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), "vgpr": r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), "lds": r.get("LDS_Block_Size")})
        a["calls"] += 1
        a["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for a in d.values():
        a["avg_us"] = a.pop("us") / a["calls"]
    out[name] = {"kernels": d}
for pd in sorted(os.listdir(src)):
    p = os.path.join(src, pd, "pmc_counter_collection.csv")
    if not pd.startswith("pmc_") or not os.path.exists(p):
        continue
    by = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        d = by.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [by[k] for k in sorted(by)]
    for name, seg in zip(names, segments(rows, None)):
        acc = {}
        for r in seg:
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, collections.Counter())
            a["_n"] += 1
            for c, v in r["c"].items():
                a[c] += v
        for k, a in acc.items():
            n_ = a.pop("_n")
            out[name]["kernels"].setdefault(k, {}).setdefault("pmc", {}).update({c: v / n_ for c, v in a.items()})
for r in rates["layers"]:
    o = out[r["layer"]]
    o["gflop"] = r["gflop"]
    o["event_us"] = r["backward_weight_us"]
    print("== %-12s %6.1f GFLOP  %8.1f us (events, kernel + reduce)  %.3f of peak" % (r["layer"], r["gflop"], r["backward_weight_us"], r["frac_of_peak"]))
    for k, a in o["kernels"].items():
        if "avg_us" not in a:
            continue
        line = "   %-44s wgs %6d  %8.1f us" % (k[:44], a["wgs"], a["avg_us"])
        c = a.get("pmc", {})
        if k.startswith("tapwgrad"):
            line += "  %.3f of peak" % (r["gflop"] * 1e9 / (a["avg_us"] * 1e-6) / 157.3e12)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            line += "  mfma-busy %.1f%%" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024))
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            line += "  act/wait-inst/wait-any %.0f/%.0f/%.0f%%" % (100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / w, 100 * c.get("SQ_WAIT_ANY", 0) / w)
        if "FETCH_SIZE" in c:
            line += "  rd %.1f MB" % (c["FETCH_SIZE"] * 1024 * 2 / 1e6)
        if "WRITE_SIZE" in c:
            line += "  wr %.1f MB" % (c["WRITE_SIZE"] * 1024 / 1e6)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            line += "  lds-conf %.1f%%" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"])
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            line += "  L2 hit %.1f%%" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))
        print(line)
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv files (one counter per pass) into the small
per-dispatch CSVs and the summary JSON kept under profiles/.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix> <timed_launches> <workload note>"""
import csv
import json
import sys
from collections import OrderedDict


def condense(path, out_csv):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if not name.startswith(("gm::", "void gm::")):
                continue
            rows.append((name, int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"]),
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"],
                         r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]))
    # one row per dispatch: a counter is reported once per XCD/instance, sum them
    agg = OrderedDict()
    for name, did, cn, val, dur, vg, ag, sg, lds, scr in rows:
        k = (name, did, cn)
        if k not in agg:
            agg[k] = [0.0, dur, vg, ag, sg, lds, scr]
        agg[k][0] += val
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatch_Id", "Counter_Name", "Counter_Value_KiB", "Duration_ns", "VGPR", "AccVGPR", "SGPR", "LDS", "Scratch"])
        for (name, did, cn), v in agg.items():
            w.writerow([name, did, cn, f"{v[0]:.6f}", v[1], v[2], v[3], v[4], v[5], v[6]])
    return agg


def short(name):
    name = name.replace("void ", "").split("(")[0]
    # the sweep kernel's instantiations (bytes per thread, missing-genotype mode, with / without the continuation) are one
    # family: a bench run uses the continuation kernel in the dense early sweeps and the plain one afterwards
    return "gm::k_sweep" if name.startswith("gm::k_sweep<") else name


def main():
    fetch, write, prefix, timed, note = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    af = condense(fetch, prefix + "_pmc_fetch.csv")
    aw = condense(write, prefix + "_pmc_write.csv")
    kernels = OrderedDict()
    for agg in (af, aw):
        for (name, did, cn), v in sorted(agg.items(), key=lambda kv: kv[0][1]):          # dispatch order
            kernels.setdefault(short(name), OrderedDict()).setdefault(cn, []).append(v[0] * 1024.0)
    sweep = next(k for k in kernels if k.startswith("gm::k_sweep"))
    fs = kernels[sweep]["FETCH_SIZE"][-timed:]
    ws = kernels[sweep]["WRITE_SIZE"][-timed:]
    out = {
        "workload": note,
        "unit_note": "rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; values below are bytes",
        "kernels": kernels,
        "k_sweep_per_launch": {
            "FETCH_SIZE_bytes_raw": sum(fs) / len(fs), "WRITE_SIZE_bytes": sum(ws) / len(ws),
            "note": "timed launches only (the last %d of each pass). FETCH_SIZE is raw; on gfx950 it reports half the bytes of "
                    "16-B/lane streams (k_marker_stats reads exactly 125.0e9 B and reports 62.6e9), and k_sweep's column "
                    "loads are 16 B/lane since kernel v3, so the x2 reading is the comparable one." % timed,
            "FETCH_SIZE_bytes_x2": 2.0 * sum(fs) / len(fs),
        },
        "traffic_bytes_per_launch_k_sweep": 2.0 * sum(fs) / len(fs) + sum(ws) / len(ws),
    }
    if "gm::k_marker_stats" in kernels:
        m = kernels["gm::k_marker_stats"]["FETCH_SIZE"][0]
        out["k_marker_stats"] = {"FETCH_SIZE_bytes_raw": m, "FETCH_SIZE_bytes_x2": 2 * m}
    with open(prefix + "_pmc_summary.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["k_sweep_per_launch"], indent=1), out["traffic_bytes_per_launch_k_sweep"])


if __name__ == "__main__":
    main()

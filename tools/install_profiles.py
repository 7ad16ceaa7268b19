#!/usr/bin/env python3
"""Copy one run of tools/profile_round.sh (and, when present, profile_sq.sh / shard_sized.sh / the ld bench) from gpurun_out/<tag> into
profiles/<round>_* and print the numbers the docs quote.   tools/install_profiles.py r04e r04 [shard_tag]"""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def last_json(p):
    return json.loads(Path(p).read_text().strip().splitlines()[-1])


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles"
    for n in ("c2", "c3", "c3_20steps", "c4", "c5", "c6", "ld", "under_rocprof"):
        f = src / f"bench_{n}.json"
        if f.exists():
            shutil.copy(f, dst / f"{rnd}_bench_{n}.json")
            d = last_json(f)
            r = d["roofline"]
            print(n, round(d["value"] / 1e6, 3), "M/s", round(d["ms_per_step"], 2), "ms/step kernel", round(r["kernel_ms_avg"], 2), "frac", round(r["frac"], 4),
                  "warm-up", [round(x, 1) for x in r["kernel_ms_warmup_launches"]], "timed", [round(x, 1) for x in r["kernel_ms_per_launch"]][:8],
                  "rounds", d["sweep"]["sync_rounds_per_sweep"][-1], "cpu", d.get("cpu_baseline", {}).get("value"))
    for a, b in ((f"{tag}_kernel_stats.csv", f"{rnd}_kernel_stats.csv"), (f"{tag}_pmc_fetch.csv", f"{rnd}_pmc_fetch.csv"), (f"{tag}_pmc_write.csv", f"{rnd}_pmc_write.csv"),
                 (f"{tag}_pmc_summary.json", f"{rnd}_pmc_summary.json"), (f"{tag}_phase_stamps.txt", f"{rnd}_phase_stamps.txt")):
        if (src / a).exists():
            shutil.copy(src / a, dst / b)
    if (dst / f"{rnd}_pmc_summary.json").exists():
        d = json.loads((dst / f"{rnd}_pmc_summary.json").read_text())
        print("traffic per stationary launch", d["traffic_bytes_per_launch_k_sweep"] / 1e9, "GB =", d["traffic_bytes_per_launch_k_sweep"] / 125e9, "x")
    if (dst / f"{rnd}_kernel_stats.csv").exists():
        for r in list(csv.DictReader(open(dst / f"{rnd}_kernel_stats.csv")))[:2]:
            print(r["Name"][:48], r["Calls"], "avg", float(r["AverageNs"]) / 1e6, "min", float(r["MinNs"]) / 1e6, "max", float(r["MaxNs"]) / 1e6)
    summ_p = dst / f"{rnd}_pmc_sq_summary.json"
    summ = json.loads(summ_p.read_text()) if summ_p.exists() else {}
    for w in ("c3", "c5"):
        f = src / f"{tag}_pmc_sq_{w}.json"
        if f.exists():
            d = json.loads(f.read_text())
            shutil.copy(f, dst / f"{rnd}_pmc_sq_{w}.json")
            k = [x for x in d["kernels"] if "k_sweep" in x][0]
            kd = dict(d["kernels"][k]); kd["kernel"] = k
            summ[w] = kd
            print("SQ", w, {a: round(b, 3) for a, b in kd["ratios"].items()}, kd.get("duration_ns_under_pmc"))
    if summ:
        summ_p.write_text(json.dumps(summ, indent=1))
    if len(sys.argv) > 3:
        f = ROOT / "gpurun_out" / sys.argv[3] / "shard_sized.json"
        if f.exists():
            s = json.loads(f.read_text())
            (dst / f"{rnd}_shard_sized.json").write_text(json.dumps(s, indent=1))
            for e in s["shard_sized"]:
                print("shard", e["markers"], round(e["ms_per_step"], 2), "ms/step kernel", round(e["kernel_ms_avg"], 2), "host", round(e["host_ms_per_step"], 2), round(e["value"] / 1e6, 2), "M/s")
            print({k: v for k, v in s.items() if k != "shard_sized"})


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes of SQ counters (counter_collection.csv, one pass per <= 8 counters) into one
JSON: per kernel, per counter, the sum over XCDs / SEs of the LAST `timed` dispatches, averaged per launch, plus
the ratios the bound argument uses.
usage: pmc_sq_summary.py <out.json> <timed_launches> <workload> <pass1.csv> [<pass2.csv> ...]"""
import csv
import json
import sys
from collections import OrderedDict


def read_pass(path):
    per = OrderedDict()       # (kernel, dispatch) -> {counter: sum}
    dur = {}
    for r in csv.DictReader(open(path, newline="")):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not name.startswith("gm::"):
            continue
        if name.startswith("gm::k_sweep<"):              # one family: the early dense sweeps run the continuation instantiation
            name = "gm::k_sweep"
        k = (name, int(r["Dispatch_Id"]))
        per.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per, dur


def main():
    out_path, timed, wl, passes = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4:]
    kernels = OrderedDict()
    for p in passes:
        per, dur = read_pass(p)
        by_kernel = OrderedDict()
        for (name, did), cs in per.items():
            by_kernel.setdefault(name, []).append((did, cs, dur[(name, did)]))
        for name, lst in by_kernel.items():
            lst.sort()
            last = lst[-timed:] if name.startswith("gm::k_sweep") else lst
            k = kernels.setdefault(name, {"launches_averaged": len(last), "counters": OrderedDict(), "duration_ns_under_pmc": []})
            k["duration_ns_under_pmc"].append(sum(d for _, _, d in last) / len(last))
            for cn in last[0][1]:
                k["counters"][cn] = sum(cs[cn] for _, cs, _ in last) / len(last)
    for name, k in kernels.items():
        c = k["counters"]
        r = OrderedDict()

        def ratio(key, a, b, scale=1.0):
            if a in c and b in c and c[b] > 0:
                r[key] = scale * c[a] / c[b]
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; BUSY counts per SQ (guide, PMC section)
        ratio("wait_any_over_wave_cycles", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
        ratio("wait_inst_any_over_wave_cycles", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
        ratio("active_inst_any_over_wave_cycles", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
        ratio("active_inst_valu_over_wave_cycles", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")
        ratio("active_inst_lds_over_wave_cycles", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES")
        ratio("active_inst_vmem_over_wave_cycles", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES")
        ratio("active_inst_scalar_over_wave_cycles", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES")
        ratio("wait_inst_lds_over_wave_cycles", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")
        ratio("lds_bank_conflict_over_lds_idx_active", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
        ratio("mfma_busy_cycles_over_busy_cu_cycles", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES")
        ratio("mfma_busy_cycles_over_sq_cycles", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_CYCLES")
        ratio("mfma_insts_per_valu_inst", "SQ_INSTS_MFMA", "SQ_INSTS_VALU")
        k["ratios"] = r
    json.dump({"workload": wl, "note": "per launch: counter values summed over all XCDs / shader engines, averaged over the timed "
               "launches (the last %d k_sweep dispatches of each pass); one rocprofv3 --pmc pass per group of 8 SQ counters" % timed,
               "kernels": kernels}, open(out_path, "w"), indent=1)
    sw = next((k for k in kernels if k.startswith("gm::k_sweep")), None)
    if sw:
        print(sw, json.dumps(kernels[sw]["ratios"], indent=1))


if __name__ == "__main__":
    main()

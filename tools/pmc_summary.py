#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (FETCH_SIZE / WRITE_SIZE / SQ_*) per kernel.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq profiles/r01_pmc

HBM-side bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of a wide coalesced stream, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import json
import sys

fetch_dir, write_dir, sq_dir, out = sys.argv[1:5]
# provenance (bench.py prints it beside roofline.traffic): precision, the git HEAD the library was built from (passed in by
# the caller — the GPU box has no .git) and the profiled command
meta = {"precision": sys.argv[5] if len(sys.argv) > 5 else "bf16x6", "git_head": sys.argv[6] if len(sys.argv) > 6 else None,
        "command": sys.argv[7] if len(sys.argv) > 7 else None,
        "method": "rocprofv3 --kernel-trace --pmc, one pass per counter group; FETCH_SIZE x2 (gfx950), KiB -> bytes"}


def demangle(name):
    """rocprofv3 leaves kernels whose template arguments are _Float16 / __bf16 mangled (and the image has no llvm-cxxfilt):
    spell the few such symbols of this library out — _ZN3dsd3g1613gemm16_kernelIDF16_Li2ELi0EEEvNS0_4G16PE ->
    dsd::g16::gemm16_kernel<_Float16, 2, 0>."""
    import re
    if not name.startswith("_ZN"):
        return name
    i, parts = 3, []
    while i < len(name) and name[i].isdigit():
        j = i
        while name[j].isdigit():
            j += 1
        n = int(name[i:j])
        parts.append(name[j:j + n])
        i = j + n
    m = re.match(r"I(DF16_|DF16b)((?:L[ib]\d+E)*)E", name[i:])
    if not parts or not m:
        return name
    args = [v if t == "i" else ("true" if v != "0" else "false") for t, v in re.findall(r"L([ib])(\d+)E", m.group(2))]
    return "::".join(parts) + "<" + ", ".join(["_Float16" if m.group(1) == "DF16_" else "__bf16"] + args) + ">"


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def durations(path):
    d = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d


fe, wr, sq = (load(p + "/p_counter_collection.csv") for p in (fetch_dir, write_dir, sq_dir))
du = durations(sq_dir + "/p_kernel_trace.csv")
res = {}
for k in sorted(fe, key=lambda k: -sum(du.get(k, [0]))):
    n = len(fe[k]["FETCH_SIZE"])
    if n == 0 or ("dsd::" not in k and not k.startswith("_ZN3dsd")):
        continue
    rd = 2.0 * sum(fe[k]["FETCH_SIZE"]) * 1024.0
    wb = sum(wr[k]["WRITE_SIZE"]) * 1024.0
    s = {c: sum(v) for c, v in sq[k].items()}
    t_us = sum(du[k])
    e = {"launches": n, "total_us_under_pmc": round(t_us, 1),
         "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wb / n,
         "hbm_bytes_per_launch": (rd + wb) / n,
         "hbm_GBps": round((rd + wb) / (t_us * 1e-6) / 1e9, 1) if t_us else None}
    if s.get("GRBM_GUI_ACTIVE") and t_us:
        clk = s["GRBM_GUI_ACTIVE"] / 8.0 / (t_us * 1e-6)
        e["effective_clock_GHz"] = round(clk / 1e9, 3)
        if s.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["mfma_pipe_util"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4 * s["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if s.get("SQ_WAVE_CYCLES"):
        e["wave_time_split"] = {"wait_any(waitcnt/barrier)": round(s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"], 3),
                                "wait_inst_any(issue stall)": round(s.get("SQ_WAIT_INST_ANY", 0) / s["SQ_WAVE_CYCLES"], 3),
                                "active": round(s.get("SQ_ACTIVE_INST_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)}
    if s.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(s.get("SQ_LDS_BANK_CONFLICT", 0) / s["SQ_LDS_IDX_ACTIVE"], 4)
    res[demangle(k).replace("void ", "")[:110]] = e   # (long enough for every template argument of the convolution kernels)
res["_meta"] = meta
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])

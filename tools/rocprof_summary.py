#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into a small text/JSON kernel summary.

    python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats
"""
import json
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]


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


c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
res = []
lines = [f"{'kernel':<96} {'calls':>7} {'total_ms':>12} {'avg_us':>12} {'min_us':>10} {'max_us':>12} {'pct':>7}"]
for name, n, s, a, mn, mx in rows:
    name = demangle(name)
    short = name if len(name) <= 96 else name[:93] + "..."
    lines.append(f"{short:<96} {n:>7} {s / 1e6:>12.3f} {a / 1e3:>12.2f} {mn / 1e3:>10.2f} {mx / 1e3:>12.2f} {100 * s / tot:>7.2f}")
    res.append({"kernel": name, "calls": n, "total_ms": s / 1e6, "avg_us": a / 1e3, "min_us": mn / 1e3, "max_us": mx / 1e3,
                "pct": 100 * s / tot})
try:
    meta = c.execute("select vgpr_count, accum_vgpr_count, sgpr_count, lds_size, name from kernels group by name").fetchall()
    lines.append("")
    lines.append("registers / LDS per kernel (arch_vgpr, accum_vgpr, sgpr, lds_bytes):")
    for v, a, s, l, n in meta:
        n = demangle(n)
        lines.append(f"  {n[:96]:<96} {v} {a} {s} {l}")
except Exception as e:  # noqa
    pass
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines[:14]))

#!/usr/bin/env python3
"""VALU utilisation per kernel from a rocprofv3 --pmc SQ_* pass (+ --kernel-trace for the durations).

usage: summarize_sq.py <counter_collection.csv> <kernel_trace.csv> <out.md>
SQ_ACTIVE_INST_VALU is counted in quad-cycles (MI355X_MICROARCH.md): VALU busy = 4 * SQ_ACTIVE_INST_VALU / (duration * 2.4 GHz * 1024 SIMDs).
"""
import collections
import csv
import re
import sys


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def main():
    cc, kt, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(kt)):
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cnt[k] += 1
    rows = []
    for k, d in agg.items():
        if k.startswith("at::") or "elementwise" in k or k.startswith("__amd") or dur[k] < 0.5:
            continue
        waves = d.get("SQ_WAVES", 0)
        busy = 4 * d.get("SQ_ACTIVE_INST_VALU", 0) / (dur[k] * 1e-3 * 2.4e9 * 1024)
        # MFMA utilisation: cycles with the matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES, counted in cycles) over the kernel's
        # cycles on all SIMDs; 0 for a kernel that issues no MFMA instruction
        mfma = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur[k] * 1e-3 * 2.4e9 * 1024)
        rows.append((dur[k], k, cnt[k], waves, d.get("SQ_INSTS_VALU", 0) / max(waves, 1), d.get("SQ_INSTS_LDS", 0) / max(waves, 1),
                     d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), busy, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mfma))
    rows.sort(reverse=True)
    with open(out, "w") as fo:
        fo.write("| kernel | launches | ms (profiled) | waves | VALU instr / wave | LDS instr / wave | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | VALU busy | SQ_VALU_MFMA_BUSY_CYCLES | MFMA util |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for ms, k, n, waves, vi, li, wait, busy, mraw, mfma in rows:
            fo.write(f"| `{k}` | {n} | {ms:.2f} | {waves:.3e} | {vi:.0f} | {li:.0f} | {wait:.2f} | {busy:.2f} | {mraw:.0f} | {mfma:.3f} |\n")


if __name__ == "__main__":
    main()

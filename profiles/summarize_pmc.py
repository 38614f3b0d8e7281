#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs into the per-kernel HBM-traffic summary bench.py reads.

usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <H> <W> <out_prefix>

FETCH_SIZE / WRITE_SIZE are reported in KB (1024 B) per dispatch; on gfx950 FETCH_SIZE counts half of the
bytes of wide coalesced reads, so it is doubled (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).
Writes <out_prefix>.json ({"kernels": {name: {launches, read_B_per_px, write_B_per_px}}}) and <out_prefix>.md.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return per


def main():
    fetch_csv, write_csv, H, W, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    px = H * W
    fetch, write = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0]))):
        if k.startswith("at::") or "elementwise" in k or "Cijk" in k or k.startswith("__amd"):
            continue  # torch's own kernels (synthetic data generation)
        f, w = fetch.get(k, []), write.get(k, [])
        # a bench process also runs some kernels on a smaller raster (config 5 fits its forest on a 2048-row strip):
        # keep the launches on the full raster, i.e. those within 2x of the largest
        f = [v for v in f if v >= 0.5 * max(f)] if f and max(f) > 0 else f
        w = [v for v in w if v >= 0.5 * max(w)] if w and max(w) > 0 else w
        n = max(len(f), len(w))
        rd = 2.0 * 1024.0 * sum(f) / max(len(f), 1) / px
        wr = 1024.0 * sum(w) / max(len(w), 1) / px
        kernels[k] = {"launches": n, "fetch_kb_per_launch_raw": sum(f) / max(len(f), 1), "read_B_per_px": round(rd, 3),
                      "write_B_per_px": round(wr, 3)}
    json.dump({"H": H, "W": W, "note": "FETCH_SIZE doubled (gfx950), KB = 1024 B", "kernels": kernels}, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as fo:
        fo.write(f"# HBM traffic per kernel from PMC counters ({H}x{W}, 1 x MI355X)\n\n"
                 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0. Bytes per pixel per launch; "
                 "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported.\n\n"
                 "| kernel | launches | FETCH_SIZE KB/launch (raw) | read B/px (x2) | write B/px | total B/px |\n|---|---|---|---|---|---|\n")
        for k, v in kernels.items():
            fo.write(f"| `{k}` | {v['launches']} | {v['fetch_kb_per_launch_raw']:.0f} | {v['read_B_per_px']:.2f} | {v['write_B_per_px']:.2f} | "
                     f"{v['read_B_per_px'] + v['write_B_per_px']:.2f} |\n")


if __name__ == "__main__":
    main()

"""Every kernel of the last config-3 step of a rocprofv3 --kernel-trace CSV, in start order: duration and the idle gap in front of it."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("void k1_hist") and (i == 0 or not names[i - 1].startswith("void k1_hist"))]
seg = rows[starts[-1]:]
prev = None
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    print("%9.1f  %-46s %8.1f us  gap %7.1f" % ((s - t0) / 1e3, n, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = max(e, prev or 0)

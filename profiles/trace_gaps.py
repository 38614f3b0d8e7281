"""Timeline of one config-3 step from a rocprofv3 --kernel-trace CSV: busy time, idle gaps, and who sits between the sweeps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last k1_hist burst onwards
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("void k1_hist") and (i == 0 or not names[i - 1].startswith("void k1_hist"))]
i0 = starts[-1]
seg = rows[i0:]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("step span %.3f ms, kernel busy %.3f ms, idle %.3f ms, kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(seg)))
small = collections.defaultdict(lambda: [0, 0.0])
gaps = collections.defaultdict(lambda: [0, 0.0])
prev_end = None
for r in seg:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    small[n][0] += 1; small[n][1] += d
    if prev_end is not None:
        g = (int(r["Start_Timestamp"]) - prev_end) / 1e3
        gaps[n][0] += 1; gaps[n][1] += max(g, 0)
    prev_end = int(r["End_Timestamp"])
print("%-42s %5s %10s %12s" % ("kernel", "n", "busy us", "gap-before us"))
for n, (c, d) in sorted(small.items(), key=lambda kv: -kv[1][1]):
    print("%-42s %5d %10.1f %12.1f" % (n, c, d, gaps[n][1]))

"""Per-(kernel, grid) time table from a rocprofv3 kernel trace csv (tuning aid).  usage: trace_shapes.py trace.csv [div]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    key = (r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("total ms: %.2f" % (tot / 1e6 / div))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-62s grid=%-18s n=%-4d avg_us=%-8.1f ms=%.2f" % (k[0], "x".join(k[1:]), v[0] / div, v[1] / v[0] / 1e3, v[1] / 1e6 / div))

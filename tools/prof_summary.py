"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): count, total, average."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r['Kernel_Name'].split('(')[0][:64]
    k = (name, r.get('Grid_Size_X'), r.get('Grid_Size_Y'))
    agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print("%-66s %10s %4s %7s %12s %10s %6s" % ("kernel", "grid_x", "gy", "calls", "total_us", "avg_us", "%"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-66s %10s %4s %7d %12.1f %10.2f %6.1f" % (k[0], k[1], k[2], v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
print("total kernel time us: %.1f" % tot)

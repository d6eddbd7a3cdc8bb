"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py (one optimiser step between two Adam launches)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
ends = idx[1::2]
a, b = ends[-4], ends[-3]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
print("kernels in step:", len(step), "span ms", (t1 - t0) / 1e6)
agg = collections.OrderedDict(); busy = 0
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:64]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); busy += d
    c = agg.setdefault(n, [0, 0]); c[0] += 1; c[1] += d
print("busy ms", busy / 1e6)
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-66s %5d  %8.1f us total  %7.2f us avg" % (n, c, d / 1e3, d / 1e3 / c))

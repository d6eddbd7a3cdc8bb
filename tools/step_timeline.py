"""One optimiser step of a rocprofv3 kernel trace of bench.py as a timeline: start offset, duration, gap to the previous
kernel's end, grid and workgroup size per launch; then totals per kernel name (busy time, gaps attributed to the
launch that follows them).  usage: step_timeline.py <trace dir> [--full]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
# one adam_kernel launch per step (or two in older builds): take the span between the last two "step ends"
per = 2 if len(idx) >= 4 and idx[1] - idx[0] < 5 else 1
ends = idx[per - 1::per]
# the cleanest of the last steps (a profiled run now and then has a step with a stall in it): smallest span
best = None
for j in range(max(1, len(ends) - 20), len(ends) - 1):
    cand = rows[ends[j - 1] + 1:ends[j] + 1]
    span = max(int(r['End_Timestamp']) for r in cand) - int(cand[0]['Start_Timestamp'])
    if best is None or span < best[0]:
        best = (span, cand)
step = best[1]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
agg = collections.OrderedDict()
busy = gaps = 0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:56]
    grid = "%sx%sx%s" % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
    wg = r.get('Workgroup_Size_X', '?')
    gap = s - prev_end
    if '--full' in sys.argv:
        print("%9.2f us  dur %8.2f  gap %6.2f  %-56s grid %-16s wg %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n, grid, wg))
    c = agg.setdefault(n, [0, 0, 0])
    c[0] += 1; c[1] += e - s; c[2] += max(gap, 0)
    busy += e - s; gaps += max(gap, 0)
    prev_end = max(prev_end, e)
print("kernels in step: %d  span %.3f ms  busy %.3f ms  gaps %.3f ms" % (len(step), (prev_end - t0) / 1e6, busy / 1e6, gaps / 1e6))
for n, (c, d, g) in sorted(agg.items(), key=lambda x: -(x[1][1] + x[1][2])):
    print("%-56s %5d  busy %8.1f us (%6.2f avg)  gaps %7.1f us" % (n, c, d / 1e3, d / 1e3 / c, g / 1e3))

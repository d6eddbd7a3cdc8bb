"""Timeline of one optimiser step from a rocprofv3 kernel trace, one column per HSA queue (the overlapped step runs on the
caller's stream, the chain stream and the side stream).  usage: step_timeline_mq.py <trace dir> [--full]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
queues = sorted({r['Queue_Id'] for r in step}, key=lambda q: -sum(1 for r in step if r['Queue_Id'] == q))
busy = collections.Counter(); cnt = collections.Counter()
last_end = {}
for r in step:
    s, e, q = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    busy[q] += e - s; cnt[q] += 1
    if '--full' in sys.argv:
        col = queues.index(q)
        gap = (s - last_end.get(q, s)) / 1e3
        print("%9.2f us  %s dur %7.2f gap %7.2f  q%-2d %s%s" % ((s - t0) / 1e3, " " * 0, (e - s) / 1e3, gap, col, "    " * col, n))
    last_end[q] = e
end = max(int(r['End_Timestamp']) for r in step)
print("step span %.3f ms, %d kernels" % ((end - t0) / 1e6, len(step)))
for i, q in enumerate(queues):
    print("  q%d (id %s): %4d kernels, busy %.3f ms" % (i, q, cnt[q], busy[q] / 1e6))
# concurrency: time with >= 2 queues busy
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
lvl = 0; prev = ev[0][0]; t_by = collections.Counter()
for t, d in ev:
    t_by[min(lvl, 3)] += t - prev; prev = t; lvl += d
print("  time with 0/1/2/3+ kernels in flight: %s ms" % ", ".join("%.3f" % (t_by[i] / 1e6) for i in range(4)))

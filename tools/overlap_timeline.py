"""Timeline of one optimiser step from a rocprofv3 kernel trace: per 250-us window, which kernels ran and how busy
each stream (queue) was -- used to see what a side-stream overlap actually does to the recurrence."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
ends = idx[1::2]
a, b = ends[-4], ends[-3]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print("kernels", len(step), "span ms", (t1 - t0) / 1e6, "queues", sorted(set(r['Queue_Id'] for r in step)))
W = 250000
nb = (t1 - t0) // W + 1
for w in range(nb):
    lo, hi = t0 + w * W, t0 + (w + 1) * W
    per = collections.defaultdict(lambda: [0, 0, collections.Counter()])
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e <= lo or s >= hi: continue
        q = per[r['Queue_Id']]
        q[0] += 1; q[1] += min(e, hi) - max(s, lo); q[2][r['Kernel_Name'].split('(')[0].replace('void ', '')[:28]] += 1
    print("%5.2f ms:" % (w * W / 1e6), " | ".join("q%s n=%d busy=%3d%% %s" % (k, v[0], 100 * v[1] // W, ",".join("%s*%d" % kv for kv in v[2].most_common(2))) for k, v in sorted(per.items())))

"""Per-(kernel, grid, workgroup) duration statistics from a rocprofv3 kernel trace -> JSON (profiles/)."""
import csv, glob, json, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if not any(k in name for k in ('gru_step', 'gru_bwd_step', 'skinny_plain', 'attn_')):
        continue
    grid = "%sx%sx%s" % (r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
    wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
    agg[(name, grid, wg)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = []
for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    out.append({"kernel": name, "grid_threads": grid, "workgroup": wg, "calls": len(v), "avg_us": sum(v) / len(v) / 1e3,
                "median_us": v[len(v) // 2] / 1e3, "min_us": v[0] / 1e3, "p90_us": v[int(len(v) * 0.9)] / 1e3})
json.dump(out, open(sys.argv[2], "w"), indent=1)
for o in out[:14]:
    print("%-32s grid %-16s wg %-5s n=%6d avg %6.2f med %6.2f min %6.2f" % (o["kernel"][:32], o["grid_threads"], o["workgroup"], o["calls"], o["avg_us"], o["median_us"], o["min_us"]))

"""List the launches of one optimiser step that are neither recurrence-chain kernels nor GEMMs (rocprofv3 kernel trace)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
ends = idx[1::2]
a, b = ends[-4], ends[-3]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
skip = ('gru_step', 'gru_bwd_step', 'skinny_plain', 'attn_scores', 'attn_ctx', 'attn_dq', 'gemm_')
tot = 0
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if any(k in n for k in skip):
        continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print("%8.1f us  dur %6.1f  grid %sx%sx%s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], n[:70]))
print("total others %.1f us" % tot)

#!/bin/bash
# LDS bank-conflict share per kernel of a few training steps (tools/prof_decoder_fwd.py):  bash tools/pmc_lds.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out/lds
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d /tmp/pl -o l --output-format csv -- python3 $ROOT/tools/prof_decoder_fwd.py > $ROOT/gpurun_out/lds/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pl/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:48]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
rows = []
for n, c in agg.items():
    act = c.get('SQ_LDS_IDX_ACTIVE', 0.0)
    rows.append((c.get('SQ_LDS_BANK_CONFLICT', 0.0), n, act, c.get('SQ_INSTS_LDS', 0.0), c.get('SQ_WAVE_CYCLES', 0.0)))
rows.sort(reverse=True)
print('%-50s %14s %14s %8s %12s' % ('kernel', 'bank_conflict', 'lds_active', 'share', 'lds_insts'))
for bc, n, act, ins, wc in rows[:25]:
    print('%-50s %14.3e %14.3e %8.2f %12.3e' % (n, bc, act, bc / act if act else 0.0, ins))
PY

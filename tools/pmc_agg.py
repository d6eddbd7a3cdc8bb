import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
        if 'gemm_split' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-32s %.4g" % (c, v))

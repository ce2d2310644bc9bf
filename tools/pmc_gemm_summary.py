import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    if 'gemm_bf16nt' not in k: continue
    agg[k[:60]][r['Counter_Name']] += float(r['Counter_Value']); 
for k, d in agg.items():
    print(k)
    wc = d.get('SQ_WAVE_CYCLES', 0)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}  {(v / wc if wc else 0):7.3f} of WAVE_CYCLES")

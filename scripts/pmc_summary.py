"""Summarise rocprofv3 --pmc CSVs: average counter value per dispatch, grouped by kernel name."""
import csv, glob, sys, collections
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(pat, recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get('Kernel_Name', '')[:60]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if not any(s in k for s in sys.argv[2:]): continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")

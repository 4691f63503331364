import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
rows = rows[n // 2:]          # steady state
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
gaps = collections.defaultdict(lambda: [0, 0])
for a, b in zip(rows, rows[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    k = (a['Kernel_Name'][:40], b['Kernel_Name'][:40])
    gaps[k][0] += g; gaps[k][1] += 1
print(f"kernels {len(rows)} span {(t1 - t0) / 1e6:.2f} ms busy {busy / 1e6:.2f} ms gap {(t1 - t0 - busy) / 1e6:.2f} ms  mean gap {(t1 - t0 - busy) / len(rows) / 1e3:.2f} us")
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g / 1e6:8.3f} ms  n={c:5d}  {g / c / 1e3:7.2f} us/gap  {k[0]} -> {k[1]}")

"""Print the kernel sequence of the LAST network pass found in a rocprofv3 --kernel-trace CSV (name, us)."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a pass starts at k_pos_embed
starts = [i for i, n in enumerate(names) if "k_pos_embed" in n]
lo = starts[-1]
hi = len(rows)
tot = 0.0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    n = r["Kernel_Name"]
    n = n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {us:8.1f} us  grid {r.get('Grid_Size','?'):>9s}  {n[:90]}")
print(f"kernel sum {tot:.1f} us, span {(int(rows[hi-1]['End_Timestamp'])-t0)/1e3:.1f} us")

"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; see scripts/pmc_traffic.sh).
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM), hence the factor 2 on the read side.
usage: pmc_traffic_summary.py <dir with pmc_FETCH_SIZE/, pmc_WRITE_SIZE/> <out.json> [batch]"""
import csv, glob, json, re, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^>]*>)?", name)
    base = m.group(1)
    base = {"k_patch_embed_mfma": "k_patch_embed", "k_final_readout_mfma": "k_final_readout"}.get(base, base)
    # the names the library's own profiler tallies under (prof.hip): one per kernel symbol
    if base in ("k_gemm_nt", "k_mlp_stream", "k_wattn_ring") and m.group(2):
        targs = [t.strip() for t in m.group(2).strip("<>").split(",")]
        proj = base == "k_mlp_stream" and targs[-1] in ("true", "1")   # PROJ form: attention tail folded in
        return base + "<" + targs[0] + (",proj" if proj else "") + ">"
    if base == "k_row" and m.group(2):
        mode = m.group(2).strip("<>").split(",")[-1].strip()
        base += "<" + {"0": "cond_ln", "1": "merge_ln", "2": "concat", "3": "ln_f32", "4": "breakup_ln"}.get(mode, mode) + ">"
    return base


def load(dirname, counter):
    path = sorted(glob.glob(f"{dirname}/pmc_{counter}/**/*counter_collection.csv", recursive=True))[-1]
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


d, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 56
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from swingnn_amd import engine
launch_batch = engine.launch_batch(batch, bench.CONFIGS["grid"]["img_size"])  # graphs per kernel launch (part-batch)
f, nf = load(d, "FETCH_SIZE")
w, nw = load(d, "WRITE_SIZE")
res = {}
for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
    if not k.startswith("k_"):
        continue
    n = nf[k]
    res[k] = dict(launches=n, fetch_size_kib_per_launch=f[k] / n, write_size_kib_per_launch=w.get(k, 0.0) / max(nw.get(k, 1), 1),
                  hbm_bytes_per_launch=(2 * f[k] / n + w.get(k, 0.0) / max(nw.get(k, 1), 1)) * 1024)
json.dump(dict(batch=batch, launch_batch=launch_batch, method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/prof_forward.py; "
                      "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 read-side correction)", kernels=res),
          open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:24s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch']/1e6:9.2f} MB/launch")

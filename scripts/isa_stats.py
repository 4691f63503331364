"""Static instruction statistics of the kernels in one .hip file (device-only assembly of hipcc for gfx950).

    python scripts/isa_stats.py swingnn_amd/csrc/wattn_res.hip [name-filter] [--loops]

Per kernel: instruction counts by class (VALU / MFMA / LDS / VMEM / scratch / waits), registers, scratch bytes; with
--loops the same split per basic block that ends in a backward branch (the window / head loops).  Static counts: what
the compiler emitted, not what executed; used to see where VALU work and spill traffic come from before profiling.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    src = sys.argv[1]
    filt = [a for a in sys.argv[2:] if not a.startswith("--")]
    loops = "--loops" in sys.argv
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=200000", "-S",
               "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(os.path.abspath(src)),
               src, "-o", out] + [a for a in os.environ.get("EXTRA_HIPCC", "").split() if a]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if filt and not any(f in name for f in filt):
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        meta = {k: re.search(rf"\.set {re.escape(name)}\.{k}, (\d+)", text) for k in ("num_vgpr", "num_agpr", "private_seg_size")}
        meta = {k: (int(v.group(1)) if v else None) for k, v in meta.items()}
        tot = collections.Counter()
        ops = collections.Counter()
        blocks, cur, label = [], collections.Counter(), "entry"
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith((";", ".")) and not t.endswith(":"):
                continue
            if t.endswith(":") or re.match(r"^\.LBB\d+_\d+:", t):
                blocks.append((label, cur))
                cur, label = collections.Counter(), t.split(":")[0]
                continue
            op = t.split()[0]
            cls = classify(op)
            tot[cls] += 1
            ops[op] += 1
            cur[cls] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                cur["->" + t.split()[-1]] += 1
        blocks.append((label, cur))
        print(f"{dem[:110]}\n   vgpr {meta['num_vgpr']} agpr {meta['num_agpr']} scratch {meta['private_seg_size']} B/lane | "
              + " ".join(f"{k} {v}" for k, v in sorted(tot.items())))
        print("   top ops: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(28)))
        if loops:
            for lab, c in blocks:
                n = sum(v for k, v in c.items() if not k.startswith("->"))
                if n >= 40:
                    print(f"      {lab:12s} n={n:5d} " + " ".join(f"{k} {v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()

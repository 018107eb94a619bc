"""Turn the four rocprofv3 --pmc passes of tools/pmc_conv.py into profiles/<tag>_pmc_traffic.json, the file bench.py reads
`roofline.traffic` from.  One counter per pass, program directly after `--` (MI355X_MICROARCH.md, HBM section):

  cd /tmp && export TMPDIR=/tmp   # (on the GPU box: cd to the repo copy)
  for k in fwd wgrad; do for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 -d gpurun_out/pmc_${k}_${c} --output-format csv --kernel-trace --pmc $c -- python3 tools/pmc_conv.py $k 32 32 128; done; done
  python3 tools/pmc_traffic.py gpurun_out r02a 128

FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE counts 128-byte requests of wide streaming reads as 64 bytes on gfx950 ->
doubled.  The per-dispatch CSV rows of the two kernels are copied next to the summary so the numbers can be re-derived."""
import collections
import csv
import re
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNEL = {"fwd": "k_conv27", "wgrad": "k_conv_wgrad"}  # substrings: k_conv27r<0> / k_conv27<..>, k_conv_wgrad3 / k_conv_wgrad2<..>


def main():
    src, tag, size = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = {"source_hash": bench.kernel_source_hash(), "size": size, "shape": f"k3 s1 32->32 @{size}^3 batch 1", "counters_KB": {},
           "hbm_bytes_per_launch": {}}
    rows_out = []
    for kind, sub in KERNEL.items():
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            acc = collections.defaultdict(list)
            for f in glob.glob(os.path.join(src, f"pmc_{kind}_{counter}", "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and (sub in r["Kernel_Name"] or "k_wgrad_reduce" in r["Kernel_Name"]):
                        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
                        kname = m.group(0) if m else r["Kernel_Name"][:60]
                        acc[kname].append(float(r["Counter_Value"]))
                        rows_out.append({"pass": f"{kind}/{counter}", "kernel": kname,
                                         "dispatch": r.get("Dispatch_Id", ""), "value_KB": r["Counter_Value"]})
            # launches 2.. of each kernel (the first also pays cold caches / the plan's first touch)
            vals[counter] = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc.items()}
        out["counters_KB"][kind] = vals
        fetch = sum(vals["FETCH_SIZE"].values())
        write = sum(vals["WRITE_SIZE"].values())
        out["hbm_bytes_per_launch"][kind] = (2.0 * fetch + write) * 1024.0
    dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_dispatches.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["pass", "kernel", "dispatch", "value_KB"])
        w.writeheader()
        w.writerows(rows_out)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

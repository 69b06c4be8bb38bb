"""Turn a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES run of `bench.py --steps 1 --warmup 0 --lanes 1` into profiles/valu_latest.json:
VALU wave-instructions per launch of every kernel class of the step.  bench.py divides them by the live kernel times:
valu_frac = instructions x 2 cycles (a wave64 VALU instruction occupies its SIMD-32 for two cycles, MI355X_MICROARCH.md) /
(1024 SIMDs x 2.4 GHz x time) -- the share of the chip's VALU issue slots the class uses, i.e. how far it is from a compute bound.
usage: pmc_valu.py <pmc_dir> <workload> <out.json> [source-tag]"""
import csv
import glob
import json
import sys


def classify(name):
    if "k_group_pos" in name or "k_group_shape" in name:
        return "group_prepass"
    if "k_group" in name:
        return "group/" + ("wiener" if "dct8w" in name else "ht")
    if "k_aggregate" in name:
        return "aggregate/" + ("wiener" if ", 8, 8," in name else "ht")
    if "k_bm_scan" in name:
        return "scan/" + ("wiener" if ("<8>" in name or "<8," in name) else "ht")
    if "argmin" in name:
        return "argmin"
    if "k_self_select" in name or "k_self_trivial" in name:
        return "select"
    if "k_window_begin" in name or "k_window_end" in name:
        return "window_ends"
    return "other"


acc = {}
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = acc.setdefault(classify(r["Kernel_Name"]), {"SQ_INSTS_VALU": 0.0, "SQ_WAVES": 0.0, "ids": set()})
        if r["Counter_Name"] in e:
            e[r["Counter_Name"]] += float(r["Counter_Value"])
        if "_list" not in r["Kernel_Name"]:   # round 6: a group kernel's list launch (k_group_*_list) belongs to its pass: counters yes, launch count no
            e["ids"].add(r["Dispatch_Id"])
res = {"workload": sys.argv[2], "source": sys.argv[4] if len(sys.argv) > 4 else None,
       "unit": "VALU wave-instructions per launch (SQ_INSTS_VALU summed over the device)", "kernels": {}}
for k in sorted(acc):
    n = max(1, len(acc[k]["ids"]))
    res["kernels"][k] = {"launches": n, "valu_insts_per_launch": acc[k]["SQ_INSTS_VALU"] / n, "waves_per_launch": acc[k]["SQ_WAVES"] / n}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))

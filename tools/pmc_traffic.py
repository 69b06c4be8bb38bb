"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md
prescribes) of `bench.py --steps 1 --warmup 0 --lanes 1` into profiles/traffic_latest.json:
HBM bytes per launch of the roofline kernels (k_group* + k_aggregate), for the whole step and split
into the HT and the Wiener pass.

gfx950 corrections applied (MI355X_MICROARCH.md, HBM section): rocprofv3 reports the two counters in
units of 1 KB (bytes = value * 1024); FETCH_SIZE tallies the L2's 128-byte memory-side read requests
at 64 bytes, i.e. reports HALF of the bytes fetched -- for EVERY access shape: round 6's calibration
(tools/fetch_calib.hip, profiles/r06_b_fetch_calib.txt) reads a 2 GiB buffer exactly once with
16-byte-per-lane streaming loads, 4-byte-per-lane streaming loads and the aggregation's own shape
(four 64-byte rows of a 1 KB patch per wave-instruction, scattered patches) and the counter says
1.074 GB for all of them (TCC_EA0_RDREQ: 16.8 M requests = 128 B each).  Rounds 1-5 took the
aggregation's figure raw ("calibrated": raw 3.29 GB ~ filt's 3.46 GB was a coincidence) -- the
aggregation really fetches 1.9x (HT) / 2.4x (Wiener) the size of filt, because a filtered row is read
by the two or three tiles it overlaps and their reads are too far apart in time for the 4 MB L2.
usage: pmc_traffic.py <fetch_dir> <write_dir> <workload> <out.json> [source-tag]"""
import csv
import glob
import json
import sys


def classify(name):
    if "k_group_pos" in name or "k_group_shape" in name:
        return None
    if "k_group" in name:
        return ("group", "wiener" if "dct8w" in name else "ht")
    if "k_aggregate" in name:
        return ("aggregate", "wiener" if ", 8, 8," in name else "ht")
    if "k_bm_scan" in name:
        return ("scan", "wiener" if ("<8>" in name or "<8," in name) else "ht")
    if "argmin" in name:
        return ("argmin", "both")
    return None


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            key = classify(r["Kernel_Name"])
            if key is None:
                continue
            e = out.setdefault(key, [0.0, set()])
            e[0] += float(r["Counter_Value"])
            if "_list" not in r["Kernel_Name"]:   # round 6: k_group_*_list belongs to the pass of its main kernel: bytes yes, launch count no
                e[1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in out.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"workload": sys.argv[3], "source": sys.argv[5] if len(sys.argv) > 5 else None, "unit": "bytes",
       "fetch_correction": "x2 for every kernel (profiles/r06_b_fetch_calib.txt)", "kernels": {}, "per_step": {}}
tot = {"ht": [0.0, 0.0, 0], "wiener": [0.0, 0.0, 0]}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 1))
    w, nw = write.get(k, (0.0, 1))
    fb, wb = f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)
    res["kernels"]["%s/%s" % k] = {"launches": nf, "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb,
                                   "write_bytes_per_launch": wb}
    if k[0] == "group" and k[1] in tot:        # gather of window pixels: coalesced 16-byte row segments -> the 1/2 rule applies
        tot[k[1]][0] += fb + wb; tot[k[1]][1] += 2 * fb + wb; tot[k[1]][2] = max(tot[k[1]][2], nf)
    if k[0] == "aggregate" and k[1] in tot:    # ... and so are the aggregation's row gathers (round 6 calibration)
        tot[k[1]][0] += fb + wb; tot[k[1]][1] += 2 * fb + wb; tot[k[1]][2] = max(tot[k[1]][2], nf)
for kind, (raw, corr, n) in tot.items():
    res["per_step"][kind] = {"hbm_bytes_per_launch": corr, "hbm_bytes_per_launch_fetch_uncorrected": raw, "launches": n}
n = tot["ht"][2] + tot["wiener"][2]
# one pass launches one group kernel and one aggregate kernel: the pair's traffic per pass, averaged over the step
res["hbm_bytes_per_launch"] = (tot["ht"][1] * tot["ht"][2] + tot["wiener"][1] * tot["wiener"][2]) / max(n, 1)
res["hbm_bytes_per_launch_fetch_uncorrected"] = (tot["ht"][0] * tot["ht"][2] + tot["wiener"][0] * tot["wiener"][2]) / max(n, 1)
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res, indent=1))

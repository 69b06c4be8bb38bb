"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md
prescribes) of `bench.py --steps 1 --warmup 0` into profiles/traffic_latest.json:
HBM bytes per launch of the roofline kernels (k_group* + k_aggregate).

gfx950 corrections applied (MI355X_MICROARCH.md, HBM section): the counters are in KiB... they are
reported in units of 1 KB by rocprofv3 (bytes = value * 1024); FETCH_SIZE reports half the bytes of
wide coalesced streaming reads, so the read side is given both raw and doubled.
usage: pmc_traffic.py <fetch_dir> <write_dir> <workload> <out.json>"""
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            key = "group" if ("k_group" in name and "k_group_pos" not in name and "k_group_shape" not in name) else \
                  "aggregate" if "k_aggregate" in name else "scan" if "k_bm_scan" in name else "argmin" if "argmin" in name else None
            if key is None:
                continue
            e = out.setdefault(key, [0.0, set()])
            e[0] += float(r["Counter_Value"])
            e[1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in out.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"workload": sys.argv[3], "applies_to": ["lf17x17x512x512_sigma25", "lf9x9x512x512_sigma25"],
       "note": "per-pass figures depend on the 3x3x560x560 window only, not on the number of windows", "unit": "bytes", "kernels": {}}
pair_raw = pair_corr = 0.0
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 1))
    w, nw = write.get(k, (0.0, 1))
    fb, wb = f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)
    res["kernels"][k] = {"launches": nf, "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb,
                         "write_bytes_per_launch": wb}
    if k == "group":        # gather of window pixels: coalesced row segments -> the 1/2 rule applies
        pair_raw += fb + wb
        pair_corr += 2 * fb + wb
    if k == "aggregate":    # calibrated on a known byte count: filt is read exactly once and the raw
        pair_raw += fb + wb  # FETCH_SIZE equals its size (4 B/lane row-segment reads are not halved)
        pair_corr += fb + wb
# one pass launches one group kernel and one aggregate kernel: the pair's traffic per pass
res["hbm_bytes_per_launch"] = pair_corr
res["hbm_bytes_per_launch_fetch_uncorrected"] = pair_raw
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res, indent=1))

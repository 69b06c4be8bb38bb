#!/bin/bash
# FETCH_SIZE calibration (tools/fetch_calib.hip) on the GPU box:  gpurun -- 'bash tools/fetch_calib.sh'  ->  gpurun_out/fetch_calib.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o /tmp/fetch_calib tools/fetch_calib.hip 2>/dev/null
rm -rf gpurun_out/fcal; mkdir -p gpurun_out/fcal
/tmp/fetch_calib > gpurun_out/fetch_calib.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=gpurun_out/fcal/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- /tmp/fetch_calib > /dev/null 2>&1
done
python3 - >> gpurun_out/fetch_calib.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/fcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_stream" in n or "k_rows64" in n:
            acc[n.split("(")[0]][r["Counter_Name"]] = acc[n.split("(")[0]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, {a: "%.4g" % b for a, b in sorted(v.items())}, "FETCH_SIZE x 1024 = %.3f GB" % (v.get("FETCH_SIZE", 0) * 1024 / 1e9))
PY
rm -rf gpurun_out/fcal
cat gpurun_out/fetch_calib.txt

# VALU / LDS / VMEM instruction counts of k_aggregate for the two scan forms (one window pass per step, tools/pass_time.py 1).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in vec4 scalar; do
  rm -rf gpurun_out/pmcab; [ $v = scalar ] && export LFBM5D_AGG_SCALAR_SCAN=1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcab -- python3 tools/pass_time.py 1 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmcab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_aggregate" in r["Kernel_Name"]:
            acc["W" if "<true" in r["Kernel_Name"] else "HT"][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(sys.argv[1], k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
done

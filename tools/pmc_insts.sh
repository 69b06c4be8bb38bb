# VALU / LDS / VMEM instruction counts per kernel of one headline step (one lane): where the issue slots go.
#   gpurun -- 'bash tools/pmc_insts.sh'   ->  gpurun_out/pmc_insts/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_insts; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/a -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --noise torch --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/b -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --noise torch --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY' > $out/summary.txt
import csv, glob, collections
for tag in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"gpurun_out/pmc_insts/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("lfbm5d::", "").split("(")[0][:34]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    names = sorted({c for k in acc for c in acc[k]})
    print("%-36s %6s " % ("kernel (per launch)", "n") + " ".join("%18s" % c for c in names))
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:14]:
        print("%-36s %6d " % (k, len(n[k])) + " ".join("%18.4g" % (acc[k][c] / len(n[k])) for c in names))
PY
cat $out/summary.txt
rm -rf $out/a $out/b

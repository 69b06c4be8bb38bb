# LDS / VALU counters of the group kernels on one window pass per step (tools/pass_time.py); run on the GPU box:
#   gpurun -- 'bash tools/pmc_lds.sh [library]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$1" ] && export LFBM5D_HIP_LIB=$PWD/$1
rm -rf gpurun_out/pmcl
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcl/a -- python3 tools/pass_time.py 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmcl/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]; key=None
        for k in ("k_group_dct8w","k_group_id","k_aggregate<false","k_aggregate<true"):
            if k in n: key=k
        if key: acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
for k,v in acc.items():
    n=max(1,len(cnt[k]))
    print(k, "launches", n, {a:"%.3g"%(b/n) for a,b in sorted(v.items())})
PY

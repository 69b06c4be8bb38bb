# SQ counters of the group / aggregation kernels on one 3x3x512x512 window pass per step (tools/pass_time.py); run on the GPU box:
#   gpurun -- "bash tools/pmc_group.sh [library]"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$1" ] && export LFBM5D_HIP_LIB=$PWD/$1
rm -rf gpurun_out/pmcg
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d gpurun_out/pmcg/a -- python3 tools/pass_time.py 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcg/b -- python3 tools/pass_time.py 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/pmcg/c -- python3 tools/pass_time.py 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmcg/*/*/*counter_collection.csv"):
    p=f.split("/")[2]
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        key=None
        for k in ("k_group_dct8w3","k_group_dct8w2","k_group_id","k_aggregate2<false","k_aggregate2<true","k_aggregate<false","k_aggregate<true"):
            if k in n: key=k
        if key:
            acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][p].add(r["Dispatch_Id"])
for k,v in acc.items():
    n=max(len(s) for s in cnt[k].values())
    print(k, "launches", n, {a:"%.3g"%(b/n) for a,b in sorted(v.items())})
PY

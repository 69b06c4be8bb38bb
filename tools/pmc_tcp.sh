# L1 (TCP) / L2 (TCC) request counters of the group kernels on one window pass per step (tools/pass_time.py); run on the GPU box:
#   gpurun -- 'bash tools/pmc_tcp.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmct
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum --output-format csv -d gpurun_out/pmct/a -- python3 tools/pass_time.py 1 > gpurun_out/pmct_a.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmct/b -- python3 tools/pass_time.py 1 > gpurun_out/pmct_b.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmct/c -- python3 tools/pass_time.py 1 > gpurun_out/pmct_c.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmct/*/*/*counter_collection.csv"):
    p=f.split("/")[2]
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]; key=None
        for k in ("k_group_dct8w3","k_group_id_haar","k_aggregate<false","k_aggregate<true","k_bm_scan2<16","k_bm_scan2<8"):
            if k in n: key=k
        if key: acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][p].add(r["Dispatch_Id"])
for k,v in acc.items():
    n=max(len(s) for s in cnt[k].values())
    print(k, "launches", n, {a:"%.3g"%(b/n) for a,b in sorted(v.items())})
PY
tail -3 gpurun_out/pmct_a.log | cut -c1-200

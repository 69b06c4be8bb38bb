cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcs
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/pmcs -- python3 tools/gpu_check.py readme > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcs/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    key=None
    for k in ("k_bm_scan<16>","k_bm_scan<8>","k_group_dct8","k_group_id","k_aggregate","k_stereo_argmin"):
        if k in n: key=k
    if key: acc[key][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    print(k, {a:int(b) for a,b in v.items()})
PY

# SQ counters of k_group_dct8w3 for two library variants (base = HEAD, new = working tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/w3qc; rm -rf $out; mkdir -p $out
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"
)
for v in ${VARIANTS:-base new}; do
  export LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so
  i=0
  for s in "${sets[@]}"; do
    rocprofv3 --pmc $s --output-format csv -d $out/$v/p$i -- python3 tools/pass_time.py 1 > /dev/null 2>&1
    i=$((i+1))
  done
done
python3 - $out ${VARIANTS:-base new} <<'PY' | tee $out/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
for v in sys.argv[2:]:
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(f"{out}/{v}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_group_dct8w3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(v, {c: round(x / max(1, len(n[c]))) for c, x in sorted(acc.items())})
PY

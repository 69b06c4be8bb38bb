# SQ counters of one kernel (name substring) over an arbitrary python script: bash tools/scratch/sq_cmd.sh <kernel> <script.py> [args...]
kern=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sqcmd; rm -rf $out; mkdir -p $out
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
)
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d $out/p$i -- python3 "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 - "$kern" "$out" <<'PY'
import csv, glob, collections, sys
kern, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
per = {c: v / max(1, len(n[c])) for c, v in acc.items()}
g = lambda c: per.get(c, float("nan"))
wc = g("SQ_WAVE_CYCLES")
print("issue: VALU %.2f  LDS %.2f  VMEM %.2f  scalar %.2f  any %.2f;  wait %.2f;  occupancy %.2f waves/SIMD;  lds_conflict %.2f;  l2_hit %.2f" % (
    g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_SCA") / wc, g("SQ_ACTIVE_INST_ANY") / wc,
    g("SQ_WAIT_INST_ANY") / wc, wc / (4 * g("SQ_BUSY_CYCLES")), g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")),
    g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
print("instructions per launch: VALU %.3g  SALU %.3g  LDS %.3g  VMEM read %.3g  VMEM write %.3g;  waves %.3g  (VALU per wave %.0f, LDS per wave %.0f, SALU per wave %.0f)" % (
    g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_WAVES"), g("SQ_INSTS_VALU") / max(1.0, g("SQ_WAVES")), g("SQ_INSTS_LDS") / max(1.0, g("SQ_WAVES")), g("SQ_INSTS_SALU") / max(1.0, g("SQ_WAVES"))))
print("raw: " + "  ".join(f"{c}={per[c]:.4g}" for c in sorted(per)))
PY
rm -rf $out

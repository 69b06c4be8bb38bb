# TCP / UTCL1 counters of one kernel (name substring) over a program: bash tools/scratch/tcp_cmd.sh <kernel> <program> [args...]
# (the TA_* / TD_* sets hung rocprofv3 on this pool: left out; every run under its own timeout)
kern=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tcpcmd; rm -rf $out; mkdir -p $out
sets=(
 "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"
 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_TOTAL_ACCESSES"
 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1"
 "TCP_TOTAL_READ TCP_TOTAL_WRITE TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY"
)
i=0
for s in "${sets[@]}"; do
  timeout 120 rocprofv3 --pmc $s --output-format csv -d $out/p$i -- "$@" > $out/log$i.txt 2>&1
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
for c in sorted(acc): print(f"{c:48s} {acc[c]/max(1,len(n[c])):.4g}   ({len(n[c])} launches)")
PY

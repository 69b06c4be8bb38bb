# Per-kernel average durations of one window pass per step (tools/pass_time.py) under rocprofv3; run on the GPU box:
#   gpurun -- 'bash tools/kstat.sh [pattern]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstat -o s -- python3 tools/pass_time.py 10 > /dev/null 2>&1
python3 - "$1" <<PY
import csv,glob,sys
pat=sys.argv[1] if len(sys.argv)>1 else ""
f=glob.glob("gpurun_out/kstat/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))[:16]:
    if pat in r["Name"]: print("%-72s %5s calls %8.3f ms avg"%(r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e6))
PY

# rocprofv3 kernel averages of an arbitrary python script: bash tools/scratch/kstat_cmd.sh <script.py> [args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstat -o s -- python3 "$@" > gpurun_out/kstat_stdout.txt 2>&1
tail -3 gpurun_out/kstat_stdout.txt
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kstat/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    print("%-80s %5s calls %8.3f ms avg"%(r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e6))
PY

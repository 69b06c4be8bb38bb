# per-kernel averages of one wide-window pass: bash tools/scratch/ww_kstat.sh <aw> <step>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/wwk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wwk -o s -- python3 tools/wide_window_time.py $1 $2 1 > gpurun_out/wwk_out.txt 2>&1
grep window gpurun_out/wwk_out.txt
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/wwk/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))[:8]:
    print("%-90s %5s calls %9.3f ms avg %9.3f ms total"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 8 4 3 0; do
  rm -rf gpurun_out/profd
  LFBM5D_SCAN_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profd -- python3 tools/gpu_check.py readme > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/profd/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
out=[]
for r in rows:
    n=r["Kernel_Name"]
    if "scan" in n or "argmin" in n or "select" in n:
        out.append("%s=%d"%(n.split("::")[-1][:14], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))//1000))
print("debug=$d", " ".join(out))
PY
done

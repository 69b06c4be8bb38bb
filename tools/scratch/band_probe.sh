cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bands.py -x -q -m gpu 2>&1 | tail -5
for mb in 0 2000 1000 600 400 250 150; do
  echo "== LFBM5D_BAND_MB=$mb"
  if [ $mb = 0 ]; then unset LFBM5D_BAND_MB; else export LFBM5D_BAND_MB=$mb; fi
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/kstat
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstat -o s -- python3 tools/pass_time.py 6 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kstat/**/*kernel_stats.csv", recursive=True)[0]
tot={}
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for key in ("k_group_id_haar","k_group_dct8w3","k_aggregate<false","k_aggregate<true","k_group_pos","k_group_shape"):
        if key in n:
            tot[key]=tot.get(key,0)+float(r["TotalDurationNs"])
print({k: round(v/1e6/7,3) for k,v in sorted(tot.items())}, "ms per pass (7 passes per step kind)")
PY
done
rm -rf gpurun_out/kstat

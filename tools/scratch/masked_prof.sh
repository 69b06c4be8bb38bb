cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/masked_prof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o s -- python3 tools/masked_lf_time.py 1 > $out/run.log 2>&1
cp $(find $out/st -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/st
head -16 $out/kernel_stats.csv | cut -c1-150

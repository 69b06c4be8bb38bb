cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/cfg3; rm -rf $out; mkdir -p $out
python3 bench.py --workload lf17x17x512x512_sigma10_bior --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.log 2>&1; tail -1 $out/bench.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o s -- python3 bench.py --workload lf17x17x512x512_sigma10_bior --steps 1 --warmup 1 --lanes 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find $out/st -name "*kernel_stats.csv" | head -1) $out/kernel_stats_lanes1.csv; rm -rf $out/st
head -14 $out/kernel_stats_lanes1.csv | cut -c1-160

# Round profile collection (run on the GPU box):  gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r01_d'
# 1) headline bench line incl. CPU baseline, 2) rocprofv3 kernel trace + stats of the same command (no CPU leg),
# 3) FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, counters only) on the 9x9 workload (same per-pass figures).
tag=${1:-r01_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
python3 bench.py --gpus 1 --steps 2 --warmup 1 > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload lf9x9x512x512_sigma25 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload lf9x9x512x512_sigma25 > /dev/null 2>&1
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write lf9x9x512x512_sigma25 $out/traffic.json > /dev/null
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/stats/*/*kernel_trace.csv
ls -la $out; cut -c1-400 $out/bench.json

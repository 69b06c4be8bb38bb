# Round profile collection (run on the GPU box):  gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02_a'
# 1) headline bench line incl. CPU baseline (default execution: the fused two-step job on two window lanes),
# 2) rocprofv3 kernel trace + stats of `bench.py --lanes 1` (kernels alone on the GPU: the durations the roofline uses)
#    and of the default command (lanes overlap kernels of different windows),
# 3) FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, counters only) on the headline workload -> traffic.json,
# 4) an SQ_INSTS_VALU pass of the same command -> valu.json (bench.py's roofline.valu: how far each class is from VALU issue).
# Only the summaries are kept (copy bench.json, kernel_stats_*.csv, traffic.json, valu.json to profiles/ as <tag>_*; traffic.json and
# valu.json also as profiles/traffic_latest.json / valu_latest.json, which bench.py reads).
tag=${1:-r04_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
python3 bench.py --gpus 1 --steps 2 --warmup 1 > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o s -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-seam > $out/bench_prof_lanes1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -o s -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-seam > $out/bench_prof_default.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-seam > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-seam > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/pmc_valu -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-seam > /dev/null 2>&1
python3 tools/pmc_valu.py $out/pmc_valu lf17x17x512x512_sigma25 $out/valu.json "profiles/${tag}_valu.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES, bench.py --steps 1 --warmup 0 --lanes 1)" > /dev/null
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write lf17x17x512x512_sigma25 $out/traffic.json "profiles/${tag}_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bench.py --steps 1 --warmup 0 --lanes 1)" > /dev/null
cp $(find $out/stats1 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_lanes1.csv
cp $(find $out/stats3 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_default.csv
rm -rf $out/stats1 $out/stats3
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_valu
ls -la $out; cut -c1-600 $out/bench.json

# Kernel timeline of one two-step job under rocprofv3 (run on the GPU box): gpurun -- 'bash tools/timeline.sh [lanes]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
LFBM5D_LANES=${1:-2} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-seam > gpurun_out/tl_bench.json 2>gpurun_out/tl_err.log
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python3 tools/timeline_overlap.py $f 0.55 0.95

# A/B of the software-pipelined consume phase of k_aggregate (LFBM5D_AGG_PIPE), same box, same library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pipe
for rep in 1 2; do for v in base pipe; do
  if [ $v = pipe ]; then export LFBM5D_AGG_PIPE=1; else unset LFBM5D_AGG_PIPE; fi
  echo "$v: $(python tools/pass_time.py 10 2>&1 | grep step | cut -c1-62 | tr '\n' '|')"
done; done 2>&1 | tee gpurun_out/pipe/ab.txt
export LFBM5D_AGG_PIPE=1
python -m pytest tests -q -m gpu -x -k "core_pass or window_pass or random_conf" 2>&1 | grep -E "passed|failed|Error" | tee -a gpurun_out/pipe/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base pipe; do
  if [ $v = pipe ]; then export LFBM5D_AGG_PIPE=1; else unset LFBM5D_AGG_PIPE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pipe/st_$v -o s -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --lanes 1 --no-cpu-baseline > /dev/null 2>&1
  grep -h k_aggregate $(find gpurun_out/pipe/st_$v -name "*kernel_stats.csv") | cut -c1-200 | tee -a gpurun_out/pipe/ab.txt
  rm -rf gpurun_out/pipe/st_$v
done

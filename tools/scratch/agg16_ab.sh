# A/B of the packed-16-bit / EXEC-masked consume phase of k_aggregate: base = HEAD, new = working tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/agg16
#python -m pytest tests -q -m gpu -x -k "core_pass or window_pass or random_conf or headline_window or full_size or big or useSD or sd" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 | tee gpurun_out/agg16/ab.txt
for rep in 1 2; do for v in base pk16 nooobw5; do
  echo "$v: $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 2>&1 | grep step | cut -c1-62 | tr '\n' '|')"
done; done 2>&1 | tee -a gpurun_out/agg16/ab.txt

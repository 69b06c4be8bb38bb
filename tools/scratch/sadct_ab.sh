cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sadct
python -m pytest tests -q -m gpu -x -k "sadct or empty or holes or wien or core_pass or 16x16" 2>&1 | grep -E "passed|failed|Error|assert" | head | tee gpurun_out/sadct/ab.txt
for rep in 1 2; do for v in ${VARIANTS:-base new}; do for e in "" 2; do
  echo "$v empty [$e]: $(PASS_TIME_EMPTY=$e LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 5 2>&1 | grep step | cut -c1-62 | tr '\n' '|')"
done; done; done 2>&1 | tee -a gpurun_out/sadct/ab.txt

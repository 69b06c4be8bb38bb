cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/t16
for rep in 1 2; do for v in base new; do
  echo "$v: bior $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 512 bior 2>&1 | grep "step 1" | cut -c18-50) | dct $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 512 dct 2>&1 | grep "step 1" | cut -c18-50)"
done; done 2>&1 | tee gpurun_out/t16/ab2.txt
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | head -20 | tee -a gpurun_out/t16/ab2.txt

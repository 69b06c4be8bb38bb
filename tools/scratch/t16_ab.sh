# A/B of the two-round form of k_group_bior16_haar: base = HEAD, new = working tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/t16
python -m pytest tests -q -m gpu -x -k "bior or k16 or config4 or full_size or dct16 or generic" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 | tee gpurun_out/t16/ab.txt
for rep in 1 2; do for v in ${VARIANTS:-base new}; do
  echo "$v: $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 512 bior 2>&1 | grep "step 1" | cut -c1-62) | id: $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 512 id 2>&1 | grep "step 1" | cut -c18-62)"
done; done 2>&1 | tee -a gpurun_out/t16/ab.txt

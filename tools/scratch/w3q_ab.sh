# A/B of the quad form of k_group_dct8w3: HEAD (base), working tree (new: 128 VGPRs), no register cap (nocap)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/w3q
python -m pytest tests -q -m gpu -x -k "wien or headline_window or full_size or dct8w" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 | tee gpurun_out/w3q/ab.txt
for rep in 1 2; do for v in base new; do
  echo "$v: $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 2>&1 | grep "step 2" | cut -c1-62)"
done; done 2>&1 | tee -a gpurun_out/w3q/ab.txt

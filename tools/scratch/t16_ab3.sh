cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base r40w3ns r24w4; do
  echo "$v: bior $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 512 bior 2>&1 | grep "step 1" | cut -c18-50) | dct $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 512 dct 2>&1 | grep "step 1" | cut -c18-50)"
done; done

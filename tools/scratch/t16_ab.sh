# bior16 / dct16 HT kernels: variants of lfbm5d_group_ht.hip against the product build, one 560^2 window pass at configs[3]'s parameters
cd "$(dirname "$0")/../.."
for r in 1 2; do for v in base "$@"; do
  lib=$PWD/lfbm5d_amd/variants/lib_t16_$v.so; [ $v = base ] && lib=$PWD/lfbm5d_amd/liblfbm5d_hip.so
  echo "$v: $(PASS_TIME_SIGMA=10 LFBM5D_HIP_LIB=$lib python3 tools/pass_time.py 10 512 bior dct 2>&1 | grep 'step 1' | cut -c1-70) | $(PASS_TIME_SIGMA=10 LFBM5D_HIP_LIB=$lib python3 tools/pass_time.py 10 512 dct dct 2>&1 | grep 'step 1' | cut -c18-60)"
done; done

#!/bin/bash
# Same-box A/B of library variants (lfbm5d_amd/variants/lib_<name>.so): tools/pass_time.py per variant, round-robin, <reps> rounds.
#   gpurun -- 'bash tools/ab_variants.sh 2 base stnt ldnt'
cd "$(dirname "$0")/.."
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  echo "$v: $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python3 tools/pass_time.py 10 2>&1 | grep step | cut -c1-66 | tr '\n' '|')"
done; done

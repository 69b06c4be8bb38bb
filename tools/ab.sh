#!/bin/bash
# Same-box A/B of two builds of the library (the gpurun boxes differ by +-3 %, so absolute numbers of different calls do
# not compare): builds HEAD's library as variants/lib_base.so (from a clean export of HEAD, the working tree is not
# touched) and the working tree's as variants/lib_new.so; run
#   gpurun -- 'bash tools/ab.sh run [pass|bench]'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p lfbm5d_amd/variants
  make -s -C lfbm5d_amd/csrc >/dev/null && cp lfbm5d_amd/liblfbm5d_hip.so lfbm5d_amd/variants/lib_new.so
  tmp=$(mktemp -d)
  git archive HEAD lfbm5d_amd/csrc include | tar -x -C "$tmp"
  make -s -C "$tmp/lfbm5d_amd/csrc" ../liblfbm5d_hip.so >/dev/null && cp "$tmp/lfbm5d_amd/liblfbm5d_hip.so" lfbm5d_amd/variants/lib_base.so
  rm -rf "$tmp"
  ls -la lfbm5d_amd/variants/
else
  mode=${2:-pass}
  for rep in 1 2; do for v in base new; do
    if [ "$mode" = pass ]; then
      echo "$v: $(LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python tools/pass_time.py 10 2>&1 | grep step | cut -c1-62 | tr '\n' '|')"
    else
      LFBM5D_HIP_LIB=$PWD/lfbm5d_amd/variants/lib_$v.so python bench.py --steps 3 --warmup 1 --noise torch --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],1))"
    fi
  done; done
fi

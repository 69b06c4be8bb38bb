# The PMC passes of tools/collect_profiles.sh alone (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; separate counter-only runs) -> traffic.json, valu.json
#   gpurun --timeout 1200 -- 'bash tools/collect_pmc.sh r06_h'
tag=${1:-r06_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-seam > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-seam > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/pmc_valu -- python3 bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-seam > /dev/null 2>&1
python3 tools/pmc_valu.py $out/pmc_valu lf17x17x512x512_sigma25 $out/valu.json "profiles/${tag}_valu.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES, bench.py --steps 1 --warmup 0 --lanes 1)" > /dev/null
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write lf17x17x512x512_sigma25 $out/traffic.json "profiles/${tag}_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bench.py --steps 1 --warmup 0 --lanes 1)" > /dev/null
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_valu
cat $out/traffic.json | head -50

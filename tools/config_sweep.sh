# bench.py on every BASELINE.json configuration (one box, default execution): the table of DESIGN.md section 6.
#   gpurun --timeout 2400 -- 'bash tools/config_sweep.sh r04_g'   ->  gpurun_out/<tag>/configs.txt (+ the bench lines)
tag=${1:-r04_x}
cd $GRAFT_REPO_ROOT; out=gpurun_out/$tag; mkdir -p $out
for wl in lf3x3x256x256_sigma25 lf3x3x256x256_sigma25_dct lf9x9x512x512_sigma25 lf17x17x512x512_sigma25 lf17x17x512x512_sigma10_bior lf15x15x625x434_sigma50_n1; do
  steps=3; case $wl in lf3x3*) steps=50;; lf9x9*) steps=8;; esac
  python3 bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline > $out/$wl.log 2>&1
  tail -1 $out/$wl.log > $out/$wl.json
  python3 - $out/$wl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ps = d.get("roofline", {}).get("per_step", {})
pair = lambda k: round(ps.get(k, {}).get("avg_launch_ms", 0.0), 2)
q = d.get("psnr", {})
print(f"{d['config']['workload']:34s} {d['value']:7.1f} SAI-MP/s {d['ms_per_step']:8.1f} ms/step  HT pair {pair('ht')} Wiener pair {pair('wiener')} ms  psnr {q.get('noisy', 0):.2f} -> {q.get('basic', 0):.2f} -> {q.get('denoised', 0):.2f}")
PY
done | tee $out/configs.txt

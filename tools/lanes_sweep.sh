# Same-box sweep of the window-lane count for both job forms (run on the GPU box):  gpurun -- 'bash tools/lanes_sweep.sh'
for job in fused two-calls; do for l in 1 2 3 4 5 6; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --noise torch --job $job --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$job lanes $l', round(d['value'],1), round(d['ms_per_step'],1))"
done; done

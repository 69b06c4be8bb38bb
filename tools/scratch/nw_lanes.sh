for nw in 11 7; do for l in 1 2 3 4; do
  LFBM5D_SCAN_NW=$nw python bench.py --steps 4 --warmup 1 --no-cpu-baseline --noise torch --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nw $nw lanes $l', round(d['value'],1), round(d['ms_per_step'],1), round(d['kernel_ms_per_step']['block_matching'],1), d['psnr']['denoised'])"
done; done

for aw in 9 13 17; do
  for mb in "" 6144 3072 1536; do
    LFBM5D_BAND_MB=$mb timeout 300 python3 tools/wide_window_time.py $aw 1 1 2>&1 | grep window
  done
done

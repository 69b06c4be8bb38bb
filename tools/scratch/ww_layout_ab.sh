# SAI-major against group-major filt on wide windows: per-class times of one window pass (tools/wide_window_time.py)
for aw in 11 13 17; do
  for gm in "" 1; do
    LFBM5D_FILT_GROUP_MAJOR=$gm timeout 600 python3 tools/wide_window_time.py $aw ${1:-1} 1 2>&1 | grep window | sed "s/^/group_major=${gm:-0} /"
  done
done

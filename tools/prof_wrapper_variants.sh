export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "base" ]; then unset MCBS_LIBRARY; else export MCBS_LIBRARY=$PWD/marlon_amd/libmcbs_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3h/stats_$v -o run -- python3 tools/profile_run.py wrapper:headline 100 > gpurun_out/r3h/stats_$v.log 2>&1
  f=$(find gpurun_out/r3h/stats_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep wrapper_fused $f | awk -F, '{print $(NF-4), $(NF-2), $(NF-1)}')"
  rm -rf gpurun_out/r3h/stats_$v
done

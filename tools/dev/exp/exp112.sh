cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r06w; mkdir -p $OUT; rm -f $OUT/wider_systems_kernel_stats.csv
for sys in ROCKETLANDING CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$sys -o kt -- python tools/dev/wider_one.py $sys HERMITE_SIMPSON 4096 30 3 > $OUT/kt_$sys.log 2>&1
  f=$(find $OUT/kt_$sys -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "# $sys HERMITE_SIMPSON N=100 B=4096, 30 iterations, 3 solves"; grep -E "Name|hs_solve" $f; } >> $OUT/wider_systems_kernel_stats.csv
  rm -rf $OUT/kt_$sys
done
cat $OUT/wider_systems_kernel_stats.csv | cut -c1-300

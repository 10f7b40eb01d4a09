cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pyr
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pyr -- python tools/pyramid_bench.py $PYR_ARGS > gpurun_out/prof_pyr.log 2>&1
f=$(ls gpurun_out/prof_pyr/*/*kernel_stats.csv | head -1)
head -16 $f | cut -c1-160
cp $f gpurun_out/pyr_kernel_stats.csv
rm -rf gpurun_out/prof_pyr

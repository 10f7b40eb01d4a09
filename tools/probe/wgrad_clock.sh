# effective shader clock + MFMA-busy of conv_wgrad3_db_kernel under its ablation bits (one rocprofv3 --pmc pass per setting)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 0 3 2 1; do
  rm -rf gpurun_out/wgclk_$dbg
  WGRAD_PROBE_DBGS=$dbg WGRAD_PROBE_SHAPES=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/wgclk_$dbg -- python tools/probe/wgrad_probe.py > gpurun_out/wgclk_$dbg.log 2>&1
  echo "== dbg $dbg rc=$?"
  python tools/pmc_clock.py gpurun_out/wgclk_$dbg | grep wgrad3_db
done

export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/r3b
for ab in 0 1 2 3 4 8 15; do
  cd /tmp && DPK_GEMM_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/r3b/t$ab -o t --output-format csv -- python $R/tools/bench_small.py 4096 > /dev/null 2>&1; cd $R
  echo "ablate=$ab"; python tools/trace_summary.py gpurun_out/r3b/t$ab small_kernel | tail -1
done
rm -rf gpurun_out/r3b/t*

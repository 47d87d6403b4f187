# developer timing on the GPU box: the 256x256 tile's epilogue with parts ablated
#   for v in 1 2 3; do python tools/make_variant.py segabl$v gemm_split_big.hip -DLOCOV_BIG_EPI_ABLATE=$v; done
# 1 = no residual loads, 2 = no per-ROI column walk (mean-fused form), 3 = both.  Measured (8 000 proposals): mean-fused conv3
# 2.37 / 2.01 / 2.18 / 1.79 ms, conv3 with split-layout residual and output 2.58 / 2.08 ms.
for lib in product segabl1 segabl2 segabl3; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  echo "== $lib"; timeout 200 python tools/attic/dbg_segmean_big.py 2>&1 | grep "big = 1" | tail -2
  timeout 200 python tools/attic/dbg_outsplit.py 2>&1 | grep "split res -> split out" | tail -1
done

# developer A/B on the GPU box: staggered start of the 256x256 tile's first workgroups (tools/make_variant.py stag<N> gemm_split_big.hip -DLOCOV_BIG_STAGGER=<N>)
for lib in product stag24 stag48 stag96 product; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  echo "== $lib"; timeout 200 python tools/attic/dbg_outsplit.py 2>&1 | grep -E "split res -> split out|conv1 pre" | tail -2
  timeout 200 python tools/attic/dbg_segmean_big.py 2>&1 | grep "big = 1" | tail -1
done

# developer A/B on the GPU box: MFMA order inside an eighth of the 256x256 tile (tools/make_variant.py mfmaord gemm_split_big.hip -DLOCOV_BIG_MFMA_ORDER=1)
for lib in product mfmaord product mfmaord; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  echo "== $lib"; timeout 200 python tools/attic/dbg_outsplit.py 2>&1 | grep -E "split res -> split out|conv1 pre" | tail -2
done

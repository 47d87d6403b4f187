# developer timing on the GPU box: the fused conv1 + input transform with parts of its epilogue ablated (tools/make_variant.py wabl1 / wabl2)
for lib in product wabl1 wabl2; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  echo "== $lib"; timeout 200 python tools/attic/dbg_fuse12.py 2>&1 | tail -2
done

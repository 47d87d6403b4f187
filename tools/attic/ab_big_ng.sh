for lib in product bigng2 bigng4; do
  if [ $lib = product ]; then unset LOCOV_HIP_LIB; else export LOCOV_HIP_LIB=tools/liblocov_$lib.so; fi
  echo "== $lib"; timeout 200 python tools/attic/dbg_outsplit.py 2>&1 | grep -E "conv3 fp32 res -> fp32|conv1 pre" | tail -4
done

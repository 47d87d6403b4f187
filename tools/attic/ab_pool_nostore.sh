# developer timing (round 5): how much of block 0's 2048-channel shortcut pooler is its GATHER?  The same launch
# with its stores removed (tools/liblocov_poolnostore.so, wrong results on purpose) next to the product.  What a pooling of the
# shortcut inside conv3's epilogue could save is bounded by (pooler launch + conv3's residual read) - (this gather at the
# epilogue's occupancy of 8 waves per CU instead of the pooler's 16+).
cd $GRAFT_REPO_ROOT
python3 tools/make_variant.py poolnostore roi_align_nhwc.hip -DLOCOV_POOL_NO_STORE=1 > /dev/null
for rep in 1 2; do
  python3 tools/attic/ab_pool.py 2>&1 | tail -1
  LOCOV_HIP_LIB=tools/liblocov_poolnostore.so python3 tools/attic/ab_pool.py 2>&1 | tail -1
done

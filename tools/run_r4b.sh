export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
: > $O/r4b_ab.log
for v in ${VARIANTS:-ilv0 ilv7 ilv3 ilv4 ilv0 ilv7}; do
  echo "== $v" >> $O/r4b_ab.log
  V3A_LIB=$GRAFT_REPO_ROOT/gpurun_abl/libv3a_$v.so timeout 300 python tools/gemm_sweep.py 6,8,7 ${SHAPES:-0,1,2,3,4,8} 2>&1 | grep "^{" >> $O/r4b_ab.log
  V3A_LIB=$GRAFT_REPO_ROOT/gpurun_abl/libv3a_$v.so timeout 300 python tools/dit_time.py 2>&1 | tail -1 >> $O/r4b_ab.log
done
cat $O/r4b_ab.log

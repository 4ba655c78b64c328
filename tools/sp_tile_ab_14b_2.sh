S="1024x5120x5120 5120x1024x5120"
for r in 1 2; do
  echo "base $(SP_ONLY=4 python tools/sp_rank_time.py 14b 2>/dev/null | tail -1)"
  for t in 20 6; do
    o=""; for s in $S; do o="$o,$s:$t"; done
    echo "tile $t $(V3A_TILE_OVERRIDE=${o#,} SP_ONLY=4 python tools/sp_rank_time.py 14b 2>/dev/null | tail -1)"
  done
done
python tools/gemm_sweep.py 18,19,20,6 "1024x5120x5120r;5120x1024x5120;512x5120x5120r" 2>/dev/null | grep "^{"

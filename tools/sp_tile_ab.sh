#!/bin/bash
# A/B of GEMM tiles inside a sequence-parallel rank's forward (tools/sp_rank_time.py, SP_ONLY=P) through V3A_TILE_OVERRIDE; usage: sp_tile_ab.sh [14b]
M=$1
if [ "$M" = "14b" ]; then
  S="1024x5120x5120 5120x1024x5120"
  for r in 1 2; do
    echo "base $(SP_ONLY=4 python tools/sp_rank_time.py 14b 2>/dev/null | tail -1)"
    for t in 6 4 18 19 20; do
      o=""; for s in $S; do o="$o,$s:$t"; done
      echo "tile $t $(V3A_TILE_OVERRIDE=${o#,} SP_ONLY=4 python tools/sp_rank_time.py 14b 2>/dev/null | tail -1)"
    done
  done
else
  for r in 1 2; do
    echo "base  $(SP_ONLY=4 python tools/sp_rank_time.py 2>/dev/null | tail -1)"
    echo "2-stage small tiles $(V3A_TILE_OVERRIDE=1024x1536x1536:11,1536x1024x1536:11 SP_ONLY=4 python tools/sp_rank_time.py 2>/dev/null | tail -1)"
  done
fi

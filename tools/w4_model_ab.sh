#!/bin/bash
# A/B of the w4 tile inside the full-size DiT forward (tools/dit_time.py): baseline, FFN2 only, FFN2 + the K = 1536 projections
F2="8192x1536x8960:14"
ALL="$F2,8192x1536x1536:14,8192x3072x1536:14"
for r in 1 2; do
  echo "base $(python tools/dit_time.py 2>/dev/null | tail -1)"
  echo "ffn2 $(V3A_TILE_OVERRIDE=$F2 python tools/dit_time.py 2>/dev/null | tail -1)"
  echo "all  $(V3A_TILE_OVERRIDE=$ALL python tools/dit_time.py 2>/dev/null | tail -1)"
done

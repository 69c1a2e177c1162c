#!/bin/bash
# ablation timings of the Winograd weight gradient (needs the ablation build tools/probe/libsedcrnn_abl.so: see DESIGN 5d)
for a in 0 1 2 4 8 12 15; do
  echo "SED_WG_ABL=$a (1 no transforms, 2 no LDS operand reads (+ no transforms), 4 no staging loads, 8 no staging commits, 12 no staging, 15 MFMA + barrier only)"
  SED_WG_ABL=$a SED_CRNN_LIB=$GRAFT_REPO_ROOT/tools/probe/libsedcrnn_abl.so python tools/kbench.py wgrad --iters 20 2>/dev/null | grep "conv3x3_mfma_wgrad B128 T128"
done

#!/bin/bash
# Runs on the GPU box: per-kernel matcher times at 256 frames for image widths around 640 (is the row stride of the cost
# volumes -- 128 bytes a column -- an unlucky one for the memory channels?).  usage: scripts/gpu_width_sweep.sh "608 624 640 656 672"
for w in $1; do echo "== W=$w"; SWEEP_BATCHES=256 timeout -k 10 120 python scripts/dev_sweep.py $w 360 64 5 | tail -1 || exit 1; done

#!/bin/bash
set -e
o=gpurun_out/stamps.txt
: > $o
for args in "fused 256 8 2" "fused 256 375 1" "cols 256 8 2 4 1" "cols 256 375 1 4 1" "cols 256 375 1 16 4" "cols 128 3750 2 4 4" "cols 128 3750 2 4 8" "cols 128 3750 2 16 16"; do
  tools/step_stamps $args >> $o
done
cat $o

#!/bin/bash
# GPU box: sweep-kernel time per forced items-per-lane value (VR_SWEEP_ITEMS) for each tuning library given: tools/tune_items.sh "<items...>" <names...>
items=$1; shift
for name in "$@"; do
  for it in $items; do
    VR_SWEEP_ITEMS=$it tools/tune_sweep.sh $name | sed "s/^/items=$it /" || exit 1
  done
done

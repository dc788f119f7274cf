#!/bin/bash
# tools/segwin_dof.sh -- the queen-like mesh numbered dof by dof (tools/structure_zoo.py: three column clusters a third of the matrix
# apart, nine segments per block of rows) through the segment windows at 16 / 32 tiles per block; experiments build (knobs).
cd "$(dirname "$0")/.."
export SPMV_HIP_EXPERIMENTS=1
for tiles in 32 16 24; do
  for slots in 4096; do
    echo "== SPMV_HIP_SEGWIN_TILES=$tiles SPMV_HIP_SEGWIN_SLOTS=$slots"
    SPMV_HIP_SEGWIN_TILES=$tiles SPMV_HIP_SEGWIN_SLOTS=$slots python3 tools/structure_zoo.py queen_dof_major ragged_4-40_far 2>&1 | grep -v amdgpu.ids
  done
done

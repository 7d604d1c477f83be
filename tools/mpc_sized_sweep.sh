#!/bin/bash
# MPC-sized calls (2 ... 16 episodes per call): lanes, chunk size and the small-launch kernels with two chunks in flight
#   tools/mpc_sized_sweep.sh [quick]
export JMID_LIB=$PWD/safe-interactive-crowdnav_amd/csrc/libjmid_hip_diag.so     # (the experiment knobs)
for m in f16mx f16x3; do
  python tools/single_scene_sweep.py out_traj=0,1,2 $m 1 2>/dev/null | grep ms
  for e in 2 4 8; do
    python tools/single_scene_sweep.py small_lanes=0,1,2 $m $e 2>/dev/null | grep ms
  done
done

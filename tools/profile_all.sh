#!/bin/bash
# Runs on the GPU box: tools/profile_workload.sh for every single-GPU workload of the BASELINE configurations.
# Usage: tools/profile_all.sh <commit>
set -e
C=$1
tools/profile_workload.sh skull_sweep $C
tools/profile_workload.sh c60_sweep $C --volume c60
tools/profile_workload.sh standin_sweep $C --volume standin256
tools/profile_workload.sh skull_iso80 $C --iso-begin 80 --iso-count 1
tools/profile_workload.sh skull4096_iso80 $C --width 4096 --iso-begin 80 --iso-count 1
tools/profile_workload.sh c60_iso100 $C --volume c60 --iso-begin 100 --iso-count 1
tools/profile_workload.sh gyroid $C --volume gyroid1024 --width 4096 --iso-begin 128 --iso-count 32

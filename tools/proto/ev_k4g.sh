#!/bin/bash
# A/B of the hand-written C5 prototypes on one box (next round): k4e = the structure the planner
# emits today, k4g = persistent workgroups + two tiles in flight.  Build both here first
# (hipcc --offload-arch=gfx950 -O3 -o k4e_proto k4e_proto.hip; same for k4g): the binaries travel.
#   gpurun --timeout 200 -- 'bash tools/proto/ev_k4g.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/k4g; mkdir -p $OUT
cd $R/tools/proto
timeout 60 ./k4e_proto 100000000 x > $OUT/k4e.txt 2>&1; tail -2 $OUT/k4e.txt
timeout 90 ./k4g_proto 100000000 > $OUT/k4g.txt 2>&1; tail -12 $OUT/k4g.txt
GRID_PER_CU=4 timeout 90 ./k4g_proto 100000000 > $OUT/k4g_grid4.txt 2>&1; tail -6 $OUT/k4g_grid4.txt

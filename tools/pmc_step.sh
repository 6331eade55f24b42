#!/bin/bash
# rocprofv3 passes for the step kernel at 2^24 boards: kernel-trace stats + three PMC passes (own runs).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
rm -rf /tmp/ps && mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps/kt -- python3 tools/prof_step.py 16777216 10 > gpurun_out/$TAG/kt.log 2>&1
cp /tmp/ps/kt/*/*_kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/ps/p1 -- python3 tools/prof_step.py 16777216 3 > gpurun_out/$TAG/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d /tmp/ps/p2 -- python3 tools/prof_step.py 16777216 3 > gpurun_out/$TAG/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/ps/p3 -- python3 tools/prof_step.py 16777216 3 > gpurun_out/$TAG/p3.log 2>&1
for p in p1 p2 p3; do cp /tmp/ps/$p/*/*_counter_collection.csv gpurun_out/$TAG/$p.csv; done
ls gpurun_out/$TAG

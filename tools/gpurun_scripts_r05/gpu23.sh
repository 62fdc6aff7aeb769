#!/bin/bash
# the device test of the table-driven maps added after the suite run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05w
timeout 200 python -m pytest tests/test_gpu_parity.py -q -n 8 -k "table_driven_maps_on_the_device" > gpurun_out/r05w/tm_test.log 2>&1; tail -5 gpurun_out/r05w/tm_test.log

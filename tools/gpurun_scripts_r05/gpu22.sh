#!/bin/bash
# round 5: the whole device suite on the final sources
export TMPDIR=/tmp; O=gpurun_out/r05x; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); from vkfft_amd import api; print('sources', api.source_hash())" > $O/gpu_suite.log
timeout 1500 python -m pytest tests -m gpu -q -n 8 >> $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log

export TMPDIR=/tmp; O=gpurun_out/r04c; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "every_opfft_table_entry" > $O/opfft_entries.log 2>&1; tail -4 $O/opfft_entries.log

#!/bin/bash
mkdir -p gpurun_out/r04z3
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented" > gpurun_out/r04z3/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04z3/pytest.log

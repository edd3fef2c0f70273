#!/bin/bash
mkdir -p gpurun_out/r04z2
PYTHONPATH=. timeout 70 python tools/gpu_dupkeys.py > gpurun_out/r04z2/dupkeys.log 2>&1; echo "dupkeys rc=$?"; grep -v amdgpu.ids gpurun_out/r04z2/dupkeys.log | tail -16

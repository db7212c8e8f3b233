#!/bin/bash
TAG=${1:-r5gicpb}; shift
mkdir -p gpurun_out/$TAG
timeout 1200 python scripts/r5_gicp_batch_probe.py "$@" > gpurun_out/$TAG/probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/$TAG/probe.txt | tail -20

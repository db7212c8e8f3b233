#!/bin/bash
# round 5: refresh profiles/pmc_traffic.json and profiles/pmc_issue.json on the round's search kernel (previous neighbours as 4-byte
# positions); both files carry the sha256 of icp_grid.hip + icp_grid_device.h they were collected on
TAG=r5pmc
bash scripts/pmc_issue.sh $TAG/issue 200000x200000 50000x50000 > gpurun_out/$TAG.issue.log 2>&1; tail -3 gpurun_out/$TAG.issue.log
bash scripts/pmc_round.sh $TAG 200000x200000 > gpurun_out/$TAG.traffic.log 2>&1; tail -5 gpurun_out/$TAG.traffic.log
ls gpurun_out/$TAG gpurun_out/$TAG/issue 2>/dev/null | head -20

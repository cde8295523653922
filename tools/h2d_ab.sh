#!/bin/bash
# A/B of the host-frame ingest paths (run on the GPU box): fused into the setup kernel (default) vs the copy-stream pipeline
for v in "RMD_HIP_FUSED_INGEST=1" "RMD_HIP_FUSED_INGEST=0"; do echo "== $v"; env $v RMD_HIP_INGEST_PROFILE=1 timeout 100 python tools/h2d_rate.py 2>&1 | tail -8; done

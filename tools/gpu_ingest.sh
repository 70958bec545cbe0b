#!/bin/bash
# Ingest pipeline visit: text -> embedding rate with host tokenization overlapped (tools/probes/ingest_pipeline.py).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
nproc
for cfg in "8192 110 unigram" "8192 110 hash" "8192 110 unigram 512" "16384 110 unigram"; do timeout 600 python tools/probes/ingest_pipeline.py $cfg 2>&1 | tail -2; done | tee gpurun_out/ingest_pipeline.log

#!/bin/bash
# round 6: LayerNorm folding A/B on one layer's GEMMs (diagnostic library) -> gpurun_out/r06_ln_folding_ab.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(./tools/ln_fold_bench 473600 10; ./tools/ln_fold_bench 14848 50 | grep -v "^check") 2>&1 | tee gpurun_out/r06_ln_folding_gpu.log

#!/bin/bash
# parity suite + one interleaved A/B of an environment switch on one box: bash profiles/quick_ab.sh TAG VAR A B [bench args]
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/${tag}_parity.log 2>&1; echo rc=$? >> gpurun_out/${tag}_parity.log
tail -3 gpurun_out/${tag}_parity.log
bash profiles/ab_env.sh "$@" > gpurun_out/${tag}_ab.log 2>&1
tail -4 gpurun_out/${tag}_ab.log

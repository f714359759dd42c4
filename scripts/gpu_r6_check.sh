#!/bin/bash
# round 6: GPU parity suite, then the report kernel A/B (every kernel alone) of the tree against the builds named.
# gpurun --timeout 1200 -- bash scripts/gpu_r6_check.sh [pytest-selection] -- build/exp_r5 ...
ROOT=$(pwd)
mkdir -p gpurun_out
sel=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sel+=("$1"); shift; done
[ "$1" == "--" ] && shift
timeout -k 10 840 python -m pytest tests -m gpu -x -q "${sel[@]}" > gpurun_out/r6_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r6_tests.log
echo "pytest rc $rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
if [ $# -gt 0 ]; then
  bash scripts/gpu_ab_report.sh "$@" 2>&1 | tee gpurun_out/r6_ab.log
fi
exit $rc

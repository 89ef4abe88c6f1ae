#!/bin/bash
# quick GPU visit: parity tests (+ optional pytest -k filter) and smoke.  usage: scripts/gpu_tests.sh TAG [pytest args...]
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -40 | tee $OUT/pytest.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt

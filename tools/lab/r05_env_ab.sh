#!/bin/bash
# A/B of environment settings on one box: r05_env_ab.sh <kind> <rounds> "<env settings A>" "<env settings B>" ...
# (each setting: a blank-separated list of VAR=value; three solves per run, tools/lab/solve_time_probe.py)
set -u
kind=$1; rounds=$2; shift 2
for round in $(seq 1 $rounds); do
  for setting in "$@"; do
    echo "== [$setting] round $round"
    env $setting python3 tools/lab/solve_time_probe.py $kind ${PROBE_N:-1414}
  done
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  set -- $cfg
  echo "== --batch $1 --pipeline $2"
  timeout -k 10 250 bash tools/quick_bench.sh 2 --batch $1 --pipeline $2 || exit 1
done

#!/bin/bash
# sclk / power while a long force loop runs (on the GPU box): tools/watch_clock.sh [force_mode]
MODE=${1:-pair_once}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
PYTHONPATH=. python3 tools/run_sharded.py --bodies 1048576 --steps 80 --energy-every 0 --force-mode $MODE > /tmp/ws_run.txt 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | head -8
  echo --
  sleep 2
done
wait $PID
tail -2 /tmp/ws_run.txt

#!/bin/bash
# build the library (fails loudly), run the CPU lane-logic tests, then hand the rest of the command line to gpurun
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/f2q_build.log 2>&1 || { grep -E "error" -A4 /tmp/f2q_build.log | head -30; exit 1; }
timeout 900 python -m pytest tests/test_lane_logic_cpu.py -x -q > /tmp/f2q_lane.log 2>&1 || { tail -15 /tmp/f2q_lane.log; exit 1; }
tail -1 /tmp/f2q_lane.log
[ $# -gt 0 ] && exec /usr/local/graft/bin/gpurun --timeout 900 -- "$@"

#!/bin/bash
# whole GPU suite, one process; progress goes to a file under gpurun_out/ so that the box's silence watchdog sees it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/tests
mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
rc=$?
tail -15 $O/tests.log
exit $rc

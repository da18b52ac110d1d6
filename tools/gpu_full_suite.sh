#!/bin/bash
# the whole -m gpu suite as the driver runs it, then smoke()
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_suite.log 2>&1
rc=$?
tail -15 gpurun_out/full_suite.log
[ $rc -eq 0 ] && timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
exit $rc

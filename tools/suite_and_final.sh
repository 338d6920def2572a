#!/bin/bash
# round 5: the whole GPU suite with durations, then the final bench lines (tools/final_bench.sh) -- one gpurun call
O=gpurun_out/r5t; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=12 > $O/suite.log 2>&1; rc=$?
tail -22 $O/suite.log
[ $rc = 0 ] || exit $rc
bash tools/final_bench.sh > $O/final.log 2>&1
tail -3 $O/final.log

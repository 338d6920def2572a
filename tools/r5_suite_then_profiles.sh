#!/bin/bash
# round 5: whole GPU suite with durations, then the profile collection (one gpurun call each would also do; the suite gates the profiles)
O=gpurun_out/r5s; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=12 > $O/suite.log 2>&1; rc=$?
tail -22 $O/suite.log
[ $rc = 0 ] || exit $rc
bash tools/collect_profiles_r05.sh > $O/collect.log 2>&1
tail -12 $O/collect.log

#!/bin/bash
# the whole GPU suite with durations and skip reasons, then smoke() and the driver's bench command -- one gpurun call; what the round's final record is made of
# (profiles/r06_suite.md)
O=gpurun_out/r6final; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 -rs > $O/suite.log 2>&1; rc=$?
tail -22 $O/suite.log
[ $rc = 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
tail -c 900 $O/bench_line.json

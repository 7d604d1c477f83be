#!/bin/bash
# the whole -m gpu suite + smoke + the default bench line, one GPU-box call
O=gpurun_out/${R:-r06b}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; echo "gpu tests rc $?" >> $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 600 python bench.py --detail $O/bench_cfg3.json > $O/bench_cfg3.line.json 2> $O/bench_cfg3.stderr.log
tail -15 $O/gputests.log; tail -6 $O/smoke.log; cat $O/bench_cfg3.line.json

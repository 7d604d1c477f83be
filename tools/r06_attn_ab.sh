#!/bin/bash
# round 6, first GPU call: the round-6 softmax of the head-dim-128 attention kernel against the round 2-5 form (tools/attn_check.hip),
# then the parity tests and a short bench on the rebuilt library.
O=gpurun_out/${R:-r06a}; mkdir -p $O
(for m in 0 1 2; do echo "== operand set $m (0 = F16MX, 1 = F16X2, 2 = F16X3)"; timeout 120 build/attn_check 51 1200 20 $m; done
 echo "== keys whose scale grows along the sequence (the reference maximum has to move late): F16MX, F16X3"
 DRIFT=0.5 timeout 120 build/attn_check 8 1200 5 0; DRIFT=0.5 timeout 120 build/attn_check 8 1200 5 2
 echo "== large logits"; timeout 120 build/attn_check 8 1200 5 0 3.0;  timeout 120 build/attn_check 8 1200 5 2 3.0
 echo "== one scene, split-KV 6"; timeout 60 build/attn_check 1 1200 50 0 0.35 6
 echo "== static priority 1 / 2"; PRIO=1 timeout 120 build/attn_check 51 1200 20 0; PRIO=2 timeout 120 build/attn_check 51 1200 20 0
 echo "== one workgroup per CU"; ONE_WG=1 timeout 120 build/attn_check 51 1200 20 0
) 2>&1 | grep -v amdgpu.ids > $O/attn_check.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc $?" >> $O/parity.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-pmc --no-e2e --detail $O/bench_cfg3.json > $O/bench_cfg3.line.json 2> $O/bench_cfg3.stderr.log
tail -3 $O/parity.log; cat $O/attn_check.log; cat $O/bench_cfg3.line.json

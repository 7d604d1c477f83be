# Chunk plan x lanes for MPC-sized batches of cfg2 scenes in f16mx (E episodes per call; --chunk 0 = automatic: two halves side by side): is there a better plan than the default?  bash tools/small_plan_probe.sh on the GPU box -> profiles/r05f_small_plan_probe.log (all within 2 %)
run() { timeout 200 python bench.py --no-pmc --cpu-episodes 0 --no-e2e --episodes-per-gpu $1 --lanes $2 --chunk $3 --modes f16mx --steps 5 --warmup 2 --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('E', $1, 'lanes', $2, 'chunk', $3, d['ms_per_step'], d['value'])"; }
run 3 2 0; run 3 3 1; run 4 2 0; run 4 4 1; run 4 3 0; run 6 2 0; run 6 3 2; run 6 3 0; run 8 2 0; run 8 4 2; run 8 3 0; run 3 2 0; run 4 2 0

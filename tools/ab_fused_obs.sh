#!/bin/bash
# GPU box: the one-launch step (observation at the tail of the env kernel) against the two-launch form, A/B on one box.
# Usage: tools/ab_fused_obs.sh <outdir>
O=${1:-gpurun_out/ab_fused}
mkdir -p $O
python -m pytest tests/test_gpu_env.py -q -x -k "matches_oracle or one_launch or ragged" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for f in 1 0; do
  echo "== PARL_AMD_FUSED_OBS=$f" >> $O/emu.log
  PARL_AMD_FUSED_OBS=$f python tools/emu_bench.py PongNoFrameskip-v4 1024 2>&1 | grep "E=" >> $O/emu.log
  PARL_AMD_FUSED_OBS=$f python tools/emu_bench.py BreakoutNoFrameskip-v4 1024 2>&1 | grep "E=" >> $O/emu.log
  PARL_AMD_FUSED_OBS=$f python bench.py --gpus 1 --steps 20 --warmup 5 --quick --no-cpu-baseline > $O/bench_quick_fused$f.json 2> $O/bench_quick_fused$f.err
  python - <<PY
import json
d = json.load(open('$O/bench_quick_fused$f.json'))
print('fused=$f', 'frames/s %.3f M' % (d['value'] / 1e6), 'updates/s %.0f' % d['learner_updates_per_sec'], 'env_step_ms', d['roofline_env_kernel']['env_step_ms_event_timed'])
PY
done
cat $O/emu.log

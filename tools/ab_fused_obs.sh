#!/bin/bash
# GPU box: the actors' step as conv12 -> trunk GEMM -> ONE env launch (policy head + draw at its head, observation at
# its tail) against the forms with the head (PARL_AMD_FUSED_HEAD=0) and the observation (PARL_AMD_FUSED_OBS=0) as
# launches of their own, A/B on one box.   Usage: tools/ab_fused_obs.sh <outdir> [pytest -k expression]
O=${1:-gpurun_out/ab_fused}
K=${2:-"matches_oracle or one_launch or ragged or head_in_the_env"}
mkdir -p $O
python -m pytest tests/test_gpu_env.py -q -x -k "$K" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  export PARL_AMD_FUSED_OBS=$1 PARL_AMD_FUSED_HEAD=$2
  echo "== PARL_AMD_FUSED_OBS=$1 PARL_AMD_FUSED_HEAD=$2" >> $O/emu.log
  if [ "$2" == "0" ]; then
    python tools/emu_bench.py PongNoFrameskip-v4 1024 2>&1 | grep "E=" >> $O/emu.log
    python tools/emu_bench.py BreakoutNoFrameskip-v4 1024 2>&1 | grep "E=" >> $O/emu.log
  fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --quick --no-cpu-baseline > $O/bench_quick_obs$1_head$2.json 2> $O/bench_quick_obs$1_head$2.err
  python - <<PY
import json
d = json.load(open('$O/bench_quick_obs$1_head$2.json'))
print('obs=$1 head=$2', 'frames/s %.3f M' % (d['value'] / 1e6), 'updates/s %.0f' % d['learner_updates_per_sec'], 'env_step_ms', d['roofline_env_kernel']['env_step_ms_event_timed'])
PY
done
cat $O/emu.log

#!/bin/bash
# GPU box: the heads-loss kernel layouts (PARLHIP_HEADS_KERNEL = 2: two waves per sequence, 4: four waves per
# sequence, 5: four waves + bank-masked DPP adds) — parity tests, event-timed alone / beside the emulator, and the
# rocprofv3 kernel-only time.  Usage: tools/heads_variants.sh [variants...]   -> gpurun_out/heads_variants.log
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
V=${@:-2 4 5}
{
for v in $V; do
  export PARLHIP_HEADS_KERNEL=$v
  echo "===== PARLHIP_HEADS_KERNEL=$v"
  (cd $R && timeout 600 python -m pytest tests/test_gpu_scans.py -q -x -k "heads" 2>&1 | tail -3)
  timeout 120 python $R/tools/heads_loss_time.py
  timeout 300 python $R/tools/heads_beside_env.py
  O=/tmp/prof_hv$v; rm -rf $O
  timeout 300 rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python $R/tools/heads_loss_time.py > $O.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('$O/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'heads' in r['Name']:
            print('   rocprof', r['Name'][:64], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
} 2>&1 | tee $OUT/heads_variants.log

#!/bin/bash
# GPU box: rocprofv3 kernel-only time of the four-wave heads-loss kernel for a list of PARLHIP_HEADS_STAGGER values
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
export PARLHIP_HEADS_KERNEL=${HEADS_KERNEL:-5}
{
for st in "$@"; do
  export PARLHIP_HEADS_STAGGER=$st
  O=/tmp/prof_st$st; rm -rf $O
  timeout 300 rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python $R/tools/heads_loss_time.py > $O.log 2>&1
  grep standalone $O.log
  python - <<PY
import csv, glob
for f in glob.glob('$O/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'heads' in r['Name']:
            print('   stagger $st rocprof', r['Name'][:40], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
} 2>&1 | tee $OUT/heads_stagger.log

#!/bin/bash
# GPU box: rocprofv3 kernel-only time of the heads-loss kernel in the tree library and in diagnostic builds
# (build_exp/hqN.so, see PARLHIP_HQ_ABL).  Usage: tools/heads_ablate.sh tree build_exp/hq1.so ...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
{
for lib in "$@"; do
  if [ "$lib" = "tree" ]; then unset PARL_HIP_LIB; else export PARL_HIP_LIB=$R/$lib; fi
  O=/tmp/prof_abl; rm -rf $O
  timeout 300 rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python $R/tools/heads_loss_time.py > $O.log 2>&1
  echo "== $lib (PARLHIP_HEADS_KERNEL=${PARLHIP_HEADS_KERNEL:-default}): $(grep standalone $O.log)"
  python - <<PY
import csv, glob
for f in glob.glob('$O/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'heads' in r['Name']:
            print('   rocprof', r['Name'][:44], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
} 2>&1 | tee -a $OUT/heads_ablate.log

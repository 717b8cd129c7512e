#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of tools/heads_loss_time.py (the kernel alone). Usage: [tree | build_exp/x.so ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = "tree" ]; then unset PARL_HIP_LIB; else export PARL_HIP_LIB=$R/$lib; fi
  O=/tmp/prof_ha; rm -rf $O
  rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python $R/tools/heads_loss_time.py > $O.log 2>&1
  echo "== $lib"
  python - <<PY
import csv, glob
for f in glob.glob('$O/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'heads' in r['Name']:
            print('  ', r['Name'][:60], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done

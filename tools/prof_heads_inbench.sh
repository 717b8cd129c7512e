#!/bin/bash
# GPU box: rocprofv3 kernel trace of bench.py --quick, print the heads-loss kernel's in-bench durations.
# Usage: tools/prof_heads_inbench.sh [tree | build_exp/x.so ...]   (tree = the in-tree library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = "tree" ]; then unset PARL_HIP_LIB; else export PARL_HIP_LIB=$R/$lib; fi
  O=/tmp/prof_hl; rm -rf $O
  rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python $R/bench.py --quick --no-cpu-baseline --steps 10 > $O.log 2>&1
  echo "== $lib: $(tail -1 $O.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"])')"
  python - <<PY
import csv, glob
for f in glob.glob('$O/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'heads' in r['Name'] or 'atari_env' in r['Name']:
            print('  ', r['Name'][:60], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done

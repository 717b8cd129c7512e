#!/bin/bash
# GPU box: per-kernel table of one learner update (tools/learn84_kernels.py under rocprofv3 --kernel-trace --stats).
# Usage: tools/learn84_prof.sh <out.txt> [84 rows | 42]
R=$GRAFT_REPO_ROOT
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/l84prof
rocprofv3 --kernel-trace --stats -d /tmp/l84prof -o p --output-format csv -- python $R/tools/learn84_kernels.py "$@" > /dev/null 2> /tmp/l84prof.err
python - <<PY
import csv, glob
f = glob.glob('/tmp/l84prof/**/p_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('$OUT', 'w') as o:
    o.write('learn84_kernels.py $@: total kernel time per update %.3f ms (12 updates)\n' % (tot / 12e6))
    for r in rows[:22]:
        o.write('%-70s calls %5s  avg %9.2f us  per update %8.1f us  %5.1f%%\n' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 12e3, float(r['Percentage'])))
print(open('$OUT').read())
PY

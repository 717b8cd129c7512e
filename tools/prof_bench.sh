#!/bin/bash
# GPU box: bench.py plain, then the same command under rocprofv3 --kernel-trace --stats.
# Usage: tools/prof_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,...}
TAG=${1:-bench}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- python $R/bench.py "$@" --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('$OUT/kernel_stats_top.txt', 'w') as f:
    f.write('total kernel time %.2f ms over %d distinct kernels\n' % (tot / 1e6, len(rows)))
    for r in rows[:25]:
        f.write('%-72s calls %7s total %10.3f ms avg %11.2f us %6.2f%%\n' % (r['Name'][:72], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
print(open('$OUT/kernel_stats_top.txt').read())
PY
cat $OUT/bench.json

#!/bin/bash
# GPU box: SQ instruction counters of experimental builds of the env kernel (build_exp/*.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in $R/build_exp/*.so; do
  n=$(basename $so .so)
  export PARL_HIP_LIB=$so
  O=/tmp/exp_$n
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS -d $O -o p --output-format csv -- python $R/tools/emu_bench.py PongNoFrameskip-v4 1024 > $O.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'atari_env' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('$n', {c: round(v / cnt[c] / 1024 / 4) for c, v in agg.items()}, '(per wave per frame)')
PY
  grep "E=1024" $O.log
done

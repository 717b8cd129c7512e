#!/bin/bash
# GPU box: SQ instruction-mix counters of the env-step kernel per wave and emulated frame.
# Usage: tools/pmc_env.sh <out.log> [game ...]   (kernel-trace + --pmc only: no other trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LOG=$1; shift
GAMES=${@:-PongNoFrameskip-v4 BreakoutNoFrameskip-v4}
: > $LOG
for g in $GAMES; do
  O=/tmp/pmc_$g
  rm -rf $O
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $O -o p --output-format csv -- python $R/tools/emu_bench.py $g 1024 > $O.log 2>&1
  grep "E=" $O.log >> $LOG
  python - >> $LOG <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'atari_env' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('PMC $g', {c: round(v / cnt[c] / 1024 / 4) for c, v in agg.items()}, '(per wave per emulated frame; SQ_WAVE_CYCLES / SQ_WAIT_* in units of 4 clocks)')
PY
done
cat $LOG

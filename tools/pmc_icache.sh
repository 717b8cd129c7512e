#!/bin/bash
# GPU box: instruction-cache counters of the env-step kernel (is the 43 % SQ_WAIT_ANY instruction fetch?).
# Usage: tools/pmc_icache.sh <out.log> [game ...]   (kernel-trace + --pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LOG=$1; shift
GAMES=${@:-PongNoFrameskip-v4}
: > $LOG
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_BUSY_CYCLES\|SQ_INSTS_SMEM\|SQ_WAVES" | sort -u | tr '\n' ' ' >> $LOG; echo >> $LOG
for g in $GAMES; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_WAVE_CYCLES"; do
    O=/tmp/pmci_$g
    rm -rf $O
    rocprofv3 --kernel-trace --pmc $set -d $O -o p --output-format csv -- python $R/tools/emu_bench.py $g 1024 > $O.log 2>&1
    tail -2 $O.log | grep -i "error\|invalid" >> $LOG
    python - >> $LOG <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'atari_env' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('PMC $g', {c: round(v / cnt[c] / 1024 / 4) for c, v in agg.items()}, '(kernel total / 1024 waves / 4 frames)')
PY
  done
done
cat $LOG

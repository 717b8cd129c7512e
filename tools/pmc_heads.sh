#!/bin/bash
# GPU box: SQ counters of impala_heads_loss_q_kernel at the workload shape, per wave.  Usage: tools/pmc_heads.sh <out.log>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LOG=${1:-$R/gpurun_out/heads_pmc.log}
: > $LOG
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  O=/tmp/pmc_heads; rm -rf $O
  rocprofv3 --kernel-trace --pmc $set -d $O -o p --output-format csv -- python $R/tools/heads_loss_time.py > $O.log 2>&1
  python - >> $LOG <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'impala_heads_loss_q_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print({c: round(v / cnt[c]) for c, v in agg.items()}, '(per launch, summed over the chip)')
PY
done
cat $LOG

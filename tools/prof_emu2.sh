#!/bin/bash
# GPU box: second PMC set for the env-step kernel (where do parked cycles go). Usage: tools/prof_emu2.sh [E]
cd /tmp && export TMPDIR=/tmp
E=${1:-1024}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_emu2
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $OUT/counters.txt | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG -d $OUT/p3 -o p3 --output-format csv -- python $R/tools/emu_bench.py PongNoFrameskip-v4 $E > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_ACTIVE_INST_EXP_GDS SQ_IFETCH_LEVEL -d $OUT/p4 -o p4 --output-format csv -- python $R/tools/emu_bench.py PongNoFrameskip-v4 $E > $OUT/p4.log 2>&1
python - <<PY
import csv, glob, collections
for p in ('p3','p4'):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
        for k, d in agg.items():
            if 'atari_env' in k:
                print(p, k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
tail -3 $OUT/p3.log; tail -3 $OUT/p4.log
rm -rf $OUT/p3 $OUT/p4

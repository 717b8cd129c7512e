#!/bin/bash
# GPU box: PMC profile of the env-step kernel (dev tool). Usage: tools/prof_emu.sh [E]
cd /tmp && export TMPDIR=/tmp
E=${1:-1024}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_emu
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/p1 -o p1 --output-format csv -- python $R/tools/emu_bench.py PongNoFrameskip-v4 $E > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS -d $OUT/p2 -o p2 --output-format csv -- python $R/tools/emu_bench.py PongNoFrameskip-v4 $E > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for p in ('p1','p2'):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
            cnt[(k, r['Counter_Name'])] += 1
        for k, d in agg.items():
            if 'atari' in k:
                print(p, k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
tail -2 $OUT/p1.log

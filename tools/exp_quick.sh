#!/bin/bash
# GPU box: time every build_exp/*.so on emu_bench (E list = $1), parity for names in $2, PMC for names in $3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ELIST=${1:-1024}
for so in $R/build_exp/*.so; do
  n=$(basename $so .so)
  echo "== $n"; PARL_HIP_LIB=$so timeout 300 python $R/tools/emu_bench.py PongNoFrameskip-v4 $ELIST 2>&1 | grep "E="
done
for n in $2; do
  echo "== parity $n"; PARL_HIP_LIB=$R/build_exp/$n.so timeout 300 python $R/tests/tools/emu_parity.py --envs 8 --steps 300 2>&1 | tail -3
  PARL_HIP_LIB=$R/build_exp/$n.so timeout 300 python $R/tests/tools/emu_parity.py --game BreakoutNoFrameskip-v4 --envs 8 --steps 300 2>&1 | tail -2
done
for n in $3; do
  export PARL_HIP_LIB=$R/build_exp/$n.so
  O=/tmp/exp_$n
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $O -o p --output-format csv -- python $R/tools/emu_bench.py PongNoFrameskip-v4 1024 > $O.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'atari_env' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('PMC $n', {c: round(v / cnt[c] / 1024 / 4) for c, v in agg.items()}, '(per wave per frame)')
PY
done

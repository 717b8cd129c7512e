#!/bin/bash
# GPU box: PC-sampling profile (rocprofv3, host-trap) of the env kernel of ONE library build, reduced on the
# box to a histogram {instruction text / code-object offset: samples} small enough to travel back.
# Usage: tools/pcsample_env.sh <out.json> <lib.so> [game] [interval_us]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$1; LIB=$2; GAME=${3:-PongNoFrameskip-v4}; IV=${4:-20}
O=/tmp/pcs_out
rm -rf $O
PARL_HIP_LIB=$LIB timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${PCS_METHOD:-stochastic} --pc-sampling-unit ${PCS_UNIT:-cycles} \
  --pc-sampling-interval $IV --kernel-trace -d $O -o pcs --output-format csv -- python $R/tools/emu_bench.py $GAME 1024 > $O.log 2>&1
echo "rocprofv3 rc=$?" >> $O.log
tail -5 $O.log
find $O -type f | head -20
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
files = [f for f in glob.glob('/tmp/pcs_out/**/*', recursive=True) if 'pc_sampling' in f and f.endswith('.csv')]
res = {'files': files}
for f in files:
    rd = csv.reader(open(f))
    hdr = next(rd)
    res['header'] = hdr
    hist = collections.Counter()
    n = 0
    sample_rows = []
    for row in rd:
        n += 1
        if n <= 5:
            sample_rows.append(row)
        d = dict(zip(hdr, row))
        key = '|'.join(str(d.get(k, '')) for k in ('Code_Object_Id', 'Code_Object_Offset', 'Instruction', 'Instruction_Comment') if k in d)
        hist[key] += 1
    res['rows'] = n
    res['sample_rows'] = sample_rows
    res['hist'] = dict(hist.most_common(60000))
json.dump(res, open(out, 'w'))
print('rows', res.get('rows'), 'distinct', len(res.get('hist', {})), 'header', res.get('header'))
PY

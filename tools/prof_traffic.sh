#!/bin/bash
# GPU box: HBM traffic of the scan kernels from PMC counters, separate passes for FETCH_SIZE and
# WRITE_SIZE (they do not fit one pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python $R/tools/traffic_probe.py > $OUT/probe.json 2> $OUT/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python $R/tools/traffic_probe.py > /dev/null 2> $OUT/w.err
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(dict)
for tag, ctr in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == ctr:
                res[r['Kernel_Name']].setdefault(ctr, []).append(float(r['Counter_Value']))
probe = json.loads([l for l in open('$OUT/probe.json') if l.startswith('{')][-1])
out = {}
for name, info in probe.items():
    for k, v in res.items():
        if info['kernel'] in k and ('MODE' not in k):
            # take the LAST launch of that kernel instantiation in the run order that matches; kernels
            # used once per probe entry, except gae_chunk (2 passes: sum them)
            fs, ws = v.get('FETCH_SIZE', []), v.get('WRITE_SIZE', [])
            out.setdefault(name, {'algorithmic_read': info['read'], 'algorithmic_write': info['write'], 'kernels': {}})
            out[name]['kernels'][k[:90]] = {'FETCH_SIZE_KB': fs, 'WRITE_SIZE_KB': ws}
json.dump(out, open('$OUT/traffic_raw.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
rm -rf $OUT/f $OUT/w

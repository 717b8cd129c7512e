#!/bin/bash
# GPU box: where the actors' step waits inside the pipeline — rocprofv3 --kernel-trace of the quick bench, reduced on
# the box to the gaps between the consecutive kernels of one env step (emulator -> frame_post -> stack gather -> conv12
# -> fc GEMM -> policy head + draw -> emulator ...).   Usage: tools/actor_gaps.sh <out.txt> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$1; shift
O=/tmp/gaps_prof
rm -rf $O
rocprofv3 --kernel-trace -d $O -o g --output-format csv -- python $R/bench.py --gpus 1 --steps 10 --warmup 3 --quick --no-cpu-baseline "$@" > $O.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/gaps_prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def key(n):
    for k, s in (('env', 'atari_env_kernel'), ('frame_post', 'frame_post_kernel'), ('gather', 'stack_gather_kernel'), ('conv12', 'conv12_u8_mfma_kernel'),
                 ('head', 'policy_head_sample_kernel')):
        if s in n:
            return k
    return None
ev = []
for r in rows:
    k = key(r['Kernel_Name'])
    q = r.get('Queue_Id', r.get('Stream_Id', '0'))
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r['Kernel_Name'][:60], q))
ev.sort()
# the actor queue = the one the env kernel runs on
aq = collections.Counter(e[4] for e in ev if e[2] == 'env').most_common(1)[0][0]
act = [e for e in ev if e[4] == aq]
out = open(sys.argv[1], 'w')
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for a, b in zip(act, act[1:]):
    ka = a[2] or a[3][:30]; kb = b[2] or b[3][:30]
    gaps[(ka, kb)].append((b[0] - a[1]) / 1e3)
for e in act:
    durs[e[2] or e[3][:30]].append((e[1] - e[0]) / 1e3)
print('actor queue', aq, 'kernels', len(act), file=out)
for k, v in sorted(durs.items(), key=lambda x: -sum(x[1]))[:12]:
    print('dur  %-32s n %5d mean %8.1f us total %9.1f ms' % (k, len(v), sum(v) / len(v), sum(v) / 1e3), file=out)
for k, v in sorted(gaps.items(), key=lambda x: -sum(x[1]))[:14]:
    v2 = sorted(v)
    print('gap  %-28s -> %-28s n %5d mean %8.1f us median %8.1f p90 %8.1f total %8.1f ms' % (k[0], k[1], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[int(len(v2) * 0.9)], sum(v) / 1e3), file=out)
out.close()
print(open(sys.argv[1]).read())
PY

"""Dev tool (CPU only): static cost of the translated cartridge split into the hot loops' TRACE copies and the
generic blocks (marker assembly from tools/marker_asm.sh + the oracle's per-address histogram from
tests/tools/oracle_profile.py).  Usage: tools/trace_model.py <atari_env.s> <game.hist> [kernel index] [top]"""
import collections
import re
import sys


def main(asm, histf, kidx=0, top=22):
    lines = open(asm).read().split('\n')
    starts = [i for i, l in enumerate(lines) if ': ; @_ZN7parlhip5atari16atari_env_kernel' in l]
    cnt, kinds, blk = collections.Counter(), collections.defaultdict(collections.Counter), None
    for l in lines[starts[kidx]:]:
        t = l.strip()
        if t.startswith('.Lfunc_end'):   # (not the first s_endpgm: the picture waves of a two-wave kernel leave early)
            break
        m = re.match(r'; @@(BLK|TRC) ([0-9a-f]{4})', t)
        if m:
            blk = (m.group(1), int(m.group(2), 16))
            continue
        if not t or t.startswith((';', '.', '//')) or t.endswith(':') or blk is None:
            continue
        cnt[blk] += 1
        op = t.split()[0]
        k = 'branch' if op.startswith(('s_cbranch', 's_branch', 's_setpc')) else ('lane' if 'lane' in op else (
            'valu' if op.startswith('v_') else ('lds' if op.startswith('ds_') else ('wait' if op.startswith(('s_waitcnt', 's_nop')) else 'salu'))))
        kinds[blk][k] += 1
    hist = {}
    for ln in open(histf):
        pc, c, _ = ln.split()
        hist[int(pc, 16)] = float(c)
    traced = {b[1] for b in cnt if b[0] == 'TRC'}
    tr = sum(cnt[('TRC', pc)] * hist.get(pc, 0) for pc in traced)
    ge = sum(cnt[b] * hist.get(b[1], 0) for b in cnt if b[0] == 'BLK' and b[1] not in traced)
    ni_tr = sum(hist.get(pc, 0) for pc in traced)
    ni_ge = sum(v for pc, v in hist.items() if pc not in traced)
    print('trace blocks: %.0f ISA / frame for %.0f 6507 instructions (%.1f each); generic blocks: %.0f ISA / frame for %.0f (%.1f each)'
          % (tr, ni_tr, tr / max(ni_tr, 1), ge, ni_ge, ge / max(ni_ge, 1)))
    rows = sorted(((cnt[b] * hist.get(b[1], 0), b, cnt[b], hist.get(b[1], 0), dict(kinds[b])) for b in cnt
                   if not (b[0] == 'BLK' and b[1] in traced)), reverse=True)[:top]
    for w, b, c, r, k in rows:
        print('%s %04x runs %6.1f isa %4d weight %7.0f %s' % (b[0], b[1], r, c, w, k))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 22)

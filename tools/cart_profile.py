"""Dev tool (CPU only): static cost model of the translated cartridge code.

Input: the gfx950 assembly of a marker build of atari_env.hip (PARLHIP_CART_MARKERS=1 makes
gen_cart_native.py emit an asm comment at every block start) and a per-address execution histogram
of the game produced by an instrumented build of the CPU oracle (tests/tools/oracle_profile.py).  Output: ISA instructions per 6507
block weighted by how often the block runs = where the translated code spends its issue slots.
Approximate (instructions are attributed to the last marker above them in layout order), used to
pick what to optimise before spending GPU time."""
import collections
import re
import sys


def main(asm, hist, kernel_index=0, top=30):
    kernels, cur = [], None
    for ln in open(asm):
        if ln.startswith('_ZN7parlhip5atari16atari_env_kernel') and ln.rstrip().endswith(':') or ': ; @_ZN7parlhip5atari16atari_env_kernel' in ln:
            cur = []
            kernels.append(cur)
        elif cur is not None:
            cur.append(ln)
            if ln.startswith('.Lfunc_end'):
                cur = None
    lines = kernels[kernel_index]
    counts, kinds = collections.Counter(), collections.defaultdict(collections.Counter)
    blk, traced = None, set()
    for ln in lines:
        t = ln.strip()
        m = re.match(r'; @@(BLK|TRC) ([0-9a-f]{4})', t)
        if m:
            # a block of a hot loop exists twice: as part of the loop's trace (TRC) and as its generic copy (BLK);
            # the model prices the trace copy (an oracle trace shows ~80 % of Pong's iterations stay inside it)
            blk = (m.group(1), int(m.group(2), 16))
            traced.add(blk[1]) if blk[0] == 'TRC' else None
            continue
        if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
            continue
        counts[blk] += 1
        op = t.split()[0]
        k = 'branch' if op.startswith(('s_cbranch', 's_branch', 's_setpc')) else ('lane' if 'lane' in op else (
            'valu' if op.startswith('v_') else ('lds' if op.startswith('ds_') else ('wait' if op.startswith(('s_waitcnt', 's_nop')) else 'salu'))))
        kinds[blk][k] += 1
    c2, k2 = collections.Counter(), collections.defaultdict(collections.Counter)
    for (kind, pc), v in counts.items() if False else [(k, v) for k, v in counts.items() if k is not None]:
        if kind == ('TRC' if pc in traced else 'BLK'):
            c2[pc] = v
            k2[pc] = kinds[(kind, pc)]
    c2[None] = counts.get(None, 0)
    counts, kinds = c2, k2
    dyn = {}
    for ln in open(hist):
        if ln.startswith('R '):
            continue
        pc, cnt = ln.split()[:2]
        dyn[int(pc, 16)] = float(cnt)
    tot = sum(counts.get(pc, 0) * c for pc, c in dyn.items())
    ninstr = sum(dyn.values())
    print('6507 instructions / frame %.0f; modelled ISA instructions / frame %.0f (%.1f per 6507 instruction); '
          'prologue+unattributed static %d' % (ninstr, tot, tot / ninstr, counts.get(None, 0)))
    agg = collections.Counter()
    for pc, c in dyn.items():
        for k, v in kinds.get(pc, {}).items():
            agg[k] += v * c
    print('mix per frame:', {k: round(v) for k, v in agg.most_common()})
    rows = sorted(((counts.get(pc, 0) * c, pc, c) for pc, c in dyn.items()), reverse=True)[:top]
    for w, pc, c in rows:
        print('  %04x  runs %7.1f  isa %4d  weight %8.0f  %s' % (pc, c, counts.get(pc, 0), w, dict(kinds.get(pc, {}))))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 30)

"""Dev tool (CPU only): taken / not-taken counts of every conditional branch of the cartridges from the oracle's
instruction trace (tests/tools/oracle_profile.py [outdir=/tmp/prof] first; and once more with `<outdir> 30000` for the
coverage of a long run, the "executed" list) -> parl_amd/csrc/cart_branch_profile.json,
which gen_cart_native.py turns into branch-probability hints.  Tuning data derived from a run of the user-supplied
cartridges (like a compiler's PGO profile), keyed by the cartridge's CRC-32."""
import collections
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == '__main__':
    prof = sys.argv[1] if len(sys.argv) > 1 else '/tmp/prof'
    out = {}
    for name in ('pong', 'breakout'):
        rom = open(os.path.join(ROOT, 'roms', name + '.bin'), 'rb').read()
        tr = np.fromfile(os.path.join(prof, name + '.trace'), dtype=np.uint16).reshape(-1, 2)
        pcs = tr[:, 0].astype(int)
        nxt = np.roll(pcs, -1)
        isbr = np.array([(rom[p & (len(rom) - 1)] & 0x1f) == 0x10 for p in range(65536)])
        m = isbr[pcs]
        m[-1] = False
        bp, bn = pcs[m], nxt[m]
        taken = bn != ((bp + 2) & 0xffff)
        c = collections.defaultdict(lambda: [0, 0])
        for p, t in zip(bp, taken):
            c[int(p)][int(t)] += 1
        ent = {'game': name, 'branches': {'%04x' % p: [v[0], v[1]] for p, v in sorted(c.items())}}
        # every address an instruction was executed at in a LONG oracle run (tests/tools/oracle_profile.py <outdir> 30000
        # writes <game>.cov): the roots of the translator's recursive descent (gen_cart_native.Cart.discover)
        cov = os.path.join(prof, name + '.cov')
        if os.path.exists(cov):
            rows = [ln.split() for ln in open(cov)]
            rets = [r for r in rows if r[0] == 'R']
            rows = [r for r in rows if r[0] != 'R']
            ent['executed'] = ' '.join(sorted({r[0] for r in rows} | {'%04x' % p for p in set(pcs.tolist())}))
            # RTS / RTI -> where it returned to, with counts (the translator jumps straight to the usual return sites)
            ent['returns'] = ' '.join('%s>%s:%s' % (r[1], r[2], r[3]) for r in sorted(rets))
            # ... and where its JMP () instructions went (lines marked J): dispatch entries, or the interpreter walks on
            # from there until it meets one
            ent['indirect_targets'] = ' '.join(sorted(r[0] for r in rows if r[-1] == 'J'))
        out['%08x' % (zlib.crc32(rom) & 0xffffffff)] = ent
    json.dump(out, open(os.path.join(ROOT, 'parl_amd', 'csrc', 'cart_branch_profile.json'), 'w'), indent=0)
    print({k: (len(v['branches']), len(v.get('executed', '').split())) for k, v in out.items()})

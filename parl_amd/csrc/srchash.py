"""Hash of everything libparl_hip.so is built from: the hand-written sources, the public header, the
cartridge translator and the cartridges it translates.  Used by the Makefile (baked into the library
as parlhip_source_hash) and by tests/test_capi_symbols.py (recomputed from the tree)."""
import glob
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def source_files():
    fs = [f for f in glob.glob(os.path.join(HERE, '*.hip')) + glob.glob(os.path.join(HERE, '*.hpp'))
          if not f.endswith('.gen.hpp')]
    fs += [os.path.join(HERE, 'gen_cart_native.py'), os.path.join(HERE, 'cart_branch_profile.json'),
           os.path.join(HERE, 'Makefile'),
           os.path.join(ROOT, 'include', 'parl_hip.h')]
    fs += [f for f in (os.path.join(ROOT, 'roms', 'pong.bin'), os.path.join(ROOT, 'roms', 'breakout.bin'))
           if os.path.exists(f)]
    return sorted(fs)


def source_hash():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
        h.update(b'\0')
    return h.hexdigest()[:16]


if __name__ == '__main__':
    out = sys.argv[1]
    text = '#define PARLHIP_SOURCE_HASH "%s"\n' % source_hash()
    open(out, 'w').write(text)

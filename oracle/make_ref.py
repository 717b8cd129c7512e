"""Build-container only: collect reference artefacts that must not be committed.

  * roms/{pong,breakout}.bin — the cartridges the reference tree ships as fixtures
    (benchmark/fluid/DQN_variant/rom_files/).  They are user-supplied DATA for the
    emulator (like ALE's ROM import), git-ignored, and travel to the GPU box with the snapshot.
"""
import hashlib
import os
import shutil

REF = '/root/reference/benchmark/fluid/DQN_variant/rom_files'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MD5 = {'pong': '60e0ea3cbe0913d39803477945e9e5ec', 'breakout': 'f34f08e5eb96e500e851a80be3277a56'}

if __name__ == '__main__':
    out = os.path.join(ROOT, 'roms')
    os.makedirs(out, exist_ok=True)
    for name, md5 in MD5.items():
        src = os.path.join(REF, name + '.bin')
        if os.path.exists(src) and hashlib.md5(open(src, 'rb').read()).hexdigest() == md5:
            shutil.copyfile(src, os.path.join(out, name + '.bin'))
            os.chmod(os.path.join(out, name + '.bin'), 0o644)

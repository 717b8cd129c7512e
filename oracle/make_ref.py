"""Build-container only: collect reference artefacts that must not be committed.

  * roms/{pong,breakout}.bin — the cartridges the reference tree ships as fixtures
    (benchmark/fluid/DQN_variant/rom_files/).  They are user-supplied DATA for the
    emulator (like ALE's ROM import), git-ignored, and travel to the GPU box with the snapshot.
  * oracle/_ref/a2c/{train,actor,atari_agent,atari_model,a2c_config}.py — the reference's own torch
    A2C example scripts (benchmark/torch/a2c/), and oracle/_ref/{impala,a2c_paddle}/ — its headline
    examples (examples/IMPALA, examples/A2C: Paddle-flavoured), staged byte for byte so that the GPU box —
    which has no /root/reference — can run them UNMODIFIED through compat/{paddle,parl,gym}
    (tests/test_reference_scripts.py).
  * oracle/_ref/torch_alg/{a2c,atari_model}.py — parl/algorithms/torch/a2c.py and the torch ActorCritic of
    benchmark/torch/a2c/: the reference's learner update, timed on the host cores by bench.py's cpu_baseline
    leg (oracle/ref_torch_baselines.py).  oracle/_ref/ is git-ignored (never in history) but not
    gpurun-ignored, like the built .so files.

Every file is copied ONLY when its bytes differ from what is already there: an unconditional copy
gives the cartridges a new mtime on every build() and make then considers cart_native.gen.hpp and
every object stale.
"""
import hashlib
import os

REF = '/root/reference'
ROM_DIR = os.path.join(REF, 'benchmark/fluid/DQN_variant/rom_files')
A2C_DIR = os.path.join(REF, 'benchmark/torch/a2c')
A2C_SCRIPTS = ['train.py', 'actor.py', 'atari_agent.py', 'atari_model.py', 'a2c_config.py']
# the reference's OWN (Paddle-flavoured) headline examples, run unmodified through compat/{paddle,parl,gym}
EXAMPLES = {'impala': ('examples/IMPALA', ['train.py', 'actor.py', 'atari_agent.py', 'atari_model.py', 'impala_config.py']),
            'a2c_paddle': ('examples/A2C', ['train.py', 'actor.py', 'atari_agent.py', 'atari_model.py', 'a2c_config.py'])}
# the reference's torch A2C algorithm + the torch ActorCritic it trains: the CPU learner baseline of BASELINE.md
# section 3 (bench.py's cpu_baseline leg loads them by path, oracle/ref_torch_baselines.py)
TORCH_ALG = {'parl/algorithms/torch/a2c.py': 'a2c.py', 'benchmark/torch/a2c/atari_model.py': 'atari_model.py'}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MD5 = {'pong': '60e0ea3cbe0913d39803477945e9e5ec', 'breakout': 'f34f08e5eb96e500e851a80be3277a56'}


def copy_if_different(src, dst):
    """-> True when dst was (re)written"""
    data = open(src, 'rb').read()
    if os.path.exists(dst) and open(dst, 'rb').read() == data:
        return False
    tmp = dst + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(data)
    os.chmod(tmp, 0o644)
    os.replace(tmp, dst)
    return True


def main():
    changed = []
    out = os.path.join(ROOT, 'roms')
    os.makedirs(out, exist_ok=True)
    for name, md5 in MD5.items():
        src = os.path.join(ROM_DIR, name + '.bin')
        if os.path.exists(src) and hashlib.md5(open(src, 'rb').read()).hexdigest() == md5:
            if copy_if_different(src, os.path.join(out, name + '.bin')):
                changed.append('roms/%s.bin' % name)
    if os.path.isdir(A2C_DIR):
        out = os.path.join(ROOT, 'oracle', '_ref', 'a2c')
        os.makedirs(out, exist_ok=True)
        for s in A2C_SCRIPTS:
            if copy_if_different(os.path.join(A2C_DIR, s), os.path.join(out, s)):
                changed.append('oracle/_ref/a2c/' + s)
    for name, (sub, files) in EXAMPLES.items():
        src_dir = os.path.join(REF, sub)
        if os.path.isdir(src_dir):
            out = os.path.join(ROOT, 'oracle', '_ref', name)
            os.makedirs(out, exist_ok=True)
            for f in files:
                if copy_if_different(os.path.join(src_dir, f), os.path.join(out, f)):
                    changed.append('oracle/_ref/%s/%s' % (name, f))
    out = os.path.join(ROOT, 'oracle', '_ref', 'torch_alg')
    for src, name in TORCH_ALG.items():
        if os.path.exists(os.path.join(REF, src)):
            os.makedirs(out, exist_ok=True)
            if copy_if_different(os.path.join(REF, src), os.path.join(out, name)):
                changed.append('oracle/_ref/torch_alg/' + name)
    print('make_ref: ' + ('staged ' + ', '.join(changed) if changed else 'everything up to date'))


if __name__ == '__main__':
    main()

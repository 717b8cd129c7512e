"""The C-ABI library must export every symbol include/parl_hip.h declares, and the ctypes
binding table must cover exactly that set.  CPU-only: loads the .so, calls nothing that needs
a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    out = set()
    inc = os.path.join(ROOT, 'include')
    for fn in os.listdir(inc):
        if fn.endswith('.h'):
            src = open(os.path.join(inc, fn)).read()
            src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
            out |= set(re.findall(r'\b(parlhip_\w+)\s*\(', src))
    return out


@pytest.fixture(scope='module')
def built_lib():
    from parl_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _native.LIB_PATH


def test_header_symbols_exported(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 10
    nm = subprocess.check_output(['nm', '-D', '--defined-only', built_lib]).decode()
    exported = set(re.findall(r' T (parlhip_\w+)', nm))
    assert syms <= exported, 'declared but not exported: %s' % sorted(syms - exported)
    assert exported <= syms, 'exported but not declared in include/: %s' % sorted(exported - syms)


def test_ctypes_table_matches_header(built_lib):
    from parl_amd import _native
    assert set(_native.SIGNATURES) == declared_symbols()
    lib = _native.lib()  # resolves every symbol, sets argtypes
    assert lib.parlhip_version() >= 100
    assert lib.parlhip_strerror(-1) == b'invalid argument'
    assert lib.parlhip_adv_normalize_workspace_bytes(1 << 20) >= 16


def test_argument_validation_without_gpu(built_lib):
    """EINVAL paths return before touching the device"""
    from parl_amd import _native
    lib = _native.lib()
    assert lib.parlhip_vtrace_f32(None, None, None, None, None, None, None, None, 5, 4, 1.0, 1.0, None) == -1
    assert lib.parlhip_vtrace_f32(None, None, None, None, None, None, None, None, 0, 4, 1.0, 1.0, None) == 0
    assert lib.parlhip_gae_f32(None, None, None, None, None, None, None, -1, 4, 0.99, 1.0, 0, 0, None) == -1
    assert lib.parlhip_categorical_sample_f32(None, None, None, 4, 0, None) == -1


def test_product_fails_loudly_on_cpu_tensors(built_lib):
    import torch
    from parl_amd import ops, _native
    x = torch.zeros(4, 3)
    with pytest.raises(_native.ParlHipError):
        ops.vtrace(x, x, x, x, x, torch.zeros(3))


def test_no_oracle_in_product():
    """parl_amd/ must never import, load or link anything under oracle/"""
    pkg = os.path.join(ROOT, 'parl_amd')
    bad = re.compile(r'(import\s+oracle|from\s+oracle|c_oracle|libparl_oracle|oracle/_ref|emu_oracle)')
    for d, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.cpp', '.h', 'Makefile')):
                s = open(os.path.join(d, fn)).read()
                assert not bad.search(s), '%s references the oracle' % os.path.join(d, fn)


def test_native_cartridge_translation_is_built_and_tagged(built_lib):
    """When roms/ holds the cartridges, the library is built with their statically translated
    code (csrc/gen_cart_native.py) and parlhip_atari_rom_table_build tags exactly those ROMs; a
    modified ROM is not tagged (it would be interpreted).  Host-only: no GPU needed."""
    import zlib
    import numpy as np
    from parl_amd import _native
    lib = _native.lib()
    for name, game in (('pong', 1), ('breakout', 2)):
        path = os.path.join(ROOT, 'roms', name + '.bin')
        if not os.path.exists(path):
            pytest.skip('no cartridges in roms/')
        rom = np.frombuffer(open(path, 'rb').read(), np.uint8).copy()
        assert lib.parlhip_atari_native_cart(game) == (zlib.crc32(rom.tobytes()) & 0xffffffff)
        table = np.zeros(len(rom), np.uint32)
        assert lib.parlhip_atari_rom_table_build(rom.ctypes.data, len(rom), table.ctypes.data) == 0
        assert int(table[0] >> 28) == game
        assert np.all(table[1:] >> 28 == 0)
        # decode fields untouched by the tag: b1/b2 are the next two ROM bytes
        assert int(table[0] & 0xff) == rom[1] and int((table[0] >> 8) & 0xff) == rom[2]
        rom[100] ^= 0xff
        assert lib.parlhip_atari_rom_table_build(rom.ctypes.data, len(rom), table.ctypes.data) == 0
        assert int(table[0] >> 28) == 0
    assert lib.parlhip_atari_native_cart(0) == 0


def test_library_matches_the_tree(built_lib):
    """build hygiene: the library carries the hash of the sources it was built from
    (csrc/srchash.py); a stale or half-rebuilt .so does not match the tree"""
    import sys
    from parl_amd import _native
    csrc = os.path.join(ROOT, 'parl_amd', 'csrc')
    sys.path.insert(0, csrc)
    try:
        import srchash
    finally:
        sys.path.remove(csrc)
    assert _native.lib().parlhip_source_hash().decode() == srchash.source_hash(), \
        'libparl_hip.so is stale: run python __graft_entry__.py'


@pytest.mark.parametrize('dim,ny,nx', [(42, 5, 5), (84, 3, 3)])
def test_frame_post_tables_have_the_structure_the_observation_tail_assumes(built_lib, dim, ny, nx):
    """parlhip_atari_vec_step_obs converts an env's frame pair band by band (5 source rows) with constant loop
    bounds (csrc/frame_tail.hpp): every output row takes exactly `ny` y taps that stay inside its band, the
    (row inside the band, weight) pairs repeat from band to band bit for bit, a column takes at most `nx` x taps.
    Host function of the product library, no GPU."""
    import numpy as np
    from parl_amd import _native
    lib = _native.lib()
    nb = lib.parlhip_frame_post_tables_bytes(dim)
    blob = np.zeros(nb, np.uint8)
    assert lib.parlhip_frame_post_tables_init(blob.ctypes.data, dim) == 0
    hdr = blob[:32].view(np.int32)
    assert hdr[0] == dim
    xstart = blob[hdr[3]:hdr[3] + 4 * (dim + 1)].view(np.int32)
    ystart = blob[hdr[4]:hdr[4] + 4 * (dim + 1)].view(np.int32)
    yt = blob[hdr[6]:hdr[6] + 8 * int(ystart[dim])].view(np.int32).reshape(-1, 2)   # (si, alpha bits)
    assert (np.diff(ystart) == ny).all()
    assert int(np.diff(xstart).max()) <= nx and int(np.diff(xstart).min()) >= 1
    m = dim // 42
    per_band = yt.reshape(42, m * ny, 2)
    assert (per_band[:, :, 1] == per_band[0, :, 1]).all()                       # the same weights in every band
    assert (per_band[:, :, 0] == per_band[0, :, 0] + 5 * np.arange(42)[:, None]).all()   # the same rows, shifted by the band
    assert per_band[0, :, 0].min() == 0 and per_band[0, :, 0].max() == 4


@pytest.mark.parametrize('dim,ny,nx', [(42, 5, 5), (84, 3, 3)])
def test_frame_post_tables_lane_ordered_copy_equals_the_generic_tables(built_lib, dim, ny, nx):
    """The observation tail reads its x / y taps from a lane-ordered copy right behind the blob's 32-byte header
    (csrc/frame_defs.hpp tail_lane_taps_bytes: Tap[NC][NX][64] then the first band's y taps): it must be the generic
    tables re-ordered — column dx = lane + 64 c, tap k; a tap past the column's count = the column's first source pixel
    with weight 0; a column >= dim = (0, 0).  Host function of the product library, no GPU."""
    import numpy as np
    from parl_amd import _native
    lib = _native.lib()
    nb = lib.parlhip_frame_post_tables_bytes(dim)
    blob = np.zeros(nb, np.uint8)
    assert lib.parlhip_frame_post_tables_init(blob.ctypes.data, dim) == 0
    hdr = blob[:32].view(np.int32)
    xstart = blob[hdr[3]:hdr[3] + 4 * (dim + 1)].view(np.int32)
    ystart = blob[hdr[4]:hdr[4] + 4 * (dim + 1)].view(np.int32)
    xt = blob[hdr[5]:hdr[5] + 8 * int(xstart[dim])].view(np.int32).reshape(-1, 2)   # (si, alpha bits)
    yt = blob[hdr[6]:hdr[6] + 8 * int(ystart[dim])].view(np.int32).reshape(-1, 2)
    nc = 2 if dim > 64 else 1
    assert hdr[3] == 32 + (nc * nx * 64 + 8) * 8                                   # the copy sits between header and xstart
    lane = blob[32:32 + nc * nx * 64 * 8].view(np.int32).reshape(nc, nx, 64, 2)
    for c in range(nc):
        for ln in range(64):
            dx = ln + 64 * c
            for k in range(nx):
                si, al = lane[c, k, ln]
                if dx >= dim:
                    assert (si, al) == (0, 0)
                    continue
                x0, n = int(xstart[dx]), int(xstart[dx + 1] - xstart[dx])
                assert n <= nx
                if k < n:
                    assert (si, al) == tuple(xt[x0 + k])
                else:
                    assert si == xt[x0, 0] and al == 0                              # weight +0.0f
    m = dim // 42
    ylane = blob[32 + nc * nx * 64 * 8:32 + (nc * nx * 64 + 8) * 8].view(np.int32).reshape(8, 2)
    assert (ylane[:m * ny] == yt[:m * ny]).all() and (ylane[m * ny:] == 0).all()


def test_state_blob_constants_python_side_match_the_header():
    """DeviceVectorEnv.running_episode_steps reads MonitorEnv's step counter out of an env's state blob: the byte
    offset of the scalar slots and the slot index are csrc/atari_defs.hpp's"""
    import re
    from parl_amd.env.device_vector_env import DeviceVectorEnv
    src = open(os.path.join(ROOT, 'parl_amd', 'csrc', 'atari_defs.hpp')).read()
    off = int(re.search(r'constexpr int kOffScalars = (\d+);', src).group(1))
    body = re.search(r'enum Slot : int \{(.*?)\};', src, re.S).group(1)
    body = re.sub(r'//[^\n]*', '', body)
    names = [x.strip().split('=')[0].strip() for x in body.split(',') if x.strip()]
    assert DeviceVectorEnv._STATE_SCALARS_OFFSET == off
    assert names[DeviceVectorEnv._SLOT_NUM_STEPS] == 'S_NUM_STEPS'

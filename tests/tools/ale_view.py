"""Dev tool: run the CPU oracle emulator on a ROM and dump frames as PNG (stdlib only)."""
import ctypes
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b''.join(b'\x00' + rgb[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack('>I', len(d)) + t + d
        return c + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def palette():
    pal = (ctypes.c_uint32 * 128)()
    c_oracle.lib().oracle_palette(pal)
    p = np.array(pal, dtype=np.uint32)
    return np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], 1).astype(np.uint8)


class OracleAle:
    def __init__(self, rom_bytes, game):
        L = c_oracle.lib()
        L.oracle_ale_new.restype = ctypes.c_void_p
        L.oracle_ale_act.restype = ctypes.c_int32
        self.L = L
        self.h = ctypes.c_void_p(L.oracle_ale_new(rom_bytes, len(rom_bytes), game))
        self.fb = np.zeros((210, 160), np.uint8)

    def reset(self):
        self.L.oracle_ale_reset(self.h, self.fb.ctypes.data_as(ctypes.c_void_p))

    def act(self, a):
        return self.L.oracle_ale_act(self.h, a, self.fb.ctypes.data_as(ctypes.c_void_p))

    def ram(self):
        out = np.zeros(128, np.uint8)
        self.L.oracle_ale_ram(self.h, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def terminal(self):
        return self.L.oracle_ale_terminal(self.h)

    def jam(self):
        return self.L.oracle_ale_jam(self.h)

    def cycles(self):
        return self.L.oracle_ale_cycles(self.h)


def find_rom(name):
    for d in (os.path.join(ROOT, 'roms'), '/root/reference/benchmark/fluid/DQN_variant/rom_files'):
        p = os.path.join(d, name + '.bin')
        if os.path.exists(p):
            return open(p, 'rb').read()
    raise FileNotFoundError(name)


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'pong'
    nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    game = {'pong': 1, 'breakout': 2}.get(name, 0)
    ale = OracleAle(find_rom(name), game)
    ale.reset()
    pal = palette()
    out = '/tmp/ale_view'
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(0)
    acts = [0, 1, 3, 4, 11, 12] if game == 1 else [0, 1, 3, 4]
    tot = 0
    for i in range(nframes):
        if i % 4 == 0:
            a = acts[rng.integers(len(acts))]
        r = ale.act(a)
        tot += r
        if r:
            print('frame', i, 'reward', r, 'total', tot)
        if i % every == 0 or i == nframes - 1:
            rgb = pal[ale.fb >> 1]
            rgb = np.repeat(np.repeat(rgb, 2, 0), 3, 1)
            write_png('%s/%s_%05d.png' % (out, name, i), rgb)
            print('frame', i, 'cycles/frame', ale.cycles(), 'jam', hex(ale.jam()), 'nonzero px', int((ale.fb != 0).sum()),
                  'ram13/14', ale.ram()[13], ale.ram()[14], 'terminal', ale.terminal())

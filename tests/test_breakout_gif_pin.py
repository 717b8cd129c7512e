"""Pins the emulator oracle (oracle/atari_oracle.c: 6507 + TIA + RIOT + the ALE layer) on the one
ALE artefact the reference tree holds: `.github/Breakout.gif`, 147 frames of a real ALE Breakout
game (mid-game, score 184 -> 331) at exactly 2x the 160x210 ALE screen.  The cartridge in
`benchmark/fluid/DQN_variant/rom_files/breakout.bin` is the program that drew those frames, so
the oracle, running that same cartridge, must draw the same picture for the same game state.

What is compared (tests/golden/breakout_gif_frames.npz = the GIF at native resolution, see
make_breakout_gif_golden.py; the GIF is dithered onto a 3-3-2 colour cube, hence colour
tolerances, while GEOMETRY is compared exactly):

* for EVERY frame the game state visible in the picture (36 brick bytes, the five score /
  lives / player digit glyph pointers) is written into the oracle's RAM and one frame is
  emulated: walls, corner blocks, digit glyphs and brick cells must coincide pixel for pixel
  (asymmetric-playfield timing of the brick kernel, the 6-digit score kernel, YStart = 34);
* colours: every distinct region (wall, six brick rows, corner blocks, paddle, ball, digits)
  within the dither error of the oracle's NTSC palette entry;
* paddle: a double-size player.  Its rows, its widths, the left stop (x = 8, all 16 pixels
  visible), the right cartridge stop (x = 144) and the lattice of settled x positions — 8, 15,
  27, 37, 49, 61, 73, 83, 95, 107, 119, 129, 141: one step = 2 ALE frames of the ALE paddle
  delta (2 x 23000) counted from the maximum-resistance stop (790196) — must be what the oracle
  produces when driven through its own ALE paddle layer (resistance range and delta, INPT0
  charge timing after the dump is released, the cartridge's scanline polling, RESP0 / HMOVE
  positioning): they agree to the pixel.  THIS TEST FOUND A REAL
  BUG: the oracle (and the HIP emulator) drew double / quad size players one pixel too far
  left (Stella 2.x delays them by one pixel); both were fixed in the commit that added it;
* ball: 2 x 4 pixels.

* the ALE game layer (oracle/atari_oracle.c ale_rom_step: reward / lives / terminal decoded from
  cartridge RAM, what EpisodicLifeEnv and the reward stream depend on — parl/env/atari_wrappers.py:
  186-211, benchmark/fluid/DQN_variant/atari.py:134-165): the recording shows the end of one game
  (184, 191) and then a WHOLE new game from "000 5 1" to 331 with the lives digit going 5 -> 1.
  For every frame the score the picture shows is written as BCD into RAM 76 / 77 and the lives into
  RAM 57 — the bytes the ALE layer decodes — with the five glyph pointers poisoned; the cartridge
  itself turns those bytes into glyph pointers and draws them: the five digit slots must equal the
  recording pixel for pixel, and the oracle's ALE layer must report exactly that score delta as
  the reward, that lives count, and terminal only at lives 0.  So the decode reads the same RAM
  cells, nibbles and digit order the cartridge displays (and the recording shows);
* the reward stream against the bricks: between consecutive frames of the recording
  the score grows by exactly the row values (1, 1, 4, 4, 7, 7 from the bottom row) of the brick
  cells that vanished from the picture.

What this does not pin: frame-by-frame dynamics (the GIF keeps roughly every 30th ALE frame,
consecutive GIF frames are not consecutive emulator frames), sound.  PONG HAS NO REFERENCE-HELD
GROUND TRUTH AT ALL: the reference tree holds no picture, recording or known-answer vector of
the Pong cartridge — only its ROM bytes.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
from oracle import c_oracle  # noqa: E402
import ale_view  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'breakout_gif_frames.npz')
BRIGHT = 110  # sum of RGB above which a de-dithered pixel is "lit" (darkest game colour sums to 338)


def _rom():
    try:
        return ale_view.find_rom('breakout')
    except FileNotFoundError:
        pytest.skip('breakout cartridge not provisioned (roms/)')


class Surgeon:
    """An oracle Breakout game in flight whose RAM can be overwritten before rendering a frame."""

    def __init__(self):
        self.L = c_oracle.lib()
        self.ale = ale_view.OracleAle(_rom(), 2)
        self.ale.reset()
        for i in range(80):
            self.ale.act(1 if i < 8 else 0)
        self.base = self.ale.ram().copy()
        self.pal = ale_view.palette()

    def render(self, ram):
        self.L.oracle_ale_set_ram(self.ale.h, ram.ctypes.data_as(ctypes.c_void_p))
        self.ale.act(0)
        return self.ale.fb.copy()


def brick_ram_from_picture(lit):
    """36 brick bytes (RAM 0..35) from a frame's lit mask.  Layout found by flipping every bit of
    the oracle's RAM and looking at which 4 x 6 pixel cell changes (the cartridge defines it):
    row r (0 = bottom, y = 87-6r .. 92-6r), 36 playfield cells k of 4 pixels from x = 8."""
    ram = np.zeros(36, np.uint8)
    for r in range(6):
        for k in range(36):
            if lit[87 - 6 * r:93 - 6 * r, 8 + 4 * k:12 + 4 * k].mean() > 0.5:
                if k < 2:
                    b, bit = 30 + r, 6 + k
                elif k < 10:
                    b, bit = 24 + r, 7 - (k - 2)
                elif k < 18:
                    b, bit = 18 + r, k - 10
                elif k < 22:
                    b, bit = 12 + r, 4 + (k - 18)
                elif k < 30:
                    b, bit = 6 + r, 7 - (k - 22)
                else:
                    b, bit = r, k - 30
                ram[b] |= 1 << bit
    return ram


# digit slots of the score kernel: (glyph pointer low byte in RAM, first column, last column)
SLOTS = [(80, 36, 47), (82, 52, 63), (84, 68, 79), (86, 100, 111), (88, 132, 143)]


@pytest.fixture(scope='module')
def gif():
    return np.load(GOLD)['frames']


@pytest.fixture(scope='module')
def surgeon():
    return Surgeon()


@pytest.fixture(scope='module')
def digit_templates(surgeon):
    """lit mask of rows 0..16 of every slot for every digit, drawn by the oracle"""
    t = {}
    for d in range(10):
        ram = surgeon.base.copy()
        for a, _, _ in SLOTS:
            ram[a] = 5 * d  # glyph tables hold 5 bytes per digit (cartridge data)
        lit = surgeon.render(ram)[0:17] != 0
        for a, x0, x1 in SLOTS:
            t[(a, d)] = lit[:, x0:x1 + 1]
    return t


def _ball_box(lit):
    sub = lit[32:189, 8:152].copy()
    sub[57 - 32:93 - 32] = False
    ys, xs = np.where(sub)
    if len(ys) == 0:
        return None
    return ys.min() + 32, ys.max() + 32, xs.min() + 8, xs.max() + 8


def test_every_gif_frame_is_redrawn_by_the_oracle(gif, surgeon, digit_templates):
    pal = surgeon.pal
    worst_colour = 0.0
    for f in range(gif.shape[0]):
        g = gif[f].astype(np.float32)
        lit = g.sum(-1) > BRIGHT
        ram = surgeon.base.copy()
        ram[:36] = brick_ram_from_picture(lit)
        for a, x0, x1 in SLOTS:  # read the digits with the oracle's own glyphs
            errs = [int((digit_templates[(a, d)] != lit[0:17, x0:x1 + 1]).sum()) for d in range(10)]
            d = int(np.argmin(errs))
            assert errs[d] == 0, 'frame %d slot %d: no oracle glyph matches (best %d wrong pixels)' % (f, a, errs[d])
            ram[a] = 5 * d
        fb = surgeon.render(ram)
        olit = fb != 0
        diff = lit != olit
        diff[189:193, 8:152] = False  # the paddle is compared separately (its x is an input, not state)
        bb = _ball_box(lit)
        if bb is not None:
            diff[max(bb[0] - 1, 0):bb[1] + 2, max(bb[2] - 1, 0):bb[3] + 2] = False
        ob = _ball_box(olit)
        if ob is not None:
            diff[ob[0]:ob[1] + 1, ob[2]:ob[3] + 1] = False
        # the ball may be crossing the brick band, where it cannot be told from brick pixels
        n = int(diff.sum())
        assert n <= 8, 'frame %d: %d pixels differ outside paddle / ball' % (f, n)
        assert not diff[:57].any() and not diff[93:].any(), 'frame %d: walls / digits differ' % f
        # colours of everything both pictures light up
        both = lit & olit
        both[189:193, 8:152] = False
        orgb = pal[fb >> 1].astype(np.float32)
        for c in np.unique(fb[both]):
            sel = both & (fb == c)
            if sel.sum() < 24:
                continue
            err = np.abs(g[sel].mean(0) - orgb[sel][0]).max()
            worst_colour = max(worst_colour, float(err))
            assert err <= 14.0, 'frame %d colour %d: gif %s oracle %s' % (f, c, g[sel].mean(0), orgb[sel][0])
    assert worst_colour > 0.0


def _paddle(lit):
    xs = np.where(lit[189:193, 8:152].any(0))[0] + 8
    return (int(xs.min()), int(xs.max())) if len(xs) else None


def test_paddle_geometry_and_positions_match_the_ale_paddle_layer(gif):
    # --- what the recording shows
    seen = []
    for f in range(gif.shape[0]):
        lit = gif[f].astype(np.float32).sum(-1) > BRIGHT
        assert not lit[186:189, 8:152].any() and not lit[193:197, 8:152].any() or _ball_box(lit) is not None
        p = _paddle(lit)
        if p is None:
            continue
        rows = np.where(lit[189:193, p[0]:p[1] + 1].all(1))[0]
        assert len(rows) == 4, 'paddle is 4 lines tall (rows 189..192)'
        seen.append(p)
    full = sorted({p[0] for p in seen if p[1] - p[0] + 1 == 16})
    assert min(full) == 8 and (8, 23) in seen           # left stop: all 16 pixels visible from x = 8
    assert max(p[0] for p in seen) == 144                 # right stop of the cartridge
    assert {p[1] - p[0] + 1 for p in seen if 8 < p[0] and p[1] < 151} <= {16, 12}
    # --- what the oracle does, driven through its own ALE paddle layer
    ale = ale_view.OracleAle(_rom(), 2)
    ale.reset()
    for _ in range(10):
        ale.act(1)
    for _ in range(80):
        ale.act(4)  # LEFT to the maximum-resistance stop (80: the served ball has left the paddle rows)
    assert _paddle(ale.fb != 0) == (8, 23)
    # settled positions: 2 frames of RIGHT (2 x the ALE paddle delta of 23000), then NOOPs until the
    # cartridge's own smoothing of the paddle has converged
    lattice = []
    for _ in range(15):
        lattice.append(_paddle(ale.fb != 0)[0])
        ale.act(3)
        ale.act(3)
        for _ in range(8):
            ale.act(0)
    assert lattice[:14] == [8, 15, 27, 37, 49, 61, 73, 83, 95, 107, 119, 129, 141, 144] and lattice[14] == 144
    lattice = np.array(lattice[:14])
    # the recording: 141 (clipped to 141..151 by the wall) is the last lattice point before the stop
    assert sum(1 for p in seen if p == (141, 151)) >= 5
    d = np.array([np.abs(lattice - x).min() for p in seen for x in [p[0]] if p[1] - p[0] + 1 == 16])
    assert (d == 0).mean() >= 0.45, 'settled paddle positions of the recording sit exactly on the oracle lattice'
    assert (d <= 2).mean() >= 0.98, 'the rest are the cartridge\'s smoothing transients (1-2 pixels behind)'


def test_ball_is_two_by_four(gif):
    sizes = {}
    for f in range(gif.shape[0]):
        lit = gif[f].astype(np.float32).sum(-1) > BRIGHT
        bb = _ball_box(lit)
        if bb is not None:
            k = (int(bb[3] - bb[2] + 1), int(bb[1] - bb[0] + 1))
            sizes[k] = sizes.get(k, 0) + 1
    assert max(sizes, key=sizes.get) == (2, 4) and sizes[(2, 4)] > 100
    assert all(w == 2 and h <= 4 for (w, h) in sizes), sizes  # shorter only when clipped by the brick band
    ale = ale_view.OracleAle(_rom(), 2)
    ale.reset()
    for i in range(60):
        ale.act(1 if i < 8 else 0)
    bb = _ball_box(ale.fb != 0)
    assert (bb[3] - bb[2] + 1, bb[1] - bb[0] + 1) == (2, 4)


def _digits(gif, digit_templates):
    """per frame the five digits the recording shows (exact glyph matches only)"""
    out = []
    for f in range(gif.shape[0]):
        lit = gif[f].astype(np.float32).sum(-1) > BRIGHT
        row = []
        for a, x0, x1 in SLOTS:
            errs = [int((digit_templates[(a, d)] != lit[0:17, x0:x1 + 1]).sum()) for d in range(10)]
            d = int(np.argmin(errs))
            assert errs[d] == 0
            row.append(d)
        out.append(row)
    return np.array(out)


def test_ale_score_lives_decode_is_what_the_cartridge_displays(gif, surgeon, digit_templates):
    digs = _digits(gif, digit_templates)
    score = digs[:, 0] * 100 + digs[:, 1] * 10 + digs[:, 2]
    lives = digs[:, 3]
    # the recording: the tail of one game, then a whole game from 0 with 5 lives
    assert list(score[:3]) == [184, 191, 0] and lives[2] == 5 and score[-1] == 331 and lives[-1] == 1
    assert (np.diff(score[2:]) >= 0).all() and (np.diff(lives[2:]) <= 0).all() and set(lives) == {1, 2, 3, 4, 5}
    assert (digs[:, 4] == 1).all()  # player number
    L = surgeon.L
    prev = None
    for f in range(gif.shape[0]):
        lit = gif[f].astype(np.float32).sum(-1) > BRIGHT
        ram = surgeon.base.copy()
        ram[:36] = brick_ram_from_picture(lit)
        ram[77] = ((score[f] // 10) % 10) << 4 | (score[f] % 10)   # ALE Breakout: ones / tens nibbles of RAM 77,
        ram[76] = (score[f] // 100) % 10                            # hundreds in the low nibble of RAM 76
        ram[57] = lives[f]                                          # lives in RAM 57
        for a, _, _ in SLOTS:
            ram[a] = 5 * 8 if a != 88 else ram[a]                   # poison the glyph pointers (an "8" everywhere)
        L.oracle_ale_set_ram(surgeon.ale.h, ram.ctypes.data_as(ctypes.c_void_p))
        reward = surgeon.ale.act(0)
        after = surgeon.ale.ram()
        # the cartridge derived the glyph pointers from the bytes the ALE layer decodes ...
        assert [int(after[a]) // 5 for a, _, _ in SLOTS[:4]] == list(digs[f, :4]), f
        surgeon.ale.act(0)
        olit = surgeon.ale.fb != 0
        # ... and draws the recording's digits with them, pixel for pixel
        for a, x0, x1 in SLOTS:
            assert np.array_equal(olit[0:17, x0:x1 + 1], lit[0:17, x0:x1 + 1]), (f, a)
        # and the ALE layer reads the same bytes the same way: reward = score delta, lives, terminal
        if prev is not None:
            assert reward == score[f] - prev, (f, reward, score[f], prev)
        prev = score[f]
        assert L.oracle_ale_lives(surgeon.ale.h) == lives[f]
        assert L.oracle_ale_terminal(surgeon.ale.h) == 0
    # game over is lives == 0 once the game has started (5 lives seen)
    ram = surgeon.base.copy()
    ram[57] = 0
    L.oracle_ale_set_ram(surgeon.ale.h, ram.ctypes.data_as(ctypes.c_void_p))
    surgeon.ale.act(0)
    assert L.oracle_ale_terminal(surgeon.ale.h) == 1 and L.oracle_ale_lives(surgeon.ale.h) == 0


ROW_VALUE = [1, 1, 4, 4, 7, 7]  # bottom row first (Breakout's scoring); a brick is TWO playfield cells (8 pixels) wide


def test_score_grows_by_the_value_of_the_vanished_bricks(gif, digit_templates):
    digs = _digits(gif, digit_templates)
    score = digs[:, 0] * 100 + digs[:, 1] * 10 + digs[:, 2]

    def cells(f):
        lit = gif[f].astype(np.float32).sum(-1) > BRIGHT
        return np.array([[lit[87 - 6 * r:93 - 6 * r, 8 + 4 * k:12 + 4 * k].mean() > 0.5 for k in range(36)]
                         for r in range(6)])

    start = 2  # the new game ("000 5 1"): a full wall
    c0 = cells(start)
    assert c0.all()
    lagging = []
    prev = c0
    for f in range(start + 1, gif.shape[0]):
        c = cells(f)
        assert not (c & ~prev).any(), 'frame %d: a brick came back' % f
        want = sum(ROW_VALUE[r] * int((c0[r] & ~c[r]).sum()) for r in range(6)) / 2
        lag = want - (score[f] - score[start])
        # a brick disappears from the picture in the frame of the hit; the cartridge adds its value to the score
        # a moment later: a recorded frame may fall in between (score short by exactly that ONE brick)
        assert lag == 0 or lag in ROW_VALUE, (f, score[f], want)
        if lag:
            lagging.append(f)
        prev = prev & c
    assert len(lagging) <= 5 and all(b - a > 1 for a, b in zip(lagging, lagging[1:])), lagging
    assert gif.shape[0] - 1 not in lagging and score[-1] - score[start] == 331

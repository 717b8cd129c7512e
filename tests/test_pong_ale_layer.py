"""Pong has NO reference-held ground truth: the reference tree holds no picture, recording or
known-answer vector of the Pong cartridge, only its ROM bytes
(benchmark/fluid/DQN_variant/rom_files/pong.bin).  What can be checked without one is that the oracle's ALE
game layer for Pong (oracle/atari_oracle.c ale_rom_step: cpu score = RAM 13, player score = RAM 14, reward =
delta of (player - cpu), terminal at 21, no lives; ALE's Pong.cpp from memory) reads the cells the CARTRIDGE
ITSELF treats as the two scores, the right way round:

  * RAM 13 is drawn as the LEFT score in the colour of the left paddle, RAM 14 as the RIGHT score in the
    colour of the right paddle (the cartridge draws each score in its paddle's colour);
  * the right paddle is the one that answers the agent's actions (Video Olympics swaps the paddles: ALE's
    paddle A drives INPT1), UP for ALE action RIGHT (3), DOWN for LEFT (4); the left one is the cartridge's
    own player — so RAM 14 is the agent's score and RAM 13 the opponent's;
  * reward / terminal follow those cells: +1 per player point, -1 per opponent point, game over at 21.

A self-consistency pin of the oracle against the cartridge it runs, not a comparison with ALE."""
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


@pytest.fixture(scope='module')
def game():
    try:
        rom = ale_view.find_rom('pong')
    except FileNotFoundError:
        pytest.skip('pong cartridge not provisioned (roms/)')
    ale = ale_view.OracleAle(rom, 1)
    ale.reset()
    for _ in range(120):
        ale.act(0)
    return ale


def _set(ale, **cells):
    ram = ale.ram().copy()
    for k, v in cells.items():
        ram[int(k[1:])] = v
    c_oracle.lib().oracle_ale_set_ram(ale.h, ram.ctypes.data_as(ctypes.c_void_p))


def _paddle(fb, colour):
    ys, xs = np.where(fb[34:194] == colour)
    return (int(ys.min()) + 34, int(xs.min())) if len(ys) else None


def test_pong_scores_are_the_cells_the_cartridge_displays(game):
    ale, L = game, c_oracle.lib()
    fb = ale.fb.copy()
    field = fb[34:194]
    bg = int(np.bincount(field.ravel()).argmax())
    cols = {}
    for c in np.unique(field):
        if c != bg:
            ys, xs = np.where(field == c)
            if len(ys) == 64:  # a 4 x 16 paddle
                cols['left' if xs.min() < 80 else 'right'] = int(c)
    assert set(cols) == {'left', 'right'} and cols['left'] != cols['right']
    # which paddle the agent moves: 8 steps of RIGHT (up), then LEFT (down)
    y0 = {k: _paddle(ale.fb, c)[0] for k, c in cols.items()}
    for _ in range(12):
        ale.act(3)
    y_up = _paddle(ale.fb, cols['right'])[0]
    for _ in range(24):
        ale.act(4)
    y_dn = _paddle(ale.fb, cols['right'])[0]
    assert y_up < y0['right'] < y_dn, 'the right paddle answers the agent: RIGHT = up, LEFT = down'

    # the score cells: poke one, see which digits change and in whose colour
    def shot(r13, r14):
        _set(ale, r13=r13, r14=r14)
        rew = ale.act(0)
        ale.act(0)
        return ale.fb.copy(), rew

    f00, _ = shot(0, 0)
    f50, _ = shot(5, 0)
    f05, _ = shot(0, 5)
    top = slice(0, 30)
    d13, d14 = np.argwhere(f00[top] != f50[top]), np.argwhere(f00[top] != f05[top])
    assert len(d13) and d13[:, 1].max() < 80, 'RAM 13 is the LEFT score'
    assert len(d14) and d14[:, 1].min() >= 80, 'RAM 14 is the RIGHT score'
    assert set(np.unique(f50[top][f00[top] != f50[top]])) <= {bg, cols['left']}
    assert set(np.unique(f05[top][f00[top] != f05[top]])) <= {bg, cols['right']}
    # reward / terminal decode of exactly those cells
    _set(ale, r13=3, r14=3)
    ale.act(0)
    _set(ale, r13=3, r14=4)
    assert ale.act(0) == 1 and L.oracle_ale_terminal(ale.h) == 0 and L.oracle_ale_lives(ale.h) == 0
    _set(ale, r13=4, r14=4)
    assert ale.act(0) == -1
    _set(ale, r13=4, r14=21)
    assert ale.act(0) == 17 and L.oracle_ale_terminal(ale.h) == 1
    _set(ale, r13=21, r14=4)
    ale.act(0)
    assert L.oracle_ale_terminal(ale.h) == 1

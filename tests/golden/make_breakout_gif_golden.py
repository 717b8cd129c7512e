"""Decode the reference's own ALE recording (/root/reference/.github/Breakout.gif: 147 frames,
320x420 = the 160x210 ALE screen doubled, Floyd-Steinberg dithered onto a 3-3-2 colour cube)
into native-resolution frames: every 2x2 block of the GIF is one ALE pixel, the block mean
removes most of the dither.  Build-container only (reads /root/reference); the output
tests/golden/breakout_gif_frames.npz is committed so the pin test runs anywhere.

    python tests/golden/make_breakout_gif_golden.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/.github/Breakout.gif'

if __name__ == '__main__':
    im = Image.open(SRC)
    assert im.size == (320, 420) and im.n_frames == 147
    frames = []
    for i in range(im.n_frames):
        im.seek(i)
        g = np.asarray(im.convert('RGB'), dtype=np.float32)
        frames.append(np.rint(g.reshape(210, 2, 160, 2, 3).mean(axis=(1, 3))).astype(np.uint8))
    frames = np.stack(frames)
    out = os.path.join(HERE, 'breakout_gif_frames.npz')
    np.savez_compressed(out, frames=frames, source=np.array(SRC))
    print(out, frames.shape, os.path.getsize(out), 'bytes')

/*
 * atari_oracle.c — plain-C Atari 2600 (6507 + TIA + RIOT) and the ALE layer above it.
 * TEST INFRASTRUCTURE ONLY.  See atari_oracle.h for what is restated and from where, and for
 * the "parity unpinned against ALE" statement.
 *
 * Formulation: deliberately the simplest one — the CPU ticks one bus cycle at a time and the
 * TIA is advanced one colour clock (one pixel) at a time up to the clock of each register
 * access.  The HIP emulator (parl_amd/csrc/atari_core.hpp + atari_env.hip) uses a different formulation
 * (whole-instruction cycle counts, 64-lane segment rendering with bit masks); the two must
 * agree bit-for-bit on frame buffers, RAM and rewards.
 */
#include "atari_oracle.h"
#include <string.h>

/* Stella 2.x NTSC palette (TIA colour byte >> 1 -> 0xRRGGBB), as used by ALE's getScreenRGB. */
const uint32_t atari_ntsc_palette[128] = {
  0x000000, 0x4a4a4a, 0x6f6f6f, 0x8e8e8e, 0xaaaaaa, 0xc0c0c0, 0xd6d6d6, 0xececec,
  0x484800, 0x69690f, 0x86861d, 0xa2a22a, 0xbbbb35, 0xd2d240, 0xe8e84a, 0xfcfc54,
  0x7c2c00, 0x904811, 0xa26221, 0xb47a30, 0xc3903d, 0xd2a44a, 0xdfb755, 0xecc860,
  0x901c00, 0xa33915, 0xb55328, 0xc66c3a, 0xd5824a, 0xe39759, 0xf0aa67, 0xfcbc74,
  0x940000, 0xa71a1a, 0xb83232, 0xc84848, 0xd65c5c, 0xe46f6f, 0xf08080, 0xfc9090,
  0x840064, 0x97197a, 0xa8308f, 0xb846a2, 0xc659b3, 0xd46cc3, 0xe07cd2, 0xec8ce0,
  0x500084, 0x68199a, 0x7d30ad, 0x9246c0, 0xa459d0, 0xb56ce0, 0xc57cee, 0xd48cfc,
  0x140090, 0x331aa3, 0x4e32b5, 0x6848c6, 0x7f5cd5, 0x956fe3, 0xa980f0, 0xbc90fc,
  0x000094, 0x181aa7, 0x2d32b8, 0x4248c8, 0x545cd6, 0x656fe4, 0x7580f0, 0x8490fc,
  0x001c88, 0x183b9d, 0x2d57b0, 0x4272c2, 0x548ad2, 0x65a0e1, 0x75b5ef, 0x84c8fc,
  0x003064, 0x185080, 0x2d6d98, 0x4288b0, 0x54a0c5, 0x65b7d9, 0x75cceb, 0x84e0fc,
  0x004030, 0x18624e, 0x2d8169, 0x429e82, 0x54b899, 0x65d1ae, 0x75e7c2, 0x84fcd4,
  0x004400, 0x1a661a, 0x328432, 0x48a048, 0x5cba5c, 0x6fd26f, 0x80e880, 0x90fc90,
  0x143c00, 0x355f18, 0x527e2d, 0x6e9c42, 0x87b754, 0x9ed065, 0xb4e775, 0xc8fc84,
  0x303800, 0x505916, 0x6d762b, 0x88923e, 0xa0ab4f, 0xb7c25f, 0xccd86e, 0xe0ec7c,
  0x482c00, 0x694d14, 0x866a26, 0xa28638, 0xbb9f47, 0xd2b656, 0xe8cc63, 0xfce070,
};

/* ===================================================================================== */
/* TIA                                                                                   */
/* ===================================================================================== */
#define HBLANK 68
#define CLOCKS_PER_LINE 228

enum { CX_M0P1 = 1 << 0, CX_M0P0 = 1 << 1, CX_M1P0 = 1 << 2, CX_M1P1 = 1 << 3, CX_P0PF = 1 << 4,
       CX_P0BL = 1 << 5, CX_P1PF = 1 << 6, CX_P1BL = 1 << 7, CX_M0PF = 1 << 8, CX_M0BL = 1 << 9,
       CX_M1PF = 1 << 10, CX_M1BL = 1 << 11, CX_BLPF = 1 << 12, CX_P0P1 = 1 << 13,
       CX_M0M1 = 1 << 14 };

static inline int frame_clock0(const Atari* a) { return a->cyc0 * 3; }

/* copy start offsets per NUSIZ mode (players and missiles); -1 terminates */
static const int8_t k_copies[8][3] = {{0, -1, -1}, {0, 16, -1}, {0, 32, -1}, {0, 16, 32},
                                      {0, 64, -1}, {0, -1, -1}, {0, 32, 64}, {0, -1, -1}};

static int player_pixel(int x, int pos, uint8_t nusiz, uint8_t refl, uint8_t grp, int suppress) {
  if (!grp) return 0;
  const int mode = nusiz & 7;
  const int scale_shift = mode == 5 ? 1 : (mode == 7 ? 2 : 0);
  int d = x - pos;
  if (d < 0) d += 160;
  /* Stella 2.x computePlayerMaskTable: "in double [quad] size mode the player's output is delayed
   * by one pixel" (mask set for 0 < x <= 16 [32], bit (x-1)/2 [/4]).  Pinned by the reference's own
   * ALE recording (.github/Breakout.gif, tests/test_breakout_gif_pin.py): the 16-pixel paddle of
   * Breakout is a double-size player and sits one pixel right of the undelayed position. */
  if (scale_shift) d -= 1;
  for (int c = 0; c < 3 && k_copies[mode][c] >= 0; ++c) {
    if (c == 0 && suppress) continue;
    int off = d - k_copies[mode][c];
    if (off >= 0 && off < (8 << scale_shift)) {
      int k = off >> scale_shift;
      return (refl & 0x08) ? ((grp >> k) & 1) : ((grp >> (7 - k)) & 1);
    }
  }
  return 0;
}

static int missile_pixel(int x, int pos, uint8_t nusiz, uint8_t enam, uint8_t resmp) {
  if (!(enam & 0x02) || (resmp & 0x02)) return 0;
  const int mode = nusiz & 7;
  const int width = 1 << ((nusiz >> 4) & 3);
  int d = x - pos;
  if (d < 0) d += 160;
  for (int c = 0; c < 3 && k_copies[mode][c] >= 0; ++c) {
    int off = d - k_copies[mode][c];
    if (off >= 0 && off < width) return 1;
  }
  return 0;
}

static int ball_pixel(int x, int pos, uint8_t ctrlpf, int enabled) {
  if (!enabled) return 0;
  int d = x - pos;
  if (d < 0) d += 160;
  return d < (1 << ((ctrlpf >> 4) & 3));
}

static int playfield_pixel(const Atari* a, int x) {
  int i = x >> 2; /* 0..39 */
  if (i >= 20) i = (a->ctrlpf & 0x01) ? 39 - i : i - 20;
  if (i < 4) return (a->pf0 >> (4 + i)) & 1;
  if (i < 12) return (a->pf1 >> (11 - i)) & 1;
  return (a->pf2 >> (i - 12)) & 1;
}

static void tia_pixel(Atari* a, int x, int row) {
  uint8_t color;
  if (a->vblank & 0x02) {
    color = 0; /* Stella: blanked, no collision processing */
  } else {
    const uint8_t g0 = a->vdelp0 & 1 ? a->dgrp0 : a->grp0;
    const uint8_t g1 = a->vdelp1 & 1 ? a->dgrp1 : a->grp1;
    const int ebl = ((a->vdelbl & 1 ? a->denabl : a->enabl) & 0x02) != 0;
    const int p0 = player_pixel(x, a->posp0, a->nusiz0, a->refp0, g0, a->sup0);
    const int p1 = player_pixel(x, a->posp1, a->nusiz1, a->refp1, g1, a->sup1);
    const int m0 = missile_pixel(x, a->posm0, a->nusiz0, a->enam0, a->resmp0);
    const int m1 = missile_pixel(x, a->posm1, a->nusiz1, a->enam1, a->resmp1);
    const int bl = ball_pixel(x, a->posbl, a->ctrlpf, ebl);
    const int pf = playfield_pixel(a, x);
    uint16_t cx = 0;
    if (m0 && p1) cx |= CX_M0P1;
    if (m0 && p0) cx |= CX_M0P0;
    if (m1 && p0) cx |= CX_M1P0;
    if (m1 && p1) cx |= CX_M1P1;
    if (p0 && pf) cx |= CX_P0PF;
    if (p0 && bl) cx |= CX_P0BL;
    if (p1 && pf) cx |= CX_P1PF;
    if (p1 && bl) cx |= CX_P1BL;
    if (m0 && pf) cx |= CX_M0PF;
    if (m0 && bl) cx |= CX_M0BL;
    if (m1 && pf) cx |= CX_M1PF;
    if (m1 && bl) cx |= CX_M1BL;
    if (bl && pf) cx |= CX_BLPF;
    if (p0 && p1) cx |= CX_P0P1;
    if (m0 && m1) cx |= CX_M0M1;
    a->cx |= cx;
    /* Stella priority encoder (computePriorityEncoder): 0 BK, 1 PF, 2 P0, 3 P1 */
    int sel = 0;
    if (a->ctrlpf & 0x04) {
      if (p1 || m1) sel = 3;
      if (p0 || m0) sel = 2;
      if (bl) sel = 1;
      if (pf) sel = 1;
    } else {
      if (bl) sel = 1;
      if (pf) sel = (a->ctrlpf & 0x02) ? (x < 80 ? 2 : 3) : 1;
      if (p1 || m1) sel = (sel != 2) ? 3 : 2;
      if (p0 || m0) sel = 2;
    }
    color = sel == 0 ? a->colubk : sel == 1 ? a->colupf : sel == 2 ? a->colup0 : a->colup1;
    if (a->hmove_blank && x < 8) color = 0;
  }
  if (a->fb && row >= 0 && row < ATARI_H) a->fb[row * ATARI_W + x] = color;
}

/* advance the TIA to colour clock `clock` (same origin as cyc*3) */
static void tia_update(Atari* a, int32_t clock) {
  const int32_t c0 = frame_clock0(a);
  const int32_t start = c0 + CLOCKS_PER_LINE * ATARI_YSTART;
  const int32_t stop = start + CLOCKS_PER_LINE * ATARI_H;
  if (clock > stop) clock = stop;
  if (a->last_clock < start) a->last_clock = start;
  while (a->last_clock < clock) {
    const int32_t rel = a->last_clock - c0;
    const int hpos = rel % CLOCKS_PER_LINE;
    const int line = rel / CLOCKS_PER_LINE;
    if (hpos >= HBLANK) tia_pixel(a, hpos - HBLANK, line - ATARI_YSTART);
    if (hpos == CLOCKS_PER_LINE - 1) { /* end of scanline */
      a->sup0 = a->sup1 = 0;
      a->hmove_blank = 0;
    }
    a->last_clock++;
  }
}

static const int8_t k_poke_delay[64] = {
    /* VSYNC VBLANK WSYNC RSYNC NUSIZ0 NUSIZ1 COLUP0 COLUP1 COLUPF COLUBK CTRLPF REFP0 REFP1 PF0 PF1 PF2 */
    0, 1, 0, 0, 8, 8, 0, 0, 0, 0, 0, 1, 1, -1, -1, -1,
    /* RESP0 RESP1 RESM0 RESM1 RESBL AUDC0 AUDC1 AUDF0 AUDF1 AUDV0 AUDV1 GRP0 GRP1 ENAM0 ENAM1 ENABL */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1,
    /* HMP0 HMP1 HMM0 HMM1 HMBL VDELP0 VDELP1 VDELBL RESMP0 RESMP1 HMOVE HMCLR CXCLR ... */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static int16_t wrap160(int v) {
  while (v < 0) v += 160;
  while (v >= 160) v -= 160;
  return (int16_t)v;
}

static int hm_motion(uint8_t hm) { /* standard HMOVE row: 0,-1..-7, +8..+1 */
  int v = hm >> 4;
  return v < 8 ? -v : 16 - v;
}

static int missile_center(uint8_t nusiz) {
  const int mode = nusiz & 7;
  return mode == 5 ? 8 : (mode == 7 ? 16 : 4);
}

static void tia_write(Atari* a, uint8_t reg, uint8_t v) {
  reg &= 0x3f;
  const int32_t clock = a->cyc * 3;
  int delay = k_poke_delay[reg];
  const int hpos = (clock - frame_clock0(a)) % CLOCKS_PER_LINE;
  if (delay < 0) {
    static const int d[4] = {4, 5, 2, 3};
    delay = d[(hpos / 3) & 3];
  }
  tia_update(a, clock + delay);
  switch (reg) {
    case 0x00: /* VSYNC */
      a->vsync = v;
      if (v & 0x02) {
        a->vsync_finish_clock = clock + CLOCKS_PER_LINE;
      } else if (clock >= a->vsync_finish_clock) {
        a->vsync_finish_clock = 0x7fffffff;
        a->stop = 1;
      }
      break;
    case 0x01: /* VBLANK */
      if (!(a->vblank & 0x80) && (v & 0x80)) a->dump_enabled = 1;
      if ((a->vblank & 0x80) && !(v & 0x80)) {
        a->dump_enabled = 0;
        a->dump_disabled_cyc = a->cyc;
      }
      a->vblank = v;
      break;
    case 0x02: { /* WSYNC: halt until the start of the next scanline */
      int into = (a->cyc - a->cyc0) % 76;
      int rem = 76 - into;
      if (rem < 76) a->cyc += rem;
      break;
    }
    case 0x03: break; /* RSYNC: not modelled */
    case 0x04: a->nusiz0 = v; break;
    case 0x05: a->nusiz1 = v; break;
    case 0x06: a->colup0 = v & 0xfe; break;
    case 0x07: a->colup1 = v & 0xfe; break;
    case 0x08: a->colupf = v & 0xfe; break;
    case 0x09: a->colubk = v & 0xfe; break;
    case 0x0a: a->ctrlpf = v; break;
    case 0x0b: a->refp0 = v; break;
    case 0x0c: a->refp1 = v; break;
    case 0x0d: a->pf0 = v; break;
    case 0x0e: a->pf1 = v; break;
    case 0x0f: a->pf2 = v; break;
    case 0x10: a->posp0 = hpos < HBLANK ? 3 : wrap160(hpos - HBLANK + 5); a->sup0 = 1; break;
    case 0x11: a->posp1 = hpos < HBLANK ? 3 : wrap160(hpos - HBLANK + 5); a->sup1 = 1; break;
    case 0x12: a->posm0 = hpos < HBLANK ? 2 : wrap160(hpos - HBLANK + 4); break;
    case 0x13: a->posm1 = hpos < HBLANK ? 2 : wrap160(hpos - HBLANK + 4); break;
    case 0x14: a->posbl = hpos < HBLANK ? 2 : wrap160(hpos - HBLANK + 4); break;
    case 0x1b: a->grp0 = v; a->dgrp1 = a->grp1; break;
    case 0x1c: a->grp1 = v; a->dgrp0 = a->grp0; a->denabl = a->enabl; break;
    case 0x1d: a->enam0 = v; break;
    case 0x1e: a->enam1 = v; break;
    case 0x1f: a->enabl = v; break;
    case 0x20: a->hmp0 = v; break;
    case 0x21: a->hmp1 = v; break;
    case 0x22: a->hmm0 = v; break;
    case 0x23: a->hmm1 = v; break;
    case 0x24: a->hmbl = v; break;
    case 0x25: a->vdelp0 = v; break;
    case 0x26: a->vdelp1 = v; break;
    case 0x27: a->vdelbl = v; break;
    case 0x28:
      if ((a->resmp0 & 2) && !(v & 2)) a->posm0 = wrap160(a->posp0 + missile_center(a->nusiz0));
      a->resmp0 = v;
      break;
    case 0x29:
      if ((a->resmp1 & 2) && !(v & 2)) a->posm1 = wrap160(a->posp1 + missile_center(a->nusiz1));
      a->resmp1 = v;
      break;
    case 0x2a: /* HMOVE */
      if (hpos / 3 < 21) a->hmove_blank = 1;
      a->posp0 = wrap160(a->posp0 + hm_motion(a->hmp0));
      a->posp1 = wrap160(a->posp1 + hm_motion(a->hmp1));
      a->posm0 = wrap160(a->posm0 + hm_motion(a->hmm0));
      a->posm1 = wrap160(a->posm1 + hm_motion(a->hmm1));
      a->posbl = wrap160(a->posbl + hm_motion(a->hmbl));
      break;
    case 0x2b: a->hmp0 = a->hmp1 = a->hmm0 = a->hmm1 = a->hmbl = 0; break;
    case 0x2c: a->cx = 0; break;
    default: break; /* audio and unmapped */
  }
}

static uint8_t paddle_inpt(Atari* a, int which) {
  const int32_t r = a->paddle_res[which];
  if (a->dump_enabled) return 0x00;
  /* Stella: t = 1.6 * r * 0.01e-6 s ; needed = t * 1.19e6 cycles = r * 0.01904 (integer form) */
  const int32_t needed = (int32_t)(((int64_t)r * 1904) / 100000);
  return (a->cyc > a->dump_disabled_cyc + needed) ? 0x80 : 0x00;
}

static uint8_t tia_read(Atari* a, uint8_t reg) {
  tia_update(a, a->cyc * 3);
  const uint8_t noise = a->bus & 0x3f;
  uint8_t v = 0;
  const uint16_t c = a->cx;
  switch (reg & 0x0f) {
    case 0x0: v = ((c & CX_M0P1) ? 0x80 : 0) | ((c & CX_M0P0) ? 0x40 : 0); break;
    case 0x1: v = ((c & CX_M1P0) ? 0x80 : 0) | ((c & CX_M1P1) ? 0x40 : 0); break;
    case 0x2: v = ((c & CX_P0PF) ? 0x80 : 0) | ((c & CX_P0BL) ? 0x40 : 0); break;
    case 0x3: v = ((c & CX_P1PF) ? 0x80 : 0) | ((c & CX_P1BL) ? 0x40 : 0); break;
    case 0x4: v = ((c & CX_M0PF) ? 0x80 : 0) | ((c & CX_M0BL) ? 0x40 : 0); break;
    case 0x5: v = ((c & CX_M1PF) ? 0x80 : 0) | ((c & CX_M1BL) ? 0x40 : 0); break;
    case 0x6: v = (c & CX_BLPF) ? 0x80 : 0; break;
    case 0x7: v = ((c & CX_P0P1) ? 0x80 : 0) | ((c & CX_M0M1) ? 0x40 : 0); break;
    case 0x8: v = paddle_inpt(a, 0); break;
    case 0x9: v = paddle_inpt(a, 1); break;
    /* right port: paddles 2/3 whose resistance events ALE never sets (0 = minimum resistance) */
    case 0xa: v = a->dump_enabled ? 0x00 : 0x80; break;
    case 0xb: v = a->dump_enabled ? 0x00 : 0x80; break;
    case 0xc: v = 0x80; break; /* INPT4: joystick fire not pressed */
    case 0xd: v = 0x80; break;
    default: v = 0; break;
  }
  return (uint8_t)((v & 0xc0) | noise);
}

/* ===================================================================================== */
/* RIOT                                                                                  */
/* ===================================================================================== */
static uint8_t riot_read(Atari* a, uint16_t addr) {
  if (!(addr & 0x04)) {
    switch (addr & 3) {
      case 0: { /* SWCHA: paddle fire buttons are active-low on bits 7 (paddle 0) and 6 */
        uint8_t v = 0xff;
        if (a->paddle_fire[0]) v &= 0x7f;
        if (a->paddle_fire[1]) v &= 0xbf;
        return (uint8_t)((v & ~a->ddra) | (a->swcha_out & a->ddra));
      }
      case 1: return a->ddra;
      case 2: { /* SWCHB: b0 reset, b1 select (0 = pressed), b3 colour, b6/b7 difficulty B */
        uint8_t v = 0x0b;
        if (a->sw_reset) v &= ~0x01;
        if (a->sw_select) v &= ~0x02;
        return v;
      }
      default: return a->ddrb;
    }
  }
  /* timer: Stella M6532::peek */
  const int32_t delta = (a->cyc - 1) - a->timer_set_cyc;
  int32_t t = (int32_t)a->timer - (delta >> a->timer_shift) - 1;
  if (!(addr & 1)) { /* INTIM */
    if (t >= 0) return (uint8_t)t;
    t = ((int32_t)a->timer << a->timer_shift) - delta - 1;
    return (uint8_t)t;
  }
  return t >= 0 ? 0x00 : 0x80; /* TIMINT */
}

static void riot_write(Atari* a, uint16_t addr, uint8_t v) {
  if ((addr & 0x14) == 0x14) {
    static const uint8_t shifts[4] = {0, 3, 6, 10};
    a->timer = v;
    a->timer_shift = shifts[addr & 3];
    a->timer_set_cyc = a->cyc;
  } else if (!(addr & 0x04)) {
    switch (addr & 3) {
      case 0: a->swcha_out = v; break;
      case 1: a->ddra = v; break;
      case 2: a->swchb_out = v; break;
      default: a->ddrb = v; break;
    }
  }
}

/* ===================================================================================== */
/* bus                                                                                   */
/* ===================================================================================== */
static uint8_t rd(Atari* a, uint16_t addr) {
  a->cyc++;
  uint8_t v;
  if (addr & 0x1000) v = a->rom[addr & a->rom_mask];
  else if (!(addr & 0x80)) v = tia_read(a, (uint8_t)addr);
  else if (!(addr & 0x200)) v = a->ram[addr & 0x7f];
  else v = riot_read(a, addr);
  a->bus = v;
  return v;
}

static void wr(Atari* a, uint16_t addr, uint8_t v) {
  a->cyc++;
  a->bus = v;
  if (addr & 0x1000) return;
  if (!(addr & 0x80)) tia_write(a, (uint8_t)addr, v);
  else if (!(addr & 0x200)) a->ram[addr & 0x7f] = v;
  else riot_write(a, addr, v);
}

/* ===================================================================================== */
/* 6507                                                                                  */
/* ===================================================================================== */
#define FN 0x80
#define FV 0x40
#define FU 0x20
#define FB 0x10
#define FD 0x08
#define FI 0x04
#define FZ 0x02
#define FC 0x01

static inline void set_nz(Atari* a, uint8_t v) {
  a->P = (uint8_t)((a->P & ~(FN | FZ)) | (v & 0x80) | (v ? 0 : FZ));
}
static inline uint8_t fetch(Atari* a) { return rd(a, a->PC++); }
static inline void push(Atari* a, uint8_t v) { wr(a, 0x100 | a->S, v); a->S--; }
static inline uint8_t pull(Atari* a) { a->S++; return rd(a, 0x100 | a->S); }

enum { M_IMP, M_IMM, M_ZP, M_ZPX, M_ZPY, M_ABS, M_ABX, M_ABY, M_IZX, M_IZY };

/* effective address; `write` selects the fixed extra cycle of indexed stores / RMW */
static uint16_t ea(Atari* a, int mode, int write) {
  switch (mode) {
    case M_ZP: return fetch(a);
    case M_ZPX: { uint8_t z = fetch(a); a->cyc++; return (uint8_t)(z + a->X); }
    case M_ZPY: { uint8_t z = fetch(a); a->cyc++; return (uint8_t)(z + a->Y); }
    case M_ABS: { uint16_t lo = fetch(a); return (uint16_t)(lo | (fetch(a) << 8)); }
    case M_ABX: case M_ABY: {
      uint16_t lo = fetch(a); uint16_t base = (uint16_t)(lo | (fetch(a) << 8));
      uint16_t e = (uint16_t)(base + (mode == M_ABX ? a->X : a->Y));
      if (write || ((e ^ base) & 0xff00)) a->cyc++;
      return e;
    }
    case M_IZX: {
      uint8_t z = fetch(a); a->cyc++;
      uint8_t p = (uint8_t)(z + a->X);
      uint16_t lo = rd(a, p);
      return (uint16_t)(lo | (rd(a, (uint8_t)(p + 1)) << 8));
    }
    case M_IZY: {
      uint8_t z = fetch(a);
      uint16_t lo = rd(a, z);
      uint16_t base = (uint16_t)(lo | (rd(a, (uint8_t)(z + 1)) << 8));
      uint16_t e = (uint16_t)(base + a->Y);
      if (write || ((e ^ base) & 0xff00)) a->cyc++;
      return e;
    }
    default: return 0;
  }
}

static uint8_t bcd2bin(uint8_t t) { return (uint8_t)(((t >> 4) * 10) + (t & 0x0f)); }
static uint8_t bin2bcd(int t) { return (uint8_t)((((t % 100) / 10) << 4) | (t % 10)); }

static void op_adc(Atari* a, uint8_t m) {
  const uint8_t old = a->A;
  const int c = a->P & FC;
  if (a->P & FD) { /* Stella M6502 BCD-table formulation */
    int sum = bcd2bin(a->A) + bcd2bin(m) + c;
    a->P = (uint8_t)((a->P & ~FC) | (sum > 99 ? FC : 0));
    a->A = bin2bcd(sum & 0xff);
  } else {
    int sum = a->A + m + c;
    a->P = (uint8_t)((a->P & ~FC) | (sum > 0xff ? FC : 0));
    a->A = (uint8_t)sum;
  }
  set_nz(a, a->A);
  const int v = (~(old ^ m) & (old ^ a->A) & 0x80) != 0;
  a->P = (uint8_t)((a->P & ~FV) | (v ? FV : 0));
}

static void op_sbc(Atari* a, uint8_t m) {
  const uint8_t old = a->A;
  const int borrow = (a->P & FC) ? 0 : 1;
  if (a->P & FD) {
    int diff = bcd2bin(a->A) - bcd2bin(m) - borrow;
    if (diff < 0) diff += 100;
    a->A = bin2bcd(diff);
  } else {
    a->A = (uint8_t)(a->A - m - borrow);
  }
  const int carry = (int)old >= (int)m + borrow;
  a->P = (uint8_t)((a->P & ~FC) | (carry ? FC : 0));
  set_nz(a, a->A);
  const int v = ((old ^ m) & (old ^ a->A) & 0x80) != 0;
  a->P = (uint8_t)((a->P & ~FV) | (v ? FV : 0));
}

static void op_cmp(Atari* a, uint8_t r, uint8_t m) {
  a->P = (uint8_t)((a->P & ~FC) | (r >= m ? FC : 0));
  set_nz(a, (uint8_t)(r - m));
}

static void branch(Atari* a, int cond) {
  int8_t off = (int8_t)fetch(a);
  if (cond) {
    uint16_t t = (uint16_t)(a->PC + off);
    a->cyc += ((t ^ a->PC) & 0xff00) ? 2 : 1;
    a->PC = t;
  }
}

/* read-modify-write helpers: value read, one internal cycle, write back */
static uint8_t rmw_asl(Atari* a, uint8_t v) { a->P = (uint8_t)((a->P & ~FC) | (v >> 7)); v <<= 1; set_nz(a, v); return v; }
static uint8_t rmw_lsr(Atari* a, uint8_t v) { a->P = (uint8_t)((a->P & ~FC) | (v & 1)); v >>= 1; set_nz(a, v); return v; }
static uint8_t rmw_rol(Atari* a, uint8_t v) { uint8_t c = a->P & FC; a->P = (uint8_t)((a->P & ~FC) | (v >> 7)); v = (uint8_t)((v << 1) | c); set_nz(a, v); return v; }
static uint8_t rmw_ror(Atari* a, uint8_t v) { uint8_t c = a->P & FC; a->P = (uint8_t)((a->P & ~FC) | (v & 1)); v = (uint8_t)((v >> 1) | (c << 7)); set_nz(a, v); return v; }
static uint8_t rmw_inc(Atari* a, uint8_t v) { v++; set_nz(a, v); return v; }
static uint8_t rmw_dec(Atari* a, uint8_t v) { v--; set_nz(a, v); return v; }

static void cpu_step(Atari* a) {
  const uint8_t op = fetch(a);
  uint16_t e;
  uint8_t v;
#define RD(mode) ((mode) == M_IMM ? fetch(a) : rd(a, ea(a, (mode), 0)))
#define RMW(mode, fn) do { e = ea(a, mode, 1); v = rd(a, e); a->cyc++; wr(a, e, fn(a, v)); } while (0)
#define ALU8(base, stmt) \
    case base + 0x09: v = RD(M_IMM); stmt; break; case base + 0x05: v = RD(M_ZP); stmt; break; \
    case base + 0x15: v = RD(M_ZPX); stmt; break; case base + 0x0d: v = RD(M_ABS); stmt; break; \
    case base + 0x1d: v = RD(M_ABX); stmt; break; case base + 0x19: v = RD(M_ABY); stmt; break; \
    case base + 0x01: v = RD(M_IZX); stmt; break; case base + 0x11: v = RD(M_IZY); stmt; break;
  switch (op) {
    ALU8(0x00, (a->A |= v, set_nz(a, a->A)))          /* ORA */
    ALU8(0x20, (a->A &= v, set_nz(a, a->A)))          /* AND */
    ALU8(0x40, (a->A ^= v, set_nz(a, a->A)))          /* EOR */
    ALU8(0x60, op_adc(a, v))                          /* ADC */
    ALU8(0xa0, (a->A = v, set_nz(a, v)))              /* LDA */
    ALU8(0xc0, op_cmp(a, a->A, v))                    /* CMP */
    ALU8(0xe0, op_sbc(a, v))                          /* SBC */
    /* STA */
    case 0x85: wr(a, ea(a, M_ZP, 1), a->A); break;  case 0x95: wr(a, ea(a, M_ZPX, 1), a->A); break;
    case 0x8d: wr(a, ea(a, M_ABS, 1), a->A); break; case 0x9d: wr(a, ea(a, M_ABX, 1), a->A); break;
    case 0x99: wr(a, ea(a, M_ABY, 1), a->A); break; case 0x81: wr(a, ea(a, M_IZX, 1), a->A); break;
    case 0x91: wr(a, ea(a, M_IZY, 1), a->A); break;
    /* LDX / LDY */
    case 0xa2: a->X = RD(M_IMM); set_nz(a, a->X); break; case 0xa6: a->X = RD(M_ZP); set_nz(a, a->X); break;
    case 0xb6: a->X = RD(M_ZPY); set_nz(a, a->X); break; case 0xae: a->X = RD(M_ABS); set_nz(a, a->X); break;
    case 0xbe: a->X = RD(M_ABY); set_nz(a, a->X); break;
    case 0xa0: a->Y = RD(M_IMM); set_nz(a, a->Y); break; case 0xa4: a->Y = RD(M_ZP); set_nz(a, a->Y); break;
    case 0xb4: a->Y = RD(M_ZPX); set_nz(a, a->Y); break; case 0xac: a->Y = RD(M_ABS); set_nz(a, a->Y); break;
    case 0xbc: a->Y = RD(M_ABX); set_nz(a, a->Y); break;
    /* STX / STY */
    case 0x86: wr(a, ea(a, M_ZP, 1), a->X); break; case 0x96: wr(a, ea(a, M_ZPY, 1), a->X); break;
    case 0x8e: wr(a, ea(a, M_ABS, 1), a->X); break;
    case 0x84: wr(a, ea(a, M_ZP, 1), a->Y); break; case 0x94: wr(a, ea(a, M_ZPX, 1), a->Y); break;
    case 0x8c: wr(a, ea(a, M_ABS, 1), a->Y); break;
    /* CPX / CPY */
    case 0xe0: op_cmp(a, a->X, RD(M_IMM)); break; case 0xe4: op_cmp(a, a->X, RD(M_ZP)); break;
    case 0xec: op_cmp(a, a->X, RD(M_ABS)); break;
    case 0xc0: op_cmp(a, a->Y, RD(M_IMM)); break; case 0xc4: op_cmp(a, a->Y, RD(M_ZP)); break;
    case 0xcc: op_cmp(a, a->Y, RD(M_ABS)); break;
    /* BIT */
    case 0x24: case 0x2c:
      v = RD(op == 0x24 ? M_ZP : M_ABS);
      a->P = (uint8_t)((a->P & ~(FN | FV | FZ)) | (v & 0xc0) | ((a->A & v) ? 0 : FZ));
      break;
    /* shifts / inc / dec */
    case 0x0a: a->cyc++; a->A = rmw_asl(a, a->A); break; case 0x06: RMW(M_ZP, rmw_asl); break;
    case 0x16: RMW(M_ZPX, rmw_asl); break; case 0x0e: RMW(M_ABS, rmw_asl); break; case 0x1e: RMW(M_ABX, rmw_asl); break;
    case 0x4a: a->cyc++; a->A = rmw_lsr(a, a->A); break; case 0x46: RMW(M_ZP, rmw_lsr); break;
    case 0x56: RMW(M_ZPX, rmw_lsr); break; case 0x4e: RMW(M_ABS, rmw_lsr); break; case 0x5e: RMW(M_ABX, rmw_lsr); break;
    case 0x2a: a->cyc++; a->A = rmw_rol(a, a->A); break; case 0x26: RMW(M_ZP, rmw_rol); break;
    case 0x36: RMW(M_ZPX, rmw_rol); break; case 0x2e: RMW(M_ABS, rmw_rol); break; case 0x3e: RMW(M_ABX, rmw_rol); break;
    case 0x6a: a->cyc++; a->A = rmw_ror(a, a->A); break; case 0x66: RMW(M_ZP, rmw_ror); break;
    case 0x76: RMW(M_ZPX, rmw_ror); break; case 0x6e: RMW(M_ABS, rmw_ror); break; case 0x7e: RMW(M_ABX, rmw_ror); break;
    case 0xe6: RMW(M_ZP, rmw_inc); break; case 0xf6: RMW(M_ZPX, rmw_inc); break;
    case 0xee: RMW(M_ABS, rmw_inc); break; case 0xfe: RMW(M_ABX, rmw_inc); break;
    case 0xc6: RMW(M_ZP, rmw_dec); break; case 0xd6: RMW(M_ZPX, rmw_dec); break;
    case 0xce: RMW(M_ABS, rmw_dec); break; case 0xde: RMW(M_ABX, rmw_dec); break;
    /* register ops (2 cycles) */
    case 0xe8: a->cyc++; a->X++; set_nz(a, a->X); break; case 0xc8: a->cyc++; a->Y++; set_nz(a, a->Y); break;
    case 0xca: a->cyc++; a->X--; set_nz(a, a->X); break; case 0x88: a->cyc++; a->Y--; set_nz(a, a->Y); break;
    case 0xaa: a->cyc++; a->X = a->A; set_nz(a, a->X); break; case 0xa8: a->cyc++; a->Y = a->A; set_nz(a, a->Y); break;
    case 0x8a: a->cyc++; a->A = a->X; set_nz(a, a->A); break; case 0x98: a->cyc++; a->A = a->Y; set_nz(a, a->A); break;
    case 0xba: a->cyc++; a->X = a->S; set_nz(a, a->X); break; case 0x9a: a->cyc++; a->S = a->X; break;
    case 0x18: a->cyc++; a->P &= ~FC; break; case 0x38: a->cyc++; a->P |= FC; break;
    case 0x58: a->cyc++; a->P &= ~FI; break; case 0x78: a->cyc++; a->P |= FI; break;
    case 0xb8: a->cyc++; a->P &= ~FV; break;
    case 0xd8: a->cyc++; a->P &= ~FD; break; case 0xf8: a->cyc++; a->P |= FD; break;
    case 0xea: a->cyc++; break;
    /* stack */
    case 0x48: a->cyc++; push(a, a->A); break;
    case 0x08: a->cyc++; push(a, a->P | FB | FU); break;
    /* PLA/PLP: the two dummy reads (next opcode byte, then the old stack top) are real bus
     * cycles — they decide the data-bus 'noise' bits when the stack sits in TIA space
     * (Breakout: PHP/PLA at S=$1F strobes ENABL) */
    case 0x68: rd(a, a->PC); rd(a, 0x100 | a->S); a->A = pull(a); set_nz(a, a->A); break;
    case 0x28: rd(a, a->PC); rd(a, 0x100 | a->S); a->P = (uint8_t)((pull(a) & ~FB) | FU); break;
    /* branches */
    case 0x10: branch(a, !(a->P & FN)); break; case 0x30: branch(a, a->P & FN); break;
    case 0x50: branch(a, !(a->P & FV)); break; case 0x70: branch(a, a->P & FV); break;
    case 0x90: branch(a, !(a->P & FC)); break; case 0xb0: branch(a, a->P & FC); break;
    case 0xd0: branch(a, !(a->P & FZ)); break; case 0xf0: branch(a, a->P & FZ); break;
    /* jumps */
    case 0x4c: { uint16_t lo = fetch(a); a->PC = (uint16_t)(lo | (fetch(a) << 8)); break; }
    case 0x6c: {
      uint16_t lo = fetch(a); uint16_t p = (uint16_t)(lo | (fetch(a) << 8));
      uint16_t l = rd(a, p);
      a->PC = (uint16_t)(l | (rd(a, (uint16_t)((p & 0xff00) | ((p + 1) & 0xff))) << 8)); /* page-wrap bug */
      break;
    }
    case 0x20: {
      uint16_t lo = fetch(a); a->cyc++;
      push(a, (uint8_t)(a->PC >> 8)); push(a, (uint8_t)a->PC);
      a->PC = (uint16_t)(lo | (fetch(a) << 8));
      break;
    }
    case 0x60: {
      a->cyc += 2;
      uint16_t lo = pull(a); a->PC = (uint16_t)(lo | (pull(a) << 8));
      a->PC++; a->cyc++;
      break;
    }
    case 0x40: {
      a->cyc += 2;
      a->P = (uint8_t)((pull(a) & ~FB) | FU);
      uint16_t lo = pull(a); a->PC = (uint16_t)(lo | (pull(a) << 8));
      break;
    }
    case 0x00: { /* BRK */
      fetch(a);
      push(a, (uint8_t)(a->PC >> 8)); push(a, (uint8_t)a->PC); push(a, a->P | FB | FU);
      a->P |= FI;
      uint16_t lo = rd(a, 0xfffe); a->PC = (uint16_t)(lo | (rd(a, 0xffff) << 8));
      break;
    }
    default:
      a->jam = op ? op : 0x100; /* undocumented opcode: treat as 2-cycle NOP and flag it */
      a->cyc++;
      break;
  }
#undef RD
#undef RMW
#undef ALU8
}

/* ===================================================================================== */
/* system                                                                                */
/* ===================================================================================== */
void atari_init(Atari* a, const uint8_t* rom, uint32_t rom_size) {
  memset(a, 0, sizeof(*a));
  a->rom = rom;
  a->rom_mask = rom_size - 1; /* 2K and 4K carts, mirrored */
  atari_system_reset(a);
}

void atari_system_reset(Atari* a) {
  const uint8_t* rom = a->rom;
  const uint32_t mask = a->rom_mask;
  memset(a, 0, sizeof(*a));
  a->rom = rom;
  a->rom_mask = mask;
  a->S = 0xff;
  a->P = FU | FI;
  a->PC = (uint16_t)(rom[0xffc & mask] | (rom[0xffd & mask] << 8));
  a->vsync_finish_clock = 0x7fffffff;
  a->paddle_res[0] = a->paddle_res[1] = 408823;
  a->timer = 0; a->timer_shift = 10; a->timer_set_cyc = 0;
}

void atari_frame(Atari* a, uint8_t* fb) {
  /* Stella TIA::startFrame */
  const int32_t into = (a->cyc - a->cyc0) % 76;
  const int32_t old = a->cyc;
  a->cyc = 0;
  a->cyc0 = -into;
  a->timer_set_cyc -= old;
  a->dump_disabled_cyc -= old;
  if (a->vsync_finish_clock != 0x7fffffff) a->vsync_finish_clock -= old * 3;
  a->last_clock = frame_clock0(a) + CLOCKS_PER_LINE * ATARI_YSTART;
  a->fb = fb;
  a->stop = 0;
  for (int n = 0; n < 25000 && !a->stop; ++n) cpu_step(a);
  a->fb = 0;
}

/* ===================================================================================== */
/* ALE layer                                                                             */
/* ===================================================================================== */
#define PADDLE_DELTA 23000
#define PADDLE_MIN 27450
#define PADDLE_MAX 790196
#define PADDLE_DEFAULT (((PADDLE_MAX - PADDLE_MIN) / 2) + PADDLE_MIN)

int ale_minimal_actions(int game, int* out) {
  static const int pong[6] = {ACT_NOOP, ACT_FIRE, ACT_RIGHT, ACT_LEFT, ACT_RIGHTFIRE, ACT_LEFTFIRE};
  static const int brk[4] = {ACT_NOOP, ACT_FIRE, ACT_RIGHT, ACT_LEFT};
  if (game == GAME_BREAKOUT) { memcpy(out, brk, sizeof(brk)); return 4; }
  memcpy(out, pong, sizeof(pong));
  return 6;
}

void ale_init(Ale* e, const uint8_t* rom, uint32_t rom_size, int game) {
  memset(e, 0, sizeof(*e));
  atari_init(&e->emu, rom, rom_size);
  e->game = game;
  e->paddle = PADDLE_DEFAULT;
}

static void ale_apply_action(Ale* e, int act) {
  Atari* a = &e->emu;
  int delta = 0, fire = 0;
  a->sw_reset = 0;
  switch (act) {
    case ACT_RIGHT: delta = -PADDLE_DELTA; break;
    case ACT_LEFT: delta = PADDLE_DELTA; break;
    case ACT_RIGHTFIRE: delta = -PADDLE_DELTA; fire = 1; break;
    case ACT_LEFTFIRE: delta = PADDLE_DELTA; fire = 1; break;
    case ACT_FIRE: fire = 1; break;
    case ACT_RESET: a->sw_reset = 1; break;
    default: break;
  }
  e->paddle += delta;
  if (e->paddle < PADDLE_MIN) e->paddle = PADDLE_MIN;
  if (e->paddle > PADDLE_MAX) e->paddle = PADDLE_MAX;
  /* Stella props: Video Olympics has Controller.SwapPaddles=YES, so ALE's paddle A drives
   * INPT1 / SWCHA bit 6 there; Breakout is unswapped (INPT0 / SWCHA bit 7). */
  const int mine = e->game == GAME_PONG ? 1 : 0;
  a->paddle_res[mine] = e->paddle;
  a->paddle_res[1 - mine] = PADDLE_DEFAULT;
  a->paddle_fire[mine] = (uint8_t)fire;
  a->paddle_fire[1 - mine] = 0;
}

static void ale_rom_step(Ale* e) {
  const uint8_t* ram = e->emu.ram;
  if (e->game == GAME_PONG) {
    int x = ram[13], y = ram[14];
    int score = y - x;
    e->reward = score - e->score;
    e->score = score;
    e->terminal = (x == 21 || y == 21);
    e->lives = 0;
  } else if (e->game == GAME_BREAKOUT) {
    int x = ram[77], y = ram[76];
    int score = 1 * (x & 0x0f) + 10 * ((x & 0xf0) >> 4) + 100 * (y & 0x0f);
    e->reward = score - e->score;
    e->score = score;
    int lives = ram[57];
    if (!e->started && lives == 5) e->started = 1;
    e->terminal = e->started && lives == 0;
    e->lives = lives;
  } else {
    e->reward = 0;
  }
}

static void ale_rom_reset(Ale* e) {
  e->reward = 0; e->score = 0; e->terminal = 0; e->started = 0;
  e->lives = e->game == GAME_BREAKOUT ? 5 : 0;
}

int32_t ale_act(Ale* e, int act, uint8_t* fb) {
  ale_apply_action(e, act);
  atari_frame(&e->emu, fb);
  ale_rom_step(e);
  e->frame_number++;
  return e->reward;
}

void ale_reset(Ale* e, uint8_t* fb) {
  e->paddle = PADDLE_DEFAULT;
  atari_system_reset(&e->emu);
  for (int i = 0; i < 60; ++i) { ale_apply_action(e, ACT_NOOP); atari_frame(&e->emu, fb); }
  for (int i = 0; i < 4; ++i) { ale_apply_action(e, ACT_RESET); atari_frame(&e->emu, fb); }
  ale_rom_reset(e);
  e->frame_number = 0;
}

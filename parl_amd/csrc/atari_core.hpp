// atari_core.hpp — device-side Atari 2600 (6507 + TIA + RIOT) for gfx950: ONE ENV PER WAVEFRONT.
//
// What it stands for in the reference: the third-party ALE/Stella emulator behind
// gym.make('PongNoFrameskip-v4') that parl/env/atari_wrappers.py:356-385 wraps and
// examples/IMPALA/actor.py:33-39 steps.  Semantics are those documented in
// oracle/atari_oracle.h; the CPU twin there is the parity oracle (bit-exact frames/RAM/rewards).
//
// Execution model (CDNA4):
//   * the 6507, RIOT and the TIA register file are WAVE-UNIFORM: all CPU state lives in SGPRs
//     (every value is made uniform with v_readfirstlane / v_readlane), so instruction decode,
//     ALU ops and branches run on the scalar unit and never diverge inside the wave;
//   * the 128 bytes of RAM live in two VGPRs (lane i holds ram[i] and ram[64+i]) and are
//     accessed with v_readlane / v_writelane — no LDS or memory latency on the data path;
//   * the TIA registers live in one VGPR (lane r = register r) — a register write is one
//     v_writelane;
//   * the cartridge is pre-decoded on the host into one 32-bit word per address
//     (operand bytes + addressing mode/op id) and staged in LDS once per workgroup, so an
//     instruction fetch is ONE ds_read_b32;
//   * the 64 lanes come into play for the TIA: on every register access the picture is
//     caught up to the access's colour clock, 64 pixels per pass (lane = pixel), object masks
//     are reduced to collision latches with wave ballots, and the lanes store 64 contiguous
//     colour bytes of the frame buffer.
#pragma once
#include <hip/hip_runtime.h>
#include "atari_defs.hpp"

namespace parlhip {
namespace atari {

#define DEVI __device__ __forceinline__

DEVI int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEVI int rlane(uint32_t v, int lane) { return __builtin_amdgcn_readlane((int)v, lane); }
// single-lane update: v_cmp_eq (lane id vs SGPR) + v_cndmask (clang has no writelane builtin and
// v_writelane with two SGPR operands violates gfx9's one-SGPR constant-bus rule)
DEVI uint32_t wlane(uint32_t v, int lane, int val) {
  return ((int)(threadIdx.x & 63) == lane) ? (uint32_t)val : v;
}

enum : int { FN = 0x80, FV = 0x40, FU = 0x20, FB = 0x10, FD = 0x08, FI = 0x04, FZ = 0x02, FC = 0x01 };
enum : int { CX_M0P1 = 1 << 0, CX_M0P0 = 1 << 1, CX_M1P0 = 1 << 2, CX_M1P1 = 1 << 3,
             CX_P0PF = 1 << 4, CX_P0BL = 1 << 5, CX_P1PF = 1 << 6, CX_P1BL = 1 << 7,
             CX_M0PF = 1 << 8, CX_M0BL = 1 << 9, CX_M1PF = 1 << 10, CX_M1BL = 1 << 11,
             CX_BLPF = 1 << 12, CX_P0P1 = 1 << 13, CX_M0M1 = 1 << 14 };
enum : int { JAM_OPCODE = 0x1000, JAM_ZP_PTR_TIA = 0x200, JAM_RMW_TIA = 0x400, JAM_STACK = 0x800,
             JAM_JMPI = 0x2000 };

struct Emu {
  // ---- wave-uniform CPU / system state (SGPRs) ----
  int A, X, Y, S, P, PC;
  int cyc, cyc0, last_clock, vsync_finish, dump_dis_cyc, dump_en;
  int timer, timer_shift, timer_set_cyc, ddra, ddrb, swcha_out, swchb_out, cx, jam, stop;
  int paddle_res0, paddle_res1, fire0, fire1, sw_reset;
  // ---- lane-distributed state (VGPRs) ----
  uint32_t ram_lo, ram_hi, tia;
  // ---- environment of the emulation ----
  const uint32_t* romw;  // LDS: pre-decoded cartridge words
  int rom_mask;
  uint8_t* fb;           // frame buffer for the frame being rendered (nullptr: collisions only)
  int lane;

  DEVI int ram_rd(int a) const {
    const int l = a & 63;
    return (a & 64) ? rlane(ram_hi, l) : rlane(ram_lo, l);
  }
  DEVI void ram_wr(int a, int v) {
    const int l = a & 63;
    if (a & 64) ram_hi = wlane(ram_hi, l, v); else ram_lo = wlane(ram_lo, l, v);
  }
  DEVI int t(int r) const { return rlane(tia, r); }
  DEVI void tset(int r, int v) { tia = wlane(tia, r, v); }
  DEVI int rom_byte(int a) const { return rfl((int)(romw[(a - 1) & rom_mask] & 0xff)); }

  DEVI void set_nz(int v) { P = (P & ~(FN | FZ)) | (v & 0x80) | ((v & 0xff) ? 0 : FZ); }

  // ------------------------------------------------------------------------------------
  // TIA picture: catch the frame up to colour clock `clock`
  // ------------------------------------------------------------------------------------
  DEVI void render_pass(int xs, int xe, int row) {
    // lanes cover pixels x = xs + lane < xe of frame-buffer row `row`
    const int x = xs + lane;
    const bool act = x < xe;
    const int vblank = t(T_VBLANK);
    int color = 0;
    if (!(vblank & 0x02)) {
      const int nus0 = t(T_NUSIZ0), nus1 = t(T_NUSIZ1), ctrlpf = t(T_CTRLPF);
      const int g0 = (t(T_VDELP0) & 1) ? t(T_DGRP0) : t(T_GRP0);
      const int g1 = (t(T_VDELP1) & 1) ? t(T_DGRP1) : t(T_GRP1);
      const int ebl = (((t(T_VDELBL) & 1) ? t(T_DENABL) : t(T_ENABL)) & 0x02) != 0;
      auto copies = [](int mode, int& c1, int& c2) {
        // second / third copy offsets of NUSIZ mode (-1: none)
        c1 = (mode == 1 || mode == 3) ? 16 : ((mode == 2 || mode == 6) ? 32 : (mode == 4 ? 64 : -1));
        c2 = mode == 3 ? 32 : (mode == 6 ? 64 : -1);
      };
      // NOTE: lane-wise predicates below use the non-short-circuit & and | on purpose: `a && b`
      // with per-lane operands compiles to a divergent branch (s_and_saveexec), and one divergent
      // branch inside the interpreter loop makes LLVM structurize the wave-uniform code around it.
      auto player = [&](int pos, int nus, int refl, int grp, int sup) -> bool {
        const int mode = nus & 7;
        const int sh = mode == 5 ? 1 : (mode == 7 ? 2 : 0);
        int c1, c2;
        copies(mode, c1, c2);
        int d = x - pos;
        d += d < 0 ? 160 : 0;
        const int w = 8 << sh;
        const bool in0 = (d < w) & !sup;
        const int o1 = d - c1, o2 = d - c2;
        const bool in1 = (c1 >= 0) & (o1 >= 0) & (o1 < w);
        const bool in2 = (c2 >= 0) & (o2 >= 0) & (o2 < w);
        const int off = in0 ? d : (in1 ? o1 : o2);
        const int k = (off >> sh) & 7;
        const int bit = (refl & 0x08) ? ((grp >> k) & 1) : ((grp >> (7 - k)) & 1);
        return (in0 | in1 | in2) & (bit != 0);
      };
      auto missile = [&](int pos, int nus) -> bool {
        const int mode = nus & 7;
        int c1, c2;
        copies(mode, c1, c2);
        const int w = 1 << ((nus >> 4) & 3);
        int d = x - pos;
        d += d < 0 ? 160 : 0;
        const int o1 = d - c1, o2 = d - c2;
        return (d < w) | ((c1 >= 0) & (o1 >= 0) & (o1 < w)) | ((c2 >= 0) & (o2 >= 0) & (o2 < w));
      };
      // wave-uniform enables first: a disabled object costs one scalar branch, not its pixel math
      // (most Pong / Breakout scanlines have no player, missile or ball at all)
      bool p0 = false, p1 = false, m0 = false, m1 = false, bl = false;
      if (g0) p0 = act & player(t(T_POSP0), nus0, t(T_REFP0), g0, t(T_SUP0));
      if (g1) p1 = act & player(t(T_POSP1), nus1, t(T_REFP1), g1, t(T_SUP1));
      if ((t(T_ENAM0) & 0x02) && !(t(T_RESMP0) & 0x02)) m0 = act & missile(t(T_POSM0), nus0);
      if ((t(T_ENAM1) & 0x02) && !(t(T_RESMP1) & 0x02)) m1 = act & missile(t(T_POSM1), nus1);
      if (ebl) {
        int d = x - t(T_POSBL);
        d += d < 0 ? 160 : 0;
        bl = act & (d < (1 << ((ctrlpf >> 4) & 3)));
      }
      bool pf;
      {
        // 20 playfield bits of the half line as one uniform word: bit i = PF0[4+i] (i<4),
        // PF1[11-i] (i<12), PF2[i-12]; per lane only a shift by its column index remains
        const int pfw = ((t(T_PF0) >> 4) & 0xf) | ((int)(__builtin_bitreverse32((uint32_t)t(T_PF1)) >> 24) << 4) |
                        (t(T_PF2) << 12);
        int i = x >> 2;
        i = i >= 20 ? ((ctrlpf & 1) ? 39 - i : i - 20) : i;
        pf = act & (((pfw >> (i & 31)) & 1) != 0);
      }
      const unsigned long long P0 = __ballot(p0), P1 = __ballot(p1), M0 = __ballot(m0),
                               M1 = __ballot(m1), BL = __ballot(bl), PF = __ballot(pf);
      if ((P0 | P1 | M0 | M1 | BL) != 0ull) {
        int c = 0;
        c |= (M0 & P1) ? CX_M0P1 : 0;  c |= (M0 & P0) ? CX_M0P0 : 0;
        c |= (M1 & P0) ? CX_M1P0 : 0;  c |= (M1 & P1) ? CX_M1P1 : 0;
        c |= (P0 & PF) ? CX_P0PF : 0;  c |= (P0 & BL) ? CX_P0BL : 0;
        c |= (P1 & PF) ? CX_P1PF : 0;  c |= (P1 & BL) ? CX_P1BL : 0;
        c |= (M0 & PF) ? CX_M0PF : 0;  c |= (M0 & BL) ? CX_M0BL : 0;
        c |= (M1 & PF) ? CX_M1PF : 0;  c |= (M1 & BL) ? CX_M1BL : 0;
        c |= (BL & PF) ? CX_BLPF : 0;  c |= (P0 & P1) ? CX_P0P1 : 0;
        c |= (M0 & M1) ? CX_M0M1 : 0;
        cx |= c;
      }
      if (fb) {
        int sel = 0;  // Stella priority encoder: 0 BK, 1 PF, 2 P0, 3 P1
        const bool o0 = p0 | m0, o1 = p1 | m1;
        if (ctrlpf & 0x04) {
          sel = o1 ? 3 : sel;
          sel = o0 ? 2 : sel;
          sel = (bl | pf) ? 1 : sel;
        } else {
          const int score = (ctrlpf & 0x02) != 0;           // score mode: PF takes the player colours
          const int pfsel = score ? (x < 80 ? 2 : 3) : 1;   // per-lane select of two uniform values
          sel = bl ? 1 : sel;
          sel = pf ? pfsel : sel;
          sel = o1 ? ((sel != 2) ? 3 : 2) : sel;
          sel = o0 ? 2 : sel;
        }
        // the four colour registers as one uniform word, indexed per lane by a shift (branch-free)
        const uint32_t cw = (uint32_t)t(T_COLUBK) | ((uint32_t)t(T_COLUPF) << 8) | ((uint32_t)t(T_COLUP0) << 16) |
                            ((uint32_t)t(T_COLUP1) << 24);
        color = (int)((cw >> (sel * 8)) & 0xffu);
        color = ((t(T_HMBLANK) != 0) & (x < 8)) ? 0 : color;
      }
    }
    if (fb) {
      // Predication without a divergent branch: inactive lanes store to an out-of-range offset of
      // a raw buffer resource, which the hardware drops.  A per-lane `if (act)` here would be the
      // only divergent branch of the whole interpreter loop and would force LLVM to structurize
      // (and bloat) the otherwise wave-uniform control flow around it.
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(fb, 0, kFrameBytes, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)color, rsrc, act ? row * kW + x : -1, 0, 0);
    }
  }

  // wave-uniform: can any pixel of the current register state set a collision latch?
  DEVI bool can_collide() const {
    if (t(T_VBLANK) & 0x02) return false;
    const int g0 = (t(T_VDELP0) & 1) ? t(T_DGRP0) : t(T_GRP0);
    const int g1 = (t(T_VDELP1) & 1) ? t(T_DGRP1) : t(T_GRP1);
    const int ebl = ((t(T_VDELBL) & 1) ? t(T_DENABL) : t(T_ENABL)) & 0x02;
    const int m0 = (t(T_ENAM0) & 0x02) && !(t(T_RESMP0) & 0x02);
    const int m1 = (t(T_ENAM1) & 0x02) && !(t(T_RESMP1) & 0x02);
    return (g0 | g1 | ebl | m0 | m1) != 0;
  }

  // TIA write classes (bit = register number).  kPlainRegs: the write is "tset(reg, v)" and
  // nothing else (VBLANK additionally drives the paddle dump) — rewriting the value already held
  // is a complete no-op, and Pong / Breakout rewrite PF1/PF2/GRPx/ENAxx with unchanged values on
  // almost every scanline.  kStrobeRegs: position strobes, RESMPx, HMOVE, CXCLR — always need the
  // picture caught up.  Everything else (VSYNC, WSYNC, RSYNC, audio, HMxx, HMCLR, unmapped) never
  // does: HMxx only matter at the next HMOVE.  GRP0 / GRP1 (also latching the delayed copies) are
  // classified in place.
  static constexpr unsigned long long kPlainRegs =
      (1ull << 0x01) | (1ull << 0x04) | (1ull << 0x05) | (1ull << 0x06) | (1ull << 0x07) | (1ull << 0x08) |
      (1ull << 0x09) | (1ull << 0x0a) | (1ull << 0x0b) | (1ull << 0x0c) | (1ull << 0x0d) | (1ull << 0x0e) |
      (1ull << 0x0f) | (1ull << 0x1d) | (1ull << 0x1e) | (1ull << 0x1f) | (1ull << 0x25) | (1ull << 0x26) |
      (1ull << 0x27);
  static constexpr unsigned long long kStrobeRegs =
      (1ull << 0x10) | (1ull << 0x11) | (1ull << 0x12) | (1ull << 0x13) | (1ull << 0x14) | (1ull << 0x28) |
      (1ull << 0x29) | (1ull << 0x2a) | (1ull << 0x2c);

  DEVI void tia_update(int clock) {
    const int c0 = cyc0 * 3;
    const int start = c0 + kClocksPerLine * kYStart;
    const int stop_clock = start + kClocksPerLine * kH;
    clock = clock > stop_clock ? stop_clock : clock;
    last_clock = last_clock < start ? start : last_clock;
    if (last_clock >= clock) return;
#ifdef PARLHIP_EXP_NORENDER
    last_clock = clock; return;
#endif
    if (!fb && !can_collide()) {
      // collisions-only frame and nothing to latch in this span (VBLANK on, or no player /
      // missile / ball enabled): skip the pixels, keep the end-of-scanline side effects
      if ((clock - c0) / kClocksPerLine > (last_clock - c0) / kClocksPerLine) {
        tset(T_SUP0, 0);
        tset(T_SUP1, 0);
        tset(T_HMBLANK, 0);
      }
      last_clock = clock;
      return;
    }
    while (last_clock < clock) {
      const int rel = last_clock - c0;
      const int line = rel / kClocksPerLine;
      const int hpos = rel - line * kClocksPerLine;
      const int line_end = last_clock + (kClocksPerLine - hpos);
      const int seg_end = clock < line_end ? clock : line_end;
      const int h0 = hpos < kHBlank ? kHBlank : hpos;
      const int h1 = hpos + (seg_end - last_clock);
      for (int xs = h0 - kHBlank; xs < h1 - kHBlank; xs += 64) render_pass(xs, h1 - kHBlank, line - kYStart);
      if (seg_end == line_end) {  // end of scanline
        tset(T_SUP0, 0);
        tset(T_SUP1, 0);
        tset(T_HMBLANK, 0);
      }
      last_clock = seg_end;
    }
  }

  DEVI static int wrap160(int v) {
    v += v < 0 ? 160 : 0;
    v -= v >= 160 ? 160 : 0;
    return v;
  }
  DEVI static int hm_motion(int hm) {
    const int v = (hm >> 4) & 15;
    return v < 8 ? -v : 16 - v;
  }
  DEVI static int missile_center(int nusiz) {
    const int mode = nusiz & 7;
    return mode == 5 ? 8 : (mode == 7 ? 16 : 4);
  }
  DEVI static int poke_delay(int reg, int hpos) {
    if (reg == 0x01 || reg == 0x0b || reg == 0x0c || (reg >= 0x1b && reg <= 0x1f)) return 1;
    if (reg == 0x04 || reg == 0x05) return 8;
    if (reg >= 0x0d && reg <= 0x0f) {
      const int q = (hpos / 3) & 3;  // {4, 5, 2, 3}
      return q == 0 ? 4 : (q == 1 ? 5 : (q == 2 ? 2 : 3));
    }
    return 0;
  }

  // register write AFTER the picture has been caught up (tia_update(clock + delay) done by caller)
  DEVI void tia_write(int reg, int v) {
    const int clock = cyc * 3;
    const int rel = clock - cyc0 * 3;
    const int hpos = rel % kClocksPerLine;
    switch (reg) {
      case 0x00:
        tset(T_VSYNC, v);
        if (v & 0x02) {
          vsync_finish = clock + kClocksPerLine;
        } else if (clock >= vsync_finish) {
          vsync_finish = 0x7fffffff;
          stop = 1;
        }
        break;
      case 0x01: {
        const int old = t(T_VBLANK);
        if (!(old & 0x80) && (v & 0x80)) dump_en = 1;
        if ((old & 0x80) && !(v & 0x80)) { dump_en = 0; dump_dis_cyc = cyc; }
        tset(T_VBLANK, v);
        break;
      }
      case 0x02: {
        const int into = (cyc - cyc0) % kCyclesPerLine;
        const int rem = kCyclesPerLine - into;
        if (rem < kCyclesPerLine) cyc += rem;
        break;
      }
      case 0x06: case 0x07: case 0x08: case 0x09: tset(reg, v & 0xfe); break;
      case 0x10: tset(T_POSP0, hpos < kHBlank ? 3 : wrap160(hpos - kHBlank + 5)); tset(T_SUP0, 1); break;
      case 0x11: tset(T_POSP1, hpos < kHBlank ? 3 : wrap160(hpos - kHBlank + 5)); tset(T_SUP1, 1); break;
      case 0x12: tset(T_POSM0, hpos < kHBlank ? 2 : wrap160(hpos - kHBlank + 4)); break;
      case 0x13: tset(T_POSM1, hpos < kHBlank ? 2 : wrap160(hpos - kHBlank + 4)); break;
      case 0x14: tset(T_POSBL, hpos < kHBlank ? 2 : wrap160(hpos - kHBlank + 4)); break;
      case 0x1b: tset(T_GRP0, v); tset(T_DGRP1, t(T_GRP1)); break;
      case 0x1c: tset(T_GRP1, v); tset(T_DGRP0, t(T_GRP0)); tset(T_DENABL, t(T_ENABL)); break;
      case 0x28:
        if ((t(T_RESMP0) & 2) && !(v & 2)) tset(T_POSM0, wrap160(t(T_POSP0) + missile_center(t(T_NUSIZ0))));
        tset(T_RESMP0, v);
        break;
      case 0x29:
        if ((t(T_RESMP1) & 2) && !(v & 2)) tset(T_POSM1, wrap160(t(T_POSP1) + missile_center(t(T_NUSIZ1))));
        tset(T_RESMP1, v);
        break;
      case 0x2a:
        if (hpos / 3 < 21) tset(T_HMBLANK, 1);
        tset(T_POSP0, wrap160(t(T_POSP0) + hm_motion(t(T_HMP0))));
        tset(T_POSP1, wrap160(t(T_POSP1) + hm_motion(t(T_HMP1))));
        tset(T_POSM0, wrap160(t(T_POSM0) + hm_motion(t(T_HMM0))));
        tset(T_POSM1, wrap160(t(T_POSM1) + hm_motion(t(T_HMM1))));
        tset(T_POSBL, wrap160(t(T_POSBL) + hm_motion(t(T_HMBL))));
        break;
      case 0x2b:
        tset(T_HMP0, 0); tset(T_HMP1, 0); tset(T_HMM0, 0); tset(T_HMM1, 0); tset(T_HMBL, 0);
        break;
      case 0x2c: cx = 0; break;
      case 0x03: break;  // RSYNC not modelled
      case 0x04: case 0x05: case 0x0a: case 0x0b: case 0x0c: case 0x0d: case 0x0e: case 0x0f:
      case 0x1d: case 0x1e: case 0x1f: case 0x20: case 0x21: case 0x22: case 0x23: case 0x24:
      case 0x25: case 0x26: case 0x27:
        tset(reg, v);
        break;
      default: break;  // audio, unmapped
    }
  }

  DEVI int paddle_inpt(int r) const {
    const int needed = (int)(((long long)r * 1904) / 100000);
    return (!dump_en && cyc > dump_dis_cyc + needed) ? 0x80 : 0x00;
  }

  DEVI int tia_read(int reg, int noise) const {
    int v = 0;
    const int c = cx;
    switch (reg & 0x0f) {
      case 0x0: v = ((c & CX_M0P1) ? 0x80 : 0) | ((c & CX_M0P0) ? 0x40 : 0); break;
      case 0x1: v = ((c & CX_M1P0) ? 0x80 : 0) | ((c & CX_M1P1) ? 0x40 : 0); break;
      case 0x2: v = ((c & CX_P0PF) ? 0x80 : 0) | ((c & CX_P0BL) ? 0x40 : 0); break;
      case 0x3: v = ((c & CX_P1PF) ? 0x80 : 0) | ((c & CX_P1BL) ? 0x40 : 0); break;
      case 0x4: v = ((c & CX_M0PF) ? 0x80 : 0) | ((c & CX_M0BL) ? 0x40 : 0); break;
      case 0x5: v = ((c & CX_M1PF) ? 0x80 : 0) | ((c & CX_M1BL) ? 0x40 : 0); break;
      case 0x6: v = (c & CX_BLPF) ? 0x80 : 0; break;
      case 0x7: v = ((c & CX_P0P1) ? 0x80 : 0) | ((c & CX_M0M1) ? 0x40 : 0); break;
      case 0x8: v = paddle_inpt(paddle_res0); break;
      case 0x9: v = paddle_inpt(paddle_res1); break;
      case 0xa: case 0xb: v = dump_en ? 0x00 : 0x80; break;
      case 0xc: case 0xd: v = 0x80; break;
      default: v = 0; break;
    }
    return (v & 0xc0) | (noise & 0x3f);
  }

  DEVI int riot_read(int addr) const {
    if (!(addr & 0x04)) {
      const int r = addr & 3;
      if (r == 0) {
        int v = 0xff;
        v &= fire0 ? 0x7f : 0xff;
        v &= fire1 ? 0xbf : 0xff;
        return ((v & ~ddra) | (swcha_out & ddra)) & 0xff;
      }
      if (r == 1) return ddra;
      if (r == 2) return sw_reset ? 0x0a : 0x0b;
      return ddrb;
    }
    const int delta = (cyc - 1) - timer_set_cyc;
    int tt = timer - (delta >> timer_shift) - 1;
    if (!(addr & 1)) {
      if (tt >= 0) return tt & 0xff;
      tt = (timer << timer_shift) - delta - 1;
      return tt & 0xff;
    }
    return tt >= 0 ? 0x00 : 0x80;
  }

  DEVI void riot_write(int addr, int v) {
    if ((addr & 0x14) == 0x14) {
      const int r = addr & 3;
      timer = v;
      timer_shift = r == 0 ? 0 : (r == 1 ? 3 : (r == 2 ? 6 : 10));
      timer_set_cyc = cyc;
    } else if (!(addr & 0x04)) {
      const int r = addr & 3;
      if (r == 0) swcha_out = v; else if (r == 1) ddra = v; else if (r == 2) swchb_out = v; else ddrb = v;
    }
  }

  // zero-page pointer byte for (zp,X) / (zp),Y
  DEVI int zp_ptr(int p) {
    cyc++;
    if (!(p & 0x80)) { jam |= JAM_ZP_PTR_TIA; return 0; }
    return ram_rd(p & 0x7f);
  }
  DEVI void push(int v) {
    cyc++;
    if (S < 0x80) jam |= JAM_STACK;
    ram_wr(S & 0x7f, v & 0xff);
    S = (S - 1) & 0xff;
  }
  DEVI int pull() {
    S = (S + 1) & 0xff;
    cyc++;
    if (S < 0x80) jam |= JAM_STACK;
    return ram_rd(S & 0x7f);
  }

  DEVI void adc(int m) {
    const int old = A, c = P & FC;
    if (P & FD) {
      const int sum = ((A >> 4) * 10 + (A & 15)) + ((m >> 4) * 10 + (m & 15)) + c;
      P = (P & ~FC) | (sum > 99 ? FC : 0);
      const int tt = sum & 0xff;
      A = (((tt % 100) / 10) << 4) | (tt % 10);
    } else {
      const int sum = A + m + c;
      P = (P & ~FC) | (sum > 0xff ? FC : 0);
      A = sum & 0xff;
    }
    set_nz(A);
    P = (P & ~FV) | ((~(old ^ m) & (old ^ A) & 0x80) ? FV : 0);
  }
  DEVI void sbc(int m) {
    const int old = A, borrow = (P & FC) ? 0 : 1;
    if (P & FD) {
      int diff = ((A >> 4) * 10 + (A & 15)) - ((m >> 4) * 10 + (m & 15)) - borrow;
      diff += diff < 0 ? 100 : 0;
      A = ((((diff % 100) / 10) << 4) | (diff % 10)) & 0xff;
    } else {
      A = (A - m - borrow) & 0xff;
    }
    P = (P & ~FC) | ((old >= m + borrow) ? FC : 0);
    set_nz(A);
    P = (P & ~FV) | (((old ^ m) & (old ^ A) & 0x80) ? FV : 0);
  }
  DEVI void cmp(int r, int m) {
    P = (P & ~FC) | (r >= m ? FC : 0);
    set_nz((r - m) & 0xff);
  }

  // ------------------------------------------------------------------------------------
  // Fast path: instructions whose operand lives in the cartridge, the 128 bytes of RAM or (reads
  // only) the RIOT — ~85 % of what Pong / Breakout execute.  They need none of the generic path's
  // bus staging (effective-address / TIA catch-up / operand / write-back stages), only the
  // cycle count of the documented instruction timing.  Returns false WITHOUT side effects when the
  // instruction touches the TIA or is a rare one; the generic path below then executes it.
  // ------------------------------------------------------------------------------------
  enum : int { FAST_DONE = 0, FAST_GENERIC = 1, FAST_TIA_STORE = 2 };
  DEVI int step_fast(const uint32_t w, int& tia_ea, int& tia_wv) {
    const int b1 = w & 0xff, b2 = (w >> 8) & 0xff;
    const int mode = (w >> 16) & 15, kind = (w >> 20) & 3, op = (w >> 22) & 63;
    if (kind == K_READ) {
      int m = 0, dc = 0, dpc = 2;
      switch (mode) {
        case M_IMM: m = b1; dc = 2; break;
        case M_ZP:
          if (!(b1 & 0x80)) return FAST_GENERIC;
          m = ram_rd(b1 & 0x7f); dc = 3;
          break;
        case M_ZPX: case M_ZPY: {
          const int ea = (b1 + (mode == M_ZPX ? X : Y)) & 0xff;
          if (!(ea & 0x80)) return FAST_GENERIC;
          m = ram_rd(ea & 0x7f); dc = 4;
          break;
        }
        case M_ABS: case M_ABX: case M_ABY: {
          const int base = b1 | (b2 << 8);
          const int ea = mode == M_ABS ? base : ((base + (mode == M_ABX ? X : Y)) & 0xffff);
          dc = 4 + (((ea ^ base) & 0xff00) ? 1 : 0);
          dpc = 3;
          if (ea & 0x1000) {
            m = rom_byte(ea);
          } else if ((ea & 0x280) == 0x80) {
            m = ram_rd(ea & 0x7f);
          } else if ((ea & 0x280) == 0x280) {
            cyc += dc; dc = 0;      // the timer is read at the bus cycle's time
            m = riot_read(ea);
          } else if ((ea & 0x0f) >= 8) {
            // TIA input ports (Pong polls INPT0-3 with LDA abs,Y ~185x per frame): no picture
            // dependence; same bus timing as the generic path (value sampled at the read cycle,
            // bus noise of an absolute read = the address high byte)
            cyc += dc; dc = 0;
            m = tia_read(ea, b2);
          } else {
            return FAST_GENERIC;    // collision latches: the picture must be caught up first
          }
          break;
        }
        case M_IZY: {
          if (b1 < 0x80 || b1 == 0xff) return FAST_GENERIC;  // pointer bytes must both be in RAM
          const int base = ram_rd(b1 & 0x7f) | (ram_rd((b1 + 1) & 0x7f) << 8);
          const int ea = (base + Y) & 0xffff;
          dc = 5 + (((ea ^ base) & 0xff00) ? 1 : 0);
          if (ea & 0x1000) m = rom_byte(ea);
          else if ((ea & 0x280) == 0x80) m = ram_rd(ea & 0x7f);
          else return FAST_GENERIC;
          break;
        }
        default: return FAST_GENERIC;  // (zp,X) and the pulls
      }
      switch (op) {
        case O_LDA: A = m; set_nz(A); break;
        case O_LDX: X = m; set_nz(X); break;
        case O_LDY: Y = m; set_nz(Y); break;
        case O_ORA: A |= m; set_nz(A); break;
        case O_AND: A &= m; set_nz(A); break;
        case O_EOR: A ^= m; set_nz(A); break;
        case O_ADC: adc(m); break;
        case O_SBC: sbc(m); break;
        case O_CMP: cmp(A, m); break;
        case O_CPX: cmp(X, m); break;
        case O_CPY: cmp(Y, m); break;
        case O_BIT: P = (P & ~(FN | FV | FZ)) | (m & 0xc0) | ((A & m) ? 0 : FZ); break;
        default: jam |= JAM_OPCODE; break;  // unreachable: every K_READ op of these modes is above
      }
      cyc += dc;
      PC = (PC + dpc) & 0xffff;
      return FAST_DONE;
    }
    if (kind == K_NONE) {
      int dc = 2, npc = (PC + 1) & 0xffff;
      switch (op) {
        case O_ASL_A: P = (P & ~FC) | (A >> 7); A = (A << 1) & 0xff; set_nz(A); break;
        case O_LSR_A: P = (P & ~FC) | (A & 1); A = A >> 1; set_nz(A); break;
        case O_ROL_A: { const int c = P & FC; P = (P & ~FC) | (A >> 7); A = ((A << 1) | c) & 0xff; set_nz(A); break; }
        case O_ROR_A: { const int c = P & FC; P = (P & ~FC) | (A & 1); A = (A >> 1) | (c << 7); set_nz(A); break; }
        case O_INX: X = (X + 1) & 0xff; set_nz(X); break;
        case O_INY: Y = (Y + 1) & 0xff; set_nz(Y); break;
        case O_DEX: X = (X - 1) & 0xff; set_nz(X); break;
        case O_DEY: Y = (Y - 1) & 0xff; set_nz(Y); break;
        case O_TAX: X = A; set_nz(X); break;
        case O_TAY: Y = A; set_nz(Y); break;
        case O_TXA: A = X; set_nz(A); break;
        case O_TYA: A = Y; set_nz(A); break;
        case O_TSX: X = S; set_nz(X); break;
        case O_TXS: S = X; break;
        case O_CLC: P &= ~FC; break;
        case O_SEC: P |= FC; break;
        case O_CLI: P &= ~FI; break;
        case O_SEI: P |= FI; break;
        case O_CLV: P &= ~FV; break;
        case O_CLD: P &= ~FD; break;
        case O_SED: P |= FD; break;
        case O_NOP: break;
        case O_BPL: case O_BMI: case O_BVC: case O_BVS: case O_BCC: case O_BCS: case O_BNE: case O_BEQ: {
          const int flag = (op == O_BPL || op == O_BMI) ? FN : ((op == O_BVC || op == O_BVS) ? FV :
                           ((op == O_BCC || op == O_BCS) ? FC : FZ));
          const bool want_set = (op == O_BMI || op == O_BVS || op == O_BCS || op == O_BEQ);
          npc = (PC + 2) & 0xffff;
          if (((P & flag) != 0) == want_set) {
            const int tgt = (npc + ((b1 & 0x80) ? b1 - 256 : b1)) & 0xffff;
            dc += ((tgt ^ npc) & 0xff00) ? 2 : 1;
            npc = tgt;
          }
          break;
        }
        case O_JMP: dc = 3; npc = b1 | (b2 << 8); break;
        default: return FAST_GENERIC;  // JSR / RTS / RTI / BRK / JMP () / JAM
      }
      cyc += dc;
      PC = npc;
      return FAST_DONE;
    }
    // stores and read-modify-writes to RAM
    int ea, dc, dpc = 2;
    switch (mode) {
      case M_ZP: ea = b1; dc = 3; break;
      case M_ZPX: ea = (b1 + X) & 0xff; dc = 4; break;
      case M_ZPY: ea = (b1 + Y) & 0xff; dc = 4; break;
      case M_PUSH: ea = S; dc = 3; dpc = 1; break;
      default: return FAST_GENERIC;  // absolute / indirect stores (RIOT timer writes etc.)
    }
    int wv;
    if (kind == K_WRITE) {
      wv = op == O_STA ? A : (op == O_STX ? X : (op == O_STY ? Y : (op == O_PHA ? A : (P | FB | FU))));
      if (mode == M_PUSH) S = (S - 1) & 0xff;
      if (!(ea & 0x80)) {
        // TIA register write (incl. the PHP-into-ENABL stack trick), ~1000x per frame: hand the
        // generic path a finished effective address / value so that it only runs its TIA stages
        // (the catch-up render keeps a single site in the instruction stream)
        cyc += dc - 1;  // the write cycle itself is counted by stage D
        PC = (PC + dpc) & 0xffff;
        tia_ea = ea & 0xff; tia_wv = wv;
        return FAST_TIA_STORE;
      }
    } else {  // K_RMW: zp / zp,X only reach here
      if (!(ea & 0x80)) return FAST_GENERIC;  // read-modify-write of a TIA address: flagged there
      const int m = ram_rd(ea & 0x7f);
      switch (op) {
        case O_ASL: P = (P & ~FC) | (m >> 7); wv = (m << 1) & 0xff; break;
        case O_LSR: P = (P & ~FC) | (m & 1); wv = m >> 1; break;
        case O_ROL: { const int c = P & FC; P = (P & ~FC) | (m >> 7); wv = ((m << 1) | c) & 0xff; break; }
        case O_ROR: { const int c = P & FC; P = (P & ~FC) | (m & 1); wv = (m >> 1) | (c << 7); break; }
        case O_INC: wv = (m + 1) & 0xff; break;
        default: wv = (m - 1) & 0xff; break;  // O_DEC
      }
      set_nz(wv);
      dc += 2;
    }
    ram_wr(ea & 0x7f, wv);
    cyc += dc;
    PC = (PC + dpc) & 0xffff;
    return FAST_DONE;
  }

  // ------------------------------------------------------------------------------------
  // one 6507 instruction
  // ------------------------------------------------------------------------------------
  DEVI void step() {
    const uint32_t w = (uint32_t)rfl((int)romw[PC & rom_mask]);
    if (!(PC & 0x1000)) jam |= JAM_OPCODE;  // executing outside the cartridge is not supported
    int ea = 0, wv = 0;
    const int fast = step_fast(w, ea, wv);
    if (fast == FAST_DONE) return;
    const bool pre = fast == FAST_TIA_STORE;  // stages A and C already done by the fast decoder
    const int b1 = w & 0xff, b2 = (w >> 8) & 0xff;
    const int mode = (w >> 16) & 15, kind = (w >> 20) & 3, op = (w >> 22) & 63;
    int noise = b1, m = 0;
    // ---- stage A: effective address ----
    if (!pre) {
    cyc++;  // opcode fetch
    switch (mode) {
      case M_IMM: cyc++; m = b1; PC = (PC + 2) & 0xffff; break;
      case M_ZP: cyc++; ea = b1; PC = (PC + 2) & 0xffff; break;
      case M_ZPX: cyc += 2; ea = (b1 + X) & 0xff; PC = (PC + 2) & 0xffff; break;
      case M_ZPY: cyc += 2; ea = (b1 + Y) & 0xff; PC = (PC + 2) & 0xffff; break;
      case M_ABS: cyc += 2; ea = b1 | (b2 << 8); noise = b2; PC = (PC + 3) & 0xffff; break;
      case M_ABX: case M_ABY: {
        cyc += 2;
        const int base = b1 | (b2 << 8);
        ea = (base + (mode == M_ABX ? X : Y)) & 0xffff;
        if (kind != K_READ || ((ea ^ base) & 0xff00)) cyc++;
        noise = b2;
        PC = (PC + 3) & 0xffff;
        break;
      }
      case M_IZX: {
        cyc += 2;
        const int p = (b1 + X) & 0xff;
        const int lo = zp_ptr(p), hi = zp_ptr((p + 1) & 0xff);
        ea = lo | (hi << 8);
        noise = hi;
        PC = (PC + 2) & 0xffff;
        break;
      }
      case M_IZY: {
        cyc++;
        const int lo = zp_ptr(b1), hi = zp_ptr((b1 + 1) & 0xff);
        const int base = lo | (hi << 8);
        ea = (base + Y) & 0xffff;
        if (kind != K_READ || ((ea ^ base) & 0xff00)) cyc++;
        noise = hi;
        PC = (PC + 2) & 0xffff;
        break;
      }
      case M_PUSH: cyc++; ea = 0x100 | S; PC = (PC + 1) & 0xffff; break;
      case M_PULL:
        // two dummy reads (next opcode byte = b1, then the old stack top) precede the pull; when
        // the stack sits in TIA space (Breakout: PHP/PLA at S=$1F) the old-top read returns
        // (latches & 0xc0) | (b1 & 0x3f), so the bus noise seen by the pull is b1 & 0x3f
        cyc += 2; S = (S + 1) & 0xff; ea = 0x100 | S; noise = b1; PC = (PC + 1) & 0xffff;
        if (S == 0) jam |= JAM_STACK;  // old top in RAM, new top in TIA: noise would be the RAM byte
        break;
      default: break;  // M_IMP / M_REL handled by the operation
    }
    }
    const bool has_ea = kind != K_NONE && mode != M_IMM;
    const bool is_tia = has_ea && !(ea & 0x1080);
    // ---- stage R: catch the picture up before a TIA access (single render site) ----
    // Writes are classified first: rewriting a register with the value it already holds (what the
    // cartridges do on almost every scanline) neither needs the picture nor changes anything.
    enum : int { W_GENERAL = 0, W_NOP, W_PLAIN };
    int upd = -1, wkind = W_GENERAL, wreg = 0, wval = 0;
    if (is_tia) {
      if (kind == K_READ) {
        // only the collision latches (CXxx, 0x0-0x7) depend on the picture; INPTx do not
        if ((ea & 0x0f) < 8) upd = (cyc + 1) * 3;
      } else if (kind == K_WRITE) {
        wreg = ea & 0x3f;
        if (!pre) wv = op == O_STA ? A : (op == O_STX ? X : (op == O_STY ? Y : (op == O_PHA ? A : (P | FB | FU))));
        const int clock = (cyc + 1) * 3;
        if ((kPlainRegs >> wreg) & 1ull) {
          wval = (wreg >= 0x06 && wreg <= 0x09) ? (wv & 0xfe) : wv;
          if (t(wreg) == wval) {
            wkind = W_NOP;
          } else {
            wkind = W_PLAIN;
            upd = clock + poke_delay(wreg, (clock - cyc0 * 3) % kClocksPerLine);
          }
        } else if (wreg == 0x1b) {
          if (t(T_GRP0) == wv && t(T_DGRP1) == t(T_GRP1)) wkind = W_NOP; else upd = clock + 1;
        } else if (wreg == 0x1c) {
          if (t(T_GRP1) == wv && t(T_DGRP0) == t(T_GRP0) && t(T_DENABL) == t(T_ENABL)) wkind = W_NOP;
          else upd = clock + 1;
        } else if ((kStrobeRegs >> wreg) & 1ull) {
          upd = clock;  // position strobes, RESMPx, HMOVE, CXCLR (poke delay 0)
        }
      } else {
        jam |= JAM_RMW_TIA;
      }
    }
    if (upd >= 0) tia_update(upd);
    // ---- stage B: operand read ----
    if (has_ea && kind != K_WRITE) {
      cyc++;
      if (ea & 0x1000) m = rom_byte(ea);
      else if (!(ea & 0x80)) m = tia_read(ea, noise);
      else if (!(ea & 0x200)) m = ram_rd(ea & 0x7f);
      else m = riot_read(ea);
    }
    // ---- stage C: operation ----
    if (!pre) {
    switch (op) {
      case O_ORA: A |= m; set_nz(A); break;
      case O_AND: A &= m; set_nz(A); break;
      case O_EOR: A ^= m; set_nz(A); break;
      case O_ADC: adc(m); break;
      case O_SBC: sbc(m); break;
      case O_CMP: cmp(A, m); break;
      case O_CPX: cmp(X, m); break;
      case O_CPY: cmp(Y, m); break;
      case O_LDA: A = m; set_nz(A); break;
      case O_LDX: X = m; set_nz(X); break;
      case O_LDY: Y = m; set_nz(Y); break;
      case O_STA: wv = A; break;
      case O_STX: wv = X; break;
      case O_STY: wv = Y; break;
      case O_BIT: P = (P & ~(FN | FV | FZ)) | (m & 0xc0) | ((A & m) ? 0 : FZ); break;
      case O_ASL: P = (P & ~FC) | (m >> 7); wv = (m << 1) & 0xff; set_nz(wv); break;
      case O_LSR: P = (P & ~FC) | (m & 1); wv = m >> 1; set_nz(wv); break;
      case O_ROL: { const int c = P & FC; P = (P & ~FC) | (m >> 7); wv = ((m << 1) | c) & 0xff; set_nz(wv); break; }
      case O_ROR: { const int c = P & FC; P = (P & ~FC) | (m & 1); wv = (m >> 1) | (c << 7); set_nz(wv); break; }
      case O_INC: wv = (m + 1) & 0xff; set_nz(wv); break;
      case O_DEC: wv = (m - 1) & 0xff; set_nz(wv); break;
      case O_ASL_A: cyc++; P = (P & ~FC) | (A >> 7); A = (A << 1) & 0xff; set_nz(A); PC = (PC + 1) & 0xffff; break;
      case O_LSR_A: cyc++; P = (P & ~FC) | (A & 1); A = A >> 1; set_nz(A); PC = (PC + 1) & 0xffff; break;
      case O_ROL_A: { cyc++; const int c = P & FC; P = (P & ~FC) | (A >> 7); A = ((A << 1) | c) & 0xff; set_nz(A); PC = (PC + 1) & 0xffff; break; }
      case O_ROR_A: { cyc++; const int c = P & FC; P = (P & ~FC) | (A & 1); A = (A >> 1) | (c << 7); set_nz(A); PC = (PC + 1) & 0xffff; break; }
      case O_INX: cyc++; X = (X + 1) & 0xff; set_nz(X); PC = (PC + 1) & 0xffff; break;
      case O_INY: cyc++; Y = (Y + 1) & 0xff; set_nz(Y); PC = (PC + 1) & 0xffff; break;
      case O_DEX: cyc++; X = (X - 1) & 0xff; set_nz(X); PC = (PC + 1) & 0xffff; break;
      case O_DEY: cyc++; Y = (Y - 1) & 0xff; set_nz(Y); PC = (PC + 1) & 0xffff; break;
      case O_TAX: cyc++; X = A; set_nz(X); PC = (PC + 1) & 0xffff; break;
      case O_TAY: cyc++; Y = A; set_nz(Y); PC = (PC + 1) & 0xffff; break;
      case O_TXA: cyc++; A = X; set_nz(A); PC = (PC + 1) & 0xffff; break;
      case O_TYA: cyc++; A = Y; set_nz(A); PC = (PC + 1) & 0xffff; break;
      case O_TSX: cyc++; X = S; set_nz(X); PC = (PC + 1) & 0xffff; break;
      case O_TXS: cyc++; S = X; PC = (PC + 1) & 0xffff; break;
      case O_CLC: cyc++; P &= ~FC; PC = (PC + 1) & 0xffff; break;
      case O_SEC: cyc++; P |= FC; PC = (PC + 1) & 0xffff; break;
      case O_CLI: cyc++; P &= ~FI; PC = (PC + 1) & 0xffff; break;
      case O_SEI: cyc++; P |= FI; PC = (PC + 1) & 0xffff; break;
      case O_CLV: cyc++; P &= ~FV; PC = (PC + 1) & 0xffff; break;
      case O_CLD: cyc++; P &= ~FD; PC = (PC + 1) & 0xffff; break;
      case O_SED: cyc++; P |= FD; PC = (PC + 1) & 0xffff; break;
      case O_NOP: cyc++; PC = (PC + 1) & 0xffff; break;
      case O_JAM: cyc++; jam |= JAM_OPCODE; PC = (PC + 1) & 0xffff; break;
      case O_PHA: wv = A; S = (S - 1) & 0xff; break;
      case O_PHP: wv = P | FB | FU; S = (S - 1) & 0xff; break;
      case O_PLA: A = m; set_nz(A); break;
      case O_PLP: P = (m & ~FB) | FU; break;
      case O_BPL: case O_BMI: case O_BVC: case O_BVS: case O_BCC: case O_BCS: case O_BNE: case O_BEQ: {
        const int flag = (op == O_BPL || op == O_BMI) ? FN : ((op == O_BVC || op == O_BVS) ? FV :
                         ((op == O_BCC || op == O_BCS) ? FC : FZ));
        const bool want_set = (op == O_BMI || op == O_BVS || op == O_BCS || op == O_BEQ);
        const bool taken = ((P & flag) != 0) == want_set;
        cyc++;
        PC = (PC + 2) & 0xffff;
        if (taken) {
          const int off = (b1 & 0x80) ? b1 - 256 : b1;
          const int tgt = (PC + off) & 0xffff;
          cyc += ((tgt ^ PC) & 0xff00) ? 2 : 1;
          PC = tgt;
        }
        break;
      }
      case O_JMP: cyc += 2; PC = b1 | (b2 << 8); break;
      case O_JMPI: {
        cyc += 4;
        const int p = b1 | (b2 << 8);
        const int p2 = (p & 0xff00) | ((p + 1) & 0xff);
        int lo = 0, hi = 0;
        if (p & 0x1000) { lo = rom_byte(p); hi = rom_byte(p2); }
        else if ((p & 0x80) && !(p & 0x200)) { lo = ram_rd(p & 0x7f); hi = ram_rd(p2 & 0x7f); }
        else jam |= JAM_JMPI;
        PC = lo | (hi << 8);
        break;
      }
      case O_JSR: {
        cyc += 2;  // operand low + internal
        const int ret = (PC + 2) & 0xffff;
        push(ret >> 8);
        push(ret & 0xff);
        cyc++;     // operand high
        PC = b1 | (b2 << 8);
        break;
      }
      case O_RTS: { cyc += 2; const int lo = pull(), hi = pull(); PC = ((lo | (hi << 8)) + 1) & 0xffff; cyc++; break; }
      case O_RTI: { cyc += 2; P = (pull() & ~FB) | FU; const int lo = pull(), hi = pull(); PC = lo | (hi << 8); break; }
      case O_BRK: {
        cyc++;
        const int ret = (PC + 2) & 0xffff;
        push(ret >> 8);
        push(ret & 0xff);
        push(P | FB | FU);
        P |= FI;
        cyc += 2;
        PC = rom_byte(0xfffe) | (rom_byte(0xffff) << 8);
        break;
      }
      default: break;
    }
    }
    // ---- stage D: write back ----
    if (has_ea && kind != K_READ) {
      if (kind == K_RMW) cyc++;  // internal modify cycle
      cyc++;
      if (ea & 0x1000) {
      } else if (!(ea & 0x80)) {
        if (kind == K_WRITE) {
          if (wkind == W_NOP) {
          } else if (wkind == W_PLAIN) {
            if (wreg == 0x01) {  // VBLANK: paddle dump transistor
              const int old = t(T_VBLANK);
              if (!(old & 0x80) && (wval & 0x80)) dump_en = 1;
              if ((old & 0x80) && !(wval & 0x80)) { dump_en = 0; dump_dis_cyc = cyc; }
            }
            tset(wreg, wval);
          } else if (wreg == 0x02) {  // WSYNC: halt the CPU until the end of the scanline
            const int into = (cyc - cyc0) % kCyclesPerLine;
            cyc += into ? kCyclesPerLine - into : 0;
          } else {
            tia_write(wreg, wv);
          }
        }
      } else if (!(ea & 0x200)) {
        ram_wr(ea & 0x7f, wv);
      } else {
        riot_write(ea, wv);
      }
    }
  }

  // Stella TIA::update(): one frame = until VSYNC is released or 25000 instructions
  DEVI void frame(uint8_t* frame_buffer) {
    const int into = (cyc - cyc0) % kCyclesPerLine;
    const int old = cyc;
    cyc = 0;
    cyc0 = -into;
    timer_set_cyc -= old;
    dump_dis_cyc -= old;
    if (vsync_finish != 0x7fffffff) vsync_finish -= old * 3;
    last_clock = cyc0 * 3 + kClocksPerLine * kYStart;
    fb = frame_buffer;
    stop = 0;
    for (int n = 0; n < kMaxInstrPerFrame && !stop; ++n) step();
    fb = nullptr;
  }

  DEVI void system_reset() {
    A = X = Y = 0; S = 0xff; P = FU | FI;
    PC = rom_byte(0xfffc) | (rom_byte(0xfffd) << 8);
    cyc = cyc0 = last_clock = 0;
    vsync_finish = 0x7fffffff;
    dump_dis_cyc = dump_en = 0;
    timer = 0; timer_shift = 10; timer_set_cyc = 0;
    ddra = ddrb = swcha_out = swchb_out = cx = jam = stop = 0;
    paddle_res0 = paddle_res1 = 408823;
    fire0 = fire1 = sw_reset = 0;
    ram_lo = ram_hi = tia = 0;
  }
};

}  // namespace atari
}  // namespace parlhip

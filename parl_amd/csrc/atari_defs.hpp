// atari_defs.hpp — shared host/device definitions of the gfx950 Atari 2600 env kernels:
// per-env state blob layout in HBM, the pre-decoded ROM word table, opcode decode table.
#pragma once
#include <stdint.h>

namespace parlhip {
namespace atari {

constexpr int kW = 160, kH = 210, kYStart = 34, kFrameBytes = kW * kH;
constexpr int kHBlank = 68, kClocksPerLine = 228, kCyclesPerLine = 76;
// the observation tail (frame_tail.hpp): the picture's 210 rows are 42 bands of 5; the env's two waves claim them in
// two chunks from one counter (RenderQueue::obs_next, atari_core.hpp) and convert a chunk in calls of a
// multiple of kObsStep bands, as far as the picture wave has declared them final (a call's first band waits a whole
// memory round trip, ~3 k clocks, for its pixels — the frame pair of 1024 envs is 69 MB — so calls are long)
// Two chunks: the first claim takes bands 0 .. kObsFirst - 1, the second the rest.  The first claimer is wave A at its
// last instruction unless the picture wave ran out of records during the frame's overscan (it rarely does: through
// the two rendered frames of a step it is the slower wave, ~20 k clocks behind at wave A's exit).  Measured at 1024
// Pong envs (round 6, emu_bench): first chunk 12 / 15 / 18 bands 9.43 M frames/s, 21: 9.39, 27: 9.31, 30: 9.28 — a
// launch lasts as long as its slowest env, not the mean one.
#ifndef PARLHIP_OBS_STEP
#define PARLHIP_OBS_STEP 3
#endif
#ifndef PARLHIP_OBS_FIRST
#define PARLHIP_OBS_FIRST 18
#endif
constexpr int kObsBandRows = 5, kObsBands = kH / kObsBandRows, kObsFirst = PARLHIP_OBS_FIRST, kObsStep = PARLHIP_OBS_STEP;
static_assert(kObsBands % kObsStep == 0 && kObsFirst % kObsStep == 0 && kObsFirst > 0 && kObsFirst < kObsBands, "whole steps");
constexpr int kMaxInstrPerFrame = 25000;  // Stella: m6502().execute(25000)

// ---- addressing modes / access kinds / operations of the pre-decoded instruction word ----
enum Mode : int { M_IMP = 0, M_IMM, M_ZP, M_ZPX, M_ZPY, M_ABS, M_ABX, M_ABY, M_IZX, M_IZY, M_REL,
                  M_PUSH, M_PULL };  // PHA/PHP and PLA/PLP go through the generic bus path (the
                                     // stack may point into TIA space: the PHP-to-ENABL trick)
enum Kind : int { K_NONE = 0, K_READ = 1, K_WRITE = 2, K_RMW = 3 };
enum Op : int {
  O_JAM = 0, O_NOP, O_ORA, O_AND, O_EOR, O_ADC, O_SBC, O_CMP, O_CPX, O_CPY, O_LDA, O_LDX, O_LDY,
  O_STA, O_STX, O_STY, O_BIT, O_ASL, O_LSR, O_ROL, O_ROR, O_INC, O_DEC, O_ASL_A, O_LSR_A, O_ROL_A,
  O_ROR_A, O_INX, O_INY, O_DEX, O_DEY, O_TAX, O_TAY, O_TXA, O_TYA, O_TSX, O_TXS, O_CLC, O_SEC,
  O_CLI, O_SEI, O_CLV, O_CLD, O_SED, O_PHA, O_PHP, O_PLA, O_PLP, O_BPL, O_BMI, O_BVC, O_BVS, O_BCC,
  O_BCS, O_BNE, O_BEQ, O_JMP, O_JMPI, O_JSR, O_RTS, O_RTI, O_BRK, O_COUNT
};
static_assert(O_COUNT <= 64, "operation id must fit 6 bits");

constexpr uint16_t enc(int mode, int kind, int op) { return (uint16_t)(mode | (kind << 4) | (op << 6)); }

// MOS 6502 documented opcode matrix -> (mode, kind, op).  Undocumented opcodes decode to O_JAM
// (executed as a flagged 2-cycle NOP, exactly as the CPU oracle does).
inline uint16_t decode_opcode(uint8_t op) {
  const int cc = op & 3, bbb = (op >> 2) & 7, aaa = op >> 5;
  static const int m01[8] = {M_IZX, M_ZP, M_IMM, M_ABS, M_IZY, M_ZPX, M_ABY, M_ABX};
  if (cc == 1) {
    static const int o01[8] = {O_ORA, O_AND, O_EOR, O_ADC, O_STA, O_LDA, O_CMP, O_SBC};
    if (op == 0x89) return enc(M_IMP, K_NONE, O_JAM);
    return enc(m01[bbb], aaa == 4 ? K_WRITE : K_READ, o01[aaa]);
  }
  switch (op) {
    case 0x00: return enc(M_IMP, K_NONE, O_BRK);
    case 0x20: return enc(M_IMP, K_NONE, O_JSR);
    case 0x40: return enc(M_IMP, K_NONE, O_RTI);
    case 0x60: return enc(M_IMP, K_NONE, O_RTS);
    case 0x4c: return enc(M_IMP, K_NONE, O_JMP);
    case 0x6c: return enc(M_IMP, K_NONE, O_JMPI);
    case 0x08: return enc(M_PUSH, K_WRITE, O_PHP);
    case 0x28: return enc(M_PULL, K_READ, O_PLP);
    case 0x48: return enc(M_PUSH, K_WRITE, O_PHA);
    case 0x68: return enc(M_PULL, K_READ, O_PLA);
    case 0x88: return enc(M_IMP, K_NONE, O_DEY);
    case 0xa8: return enc(M_IMP, K_NONE, O_TAY);
    case 0xc8: return enc(M_IMP, K_NONE, O_INY);
    case 0xe8: return enc(M_IMP, K_NONE, O_INX);
    case 0x18: return enc(M_IMP, K_NONE, O_CLC);
    case 0x38: return enc(M_IMP, K_NONE, O_SEC);
    case 0x58: return enc(M_IMP, K_NONE, O_CLI);
    case 0x78: return enc(M_IMP, K_NONE, O_SEI);
    case 0x98: return enc(M_IMP, K_NONE, O_TYA);
    case 0xb8: return enc(M_IMP, K_NONE, O_CLV);
    case 0xd8: return enc(M_IMP, K_NONE, O_CLD);
    case 0xf8: return enc(M_IMP, K_NONE, O_SED);
    case 0x8a: return enc(M_IMP, K_NONE, O_TXA);
    case 0x9a: return enc(M_IMP, K_NONE, O_TXS);
    case 0xaa: return enc(M_IMP, K_NONE, O_TAX);
    case 0xba: return enc(M_IMP, K_NONE, O_TSX);
    case 0xca: return enc(M_IMP, K_NONE, O_DEX);
    case 0xea: return enc(M_IMP, K_NONE, O_NOP);
    case 0x0a: return enc(M_IMP, K_NONE, O_ASL_A);
    case 0x2a: return enc(M_IMP, K_NONE, O_ROL_A);
    case 0x4a: return enc(M_IMP, K_NONE, O_LSR_A);
    case 0x6a: return enc(M_IMP, K_NONE, O_ROR_A);
    case 0x10: return enc(M_REL, K_NONE, O_BPL);
    case 0x30: return enc(M_REL, K_NONE, O_BMI);
    case 0x50: return enc(M_REL, K_NONE, O_BVC);
    case 0x70: return enc(M_REL, K_NONE, O_BVS);
    case 0x90: return enc(M_REL, K_NONE, O_BCC);
    case 0xb0: return enc(M_REL, K_NONE, O_BCS);
    case 0xd0: return enc(M_REL, K_NONE, O_BNE);
    case 0xf0: return enc(M_REL, K_NONE, O_BEQ);
    case 0x24: return enc(M_ZP, K_READ, O_BIT);
    case 0x2c: return enc(M_ABS, K_READ, O_BIT);
    case 0x84: return enc(M_ZP, K_WRITE, O_STY);
    case 0x94: return enc(M_ZPX, K_WRITE, O_STY);
    case 0x8c: return enc(M_ABS, K_WRITE, O_STY);
    case 0xa0: return enc(M_IMM, K_READ, O_LDY);
    case 0xa4: return enc(M_ZP, K_READ, O_LDY);
    case 0xb4: return enc(M_ZPX, K_READ, O_LDY);
    case 0xac: return enc(M_ABS, K_READ, O_LDY);
    case 0xbc: return enc(M_ABX, K_READ, O_LDY);
    case 0xc0: return enc(M_IMM, K_READ, O_CPY);
    case 0xc4: return enc(M_ZP, K_READ, O_CPY);
    case 0xcc: return enc(M_ABS, K_READ, O_CPY);
    case 0xe0: return enc(M_IMM, K_READ, O_CPX);
    case 0xe4: return enc(M_ZP, K_READ, O_CPX);
    case 0xec: return enc(M_ABS, K_READ, O_CPX);
    case 0x86: return enc(M_ZP, K_WRITE, O_STX);
    case 0x96: return enc(M_ZPY, K_WRITE, O_STX);
    case 0x8e: return enc(M_ABS, K_WRITE, O_STX);
    case 0xa2: return enc(M_IMM, K_READ, O_LDX);
    case 0xa6: return enc(M_ZP, K_READ, O_LDX);
    case 0xb6: return enc(M_ZPY, K_READ, O_LDX);
    case 0xae: return enc(M_ABS, K_READ, O_LDX);
    case 0xbe: return enc(M_ABY, K_READ, O_LDX);
    default: break;
  }
  if (cc == 2 && (bbb == 1 || bbb == 3 || bbb == 5 || bbb == 7) && aaa != 4 && aaa != 5) {
    static const int o10[8] = {O_ASL, O_ROL, O_LSR, O_ROR, 0, 0, O_DEC, O_INC};
    static const int m10[8] = {0, M_ZP, 0, M_ABS, 0, M_ZPX, 0, M_ABX};
    return enc(m10[bbb], K_RMW, o10[aaa]);
  }
  return enc(M_IMP, K_NONE, O_JAM);
}

// Pre-decoded ROM word at address a: b1 | b2<<8 | info<<16  (b1 = rom[a+1], b2 = rom[a+2]).
// The raw byte at a is the b1 field of word a-1.
inline void build_rom_words(const uint8_t* rom, uint32_t size, uint32_t* words) {
  const uint32_t mask = size - 1;
  for (uint32_t a = 0; a < size; ++a)
    words[a] = (uint32_t)rom[(a + 1) & mask] | ((uint32_t)rom[(a + 2) & mask] << 8) |
               ((uint32_t)decode_opcode(rom[a]) << 16);
}

// ---- per-env state blob in HBM (one per env, kStateBytes apart) ----
constexpr int kStateBytes = 512;
constexpr int kOffRam = 0;      // uint8[128]
constexpr int kOffTia = 128;    // uint8[64]: TIA write registers 0x00..0x2c + derived (below)
constexpr int kOffScalars = 192;  // int32[64]
// tia lane indices of derived state
enum TiaLane : int {
  T_VSYNC = 0x00, T_VBLANK = 0x01, T_NUSIZ0 = 0x04, T_NUSIZ1 = 0x05, T_COLUP0 = 0x06,
  T_COLUP1 = 0x07, T_COLUPF = 0x08, T_COLUBK = 0x09, T_CTRLPF = 0x0a, T_REFP0 = 0x0b,
  T_REFP1 = 0x0c, T_PF0 = 0x0d, T_PF1 = 0x0e, T_PF2 = 0x0f, T_GRP0 = 0x1b, T_GRP1 = 0x1c,
  T_ENAM0 = 0x1d, T_ENAM1 = 0x1e, T_ENABL = 0x1f, T_HMP0 = 0x20, T_HMP1 = 0x21, T_HMM0 = 0x22,
  T_HMM1 = 0x23, T_HMBL = 0x24, T_VDELP0 = 0x25, T_VDELP1 = 0x26, T_VDELBL = 0x27,
  T_RESMP0 = 0x28, T_RESMP1 = 0x29,
  T_POSP0 = 0x30, T_POSP1, T_POSM0, T_POSM1, T_POSBL, T_DGRP0, T_DGRP1, T_DENABL, T_SUP0, T_SUP1,
  T_HMBLANK
};
// scalar slots
enum Slot : int {
  S_A = 0, S_X, S_Y, S_S, S_P, S_PC, S_BUS, S_CYC, S_CYC0, S_LAST_CLOCK, S_VSYNC_FINISH,
  S_DUMP_DIS_CYC, S_DUMP_EN, S_TIMER, S_TIMER_SHIFT, S_TIMER_SET_CYC, S_DDRA, S_DDRB, S_SWCHA_OUT,
  S_SWCHB_OUT, S_CX, S_JAM,
  // ALE layer
  S_PADDLE, S_SCORE, S_TERMINAL, S_ALE_LIVES, S_STARTED, S_FRAME_NUMBER,
  // wrappers
  S_LIVES, S_WAS_REAL_DONE, S_HAS_EPISODE, S_CUR_REWARD, S_NUM_STEPS, S_ELAPSED, S_COMPAT_COUNT,
  S_RESET_COUNT, S_OBS_SINGLE, S_SINCE_RESET,
  // wrapper state machine suspended between launches (atari_env.hip, elastic stepping)
  S_SUSP, S_SUSP_TOTAL, S_SUSP_ACT, S_SUSP_ALE_J, S_SUSP_NOOPS, S_COUNT
};
static_assert(S_COUNT <= 64, "scalar slots");

enum Game : int { GAME_GENERIC = 0, GAME_PONG = 1, GAME_BREAKOUT = 2 };
enum AleAction : int { ACT_NOOP = 0, ACT_FIRE = 1, ACT_RIGHT = 3, ACT_LEFT = 4, ACT_RIGHTFIRE = 11,
                       ACT_LEFTFIRE = 12, ACT_RESET = 40 };

constexpr int kPaddleDelta = 23000, kPaddleMin = 27450, kPaddleMax = 790196;
constexpr int kPaddleDefault = ((kPaddleMax - kPaddleMin) / 2) + kPaddleMin;

}  // namespace atari
}  // namespace parlhip

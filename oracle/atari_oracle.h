/*
 * atari_oracle.h — CPU twin of the device Atari 2600 emulator + ALE/gym/wrapper semantics.
 * TEST INFRASTRUCTURE ONLY (see scan_oracle.c header).
 *
 * What this restates.  The reference's env step is third-party code outside its tree
 * (SURVEY.md §8c): gym 0.12.1 AtariEnv -> atari-py 0.1.7 (ALE 0.5/0.6, Stella 2.x core).
 * None of it is installed here, so this file restates the PUBLISHED behaviour of those
 * components as far as the reference's call sites rely on it:
 *   - MOS 6507 (documented 6502 instruction set, cycle counts per the MOS data sheet),
 *   - TIA as modelled by Stella 2.x's event-driven TIA (position/mask model, per-register
 *     poke delays, VSYNC-delimited frames, YStart=34, Height=210, collisions only inside the
 *     displayed window and outside VBLANK),
 *   - M6532 RIOT timer/ports, paddle capacitor timing (INPTx threshold = 1.6*R*0.01e-6*1.19e6
 *     CPU cycles after the dump is released),
 *   - ALE: one frame per act(), paddle resistance +/-23000 per frame in [27450,790196],
 *     reset = system reset + 60 NOOP frames + 4 RESET-switch frames, Pong/Breakout RomSettings
 *     (reward / terminal / lives from RAM), minimal action sets.
 * PARITY UNPINNED against ALE: neither ALE nor gym nor cv2 exist in this container (SURVEY
 * §7 hard parts); parity is pinned between this oracle and the HIP emulator (bit-exact), and
 * against the reference's call sites / wrapper code which ARE in the tree
 * (parl/env/atari_wrappers.py, vector_env.py, compat_wrappers.py).
 */
#ifndef ATARI_ORACLE_H_
#define ATARI_ORACLE_H_
#include <stdint.h>

#define ATARI_W 160
#define ATARI_H 210
#define ATARI_YSTART 34
#define ATARI_FRAME_BYTES (ATARI_W * ATARI_H)

enum { GAME_GENERIC = 0, GAME_PONG = 1, GAME_BREAKOUT = 2 };

/* ALE action codes used by the minimal action sets (ale Constants.h: PLAYER_A_*) */
enum { ACT_NOOP = 0, ACT_FIRE = 1, ACT_UP = 2, ACT_RIGHT = 3, ACT_LEFT = 4, ACT_DOWN = 5,
       ACT_RIGHTFIRE = 11, ACT_LEFTFIRE = 12, ACT_RESET = 40 };

typedef struct {
  /* 6507 */
  uint8_t A, X, Y, S, P;
  uint16_t PC;
  uint8_t bus; /* last value on the data bus (TIA read "noise") */
  uint8_t ram[128];
  const uint8_t* rom;
  uint32_t rom_mask;
  int32_t cyc;  /* CPU cycles since the frame's cycle origin (Stella resets per frame) */
  int32_t cyc0; /* cycle at which scanline 0 of this frame started (<= 0) */
  int stop;     /* frame complete (VSYNC released) */
  int jam;      /* undocumented opcode hit */
  /* TIA */
  uint8_t vsync, vblank, nusiz0, nusiz1, colup0, colup1, colupf, colubk, ctrlpf, refp0, refp1;
  uint8_t pf0, pf1, pf2, grp0, grp1, dgrp0, dgrp1, enam0, enam1, enabl, denabl;
  uint8_t vdelp0, vdelp1, vdelbl, resmp0, resmp1, hmp0, hmp1, hmm0, hmm1, hmbl;
  int16_t posp0, posp1, posm0, posm1, posbl;
  uint8_t sup0, sup1; /* suppress first player copy for the rest of this scanline */
  uint8_t hmove_blank;
  uint16_t cx; /* 15 collision latches */
  int32_t last_clock;  /* TIA rendered up to this color clock (relative to cyc origin*3) */
  int32_t vsync_finish_clock;
  int32_t dump_disabled_cyc;
  uint8_t dump_enabled;
  uint8_t inpt45_latch; /* unused for paddles; kept for joystick fire latches */
  /* RIOT */
  uint8_t timer, timer_shift;
  int32_t timer_set_cyc;
  uint8_t ddra, ddrb, swcha_out, swchb_out;
  /* console / controller inputs for the current frame */
  int32_t paddle_res[2]; /* Stella resistance units */
  uint8_t paddle_fire[2];
  uint8_t sw_reset, sw_select; /* 1 = pressed */
  /* where pixels go (NULL = collisions only) */
  uint8_t* fb;
} Atari;

void atari_init(Atari* a, const uint8_t* rom, uint32_t rom_size);
void atari_system_reset(Atari* a);
/* run one TIA frame (Stella TIA::update): until VSYNC is released or 25000 instructions */
void atari_frame(Atari* a, uint8_t* fb);

/* ---- ALE layer ---- */
typedef struct {
  Atari emu;
  int game;
  int32_t paddle; /* ALE m_left_paddle */
  int32_t score;  /* RomSettings m_score */
  int32_t reward; /* of the last act() */
  int terminal, lives, started;
  int64_t frame_number;
} Ale;

void ale_init(Ale* e, const uint8_t* rom, uint32_t rom_size, int game);
void ale_reset(Ale* e, uint8_t* fb);                 /* ALE reset_game() */
int32_t ale_act(Ale* e, int ale_action, uint8_t* fb); /* one frame; returns reward */
int ale_minimal_actions(int game, int* out);          /* returns count */

/* NTSC palette: TIA colour byte -> RGB */
extern const uint32_t atari_ntsc_palette[128];

#endif

/* atari_api.c — heap-allocating convenience wrappers so Python (ctypes) can drive the oracle
 * emulator without knowing struct layouts.  TEST INFRASTRUCTURE ONLY. */
#include "atari_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { Ale ale; uint8_t* rom; } AleBox;

void* oracle_ale_new(const uint8_t* rom, uint32_t size, int game) {
  AleBox* b = (AleBox*)calloc(1, sizeof(AleBox));
  b->rom = (uint8_t*)malloc(size);
  memcpy(b->rom, rom, size);
  ale_init(&b->ale, b->rom, size, game);
  return b;
}
void oracle_ale_free(void* p) { AleBox* b = (AleBox*)p; free(b->rom); free(b); }
void oracle_ale_reset(void* p, uint8_t* fb) { ale_reset(&((AleBox*)p)->ale, fb); }
int32_t oracle_ale_act(void* p, int action, uint8_t* fb) { return ale_act(&((AleBox*)p)->ale, action, fb); }
int oracle_ale_terminal(void* p) { return ((AleBox*)p)->ale.terminal; }
int oracle_ale_lives(void* p) { return ((AleBox*)p)->ale.lives; }
int oracle_ale_jam(void* p) { return ((AleBox*)p)->ale.emu.jam; }
void oracle_ale_ram(void* p, uint8_t* out) { memcpy(out, ((AleBox*)p)->ale.emu.ram, 128); }
int32_t oracle_ale_cycles(void* p) { return ((AleBox*)p)->ale.emu.cyc - ((AleBox*)p)->ale.emu.cyc0; }
void oracle_palette(uint32_t* out) { memcpy(out, atari_ntsc_palette, sizeof(atari_ntsc_palette)); }
/* state surgery for the reference-GIF pin (tests/test_breakout_gif_pin.py): overwrite the 128 RAM bytes */
void oracle_ale_set_ram(void* p, const uint8_t* in) { memcpy(((AleBox*)p)->ale.emu.ram, in, 128); }

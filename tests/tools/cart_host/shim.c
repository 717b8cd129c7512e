/* Test infrastructure: exposes the CPU oracle's bus / frame primitives (static in atari_oracle.c)
 * to the host build of the translated cartridge code (main.cpp).  Nothing here is product code. */
#include "../../../oracle/atari_oracle.c"

/* the bookkeeping of atari_frame() before its instruction loop (Stella TIA::startFrame) */
void host_frame_begin(Atari* a, uint8_t* fb) {
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
}
void host_cpu_step(Atari* a) { cpu_step(a); }
void host_wr(Atari* a, uint16_t addr, uint8_t v) { wr(a, addr, v); }
uint8_t host_tia_read(Atari* a, uint16_t addr, uint8_t noise) { a->bus = noise; return tia_read(a, (uint8_t)addr); }
uint8_t host_riot_read(Atari* a, uint16_t addr) { return riot_read(a, addr); }

/* ALU ops of the oracle on a scratch machine, for the exhaustive flag-arithmetic check */
uint16_t host_alu(int op, uint8_t A, uint8_t m, uint8_t P) {   /* returns (P << 8) | A */
  Atari t;
  memset(&t, 0, sizeof(t));
  t.A = A; t.P = P;
  if (op == 0) op_adc(&t, m); else if (op == 1) op_sbc(&t, m); else op_cmp(&t, A, m);
  return (uint16_t)((t.P << 8) | t.A);
}

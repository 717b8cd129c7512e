"""Dev tool (CPU only, test infrastructure): dynamic profile of a cartridge on the CPU oracle.

Builds an INSTRUMENTED COPY of oracle/atari_oracle.c under /tmp (the committed oracle is not
touched), plays the game with random actions and reports, per emulated frame:
  * 6507 instructions, TIA register writes, writes that really change the picture (the device's
    tia_store_is_nop classification), per register and per program counter;
  * the render segments a catch-up-on-real-change emulator produces: catch-ups, partial line
    segments, whole lines and whole lines that repeat the line above (render_seg);
  * a per-address execution histogram (<game>.hist: "pc runs real_writes" per frame) and an
    instruction trace (<game>.trace: uint16 pairs pc, real-write flag) for
    tools/cart_profile.py and for simulating dispatch-entry sets (gen_cart_native.Cart.entries).

    python tests/tools/oracle_profile.py [outdir=/tmp/prof] [agent steps: a long run writes <game>.cov, the addresses executed, and no trace]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PATCH_DECL = '''#include <stdio.h>
long g_pc_hist[65536], g_ind_tgt[65536]; int g_prev_pc = 0; long g_ret_n = 0; struct { unsigned short from, to; long n; } g_ret[4096]; FILE* g_trace; int g_trace_real; int g_prev_op = -1;
long g_nins, g_tiaw, g_real, g_real_reg[64], g_tot_reg[64], g_real_pc[65536], g_opc[256];
long g_seg_partial, g_seg_full, g_seg_repl, g_spans; static int32_t g_dev_last = -1000000000; int g_cur_pc;
static int is_nop_write(Atari* a, int reg, int v);
static void count_span(Atari* a);
'''

PATCH_TAIL = '''
/* the device's classification of a TIA write (atari_core.hpp tia_store_is_nop): 1 = rewrites the
 * value already held, 2 = never needs the picture, 0 = real change */
static int is_nop_write(Atari* a, int reg, int v) {
  switch (reg) {
    case 0x01: return a->vblank == v; case 0x04: return a->nusiz0 == v; case 0x05: return a->nusiz1 == v;
    case 0x06: return a->colup0 == (v & 0xfe); case 0x07: return a->colup1 == (v & 0xfe);
    case 0x08: return a->colupf == (v & 0xfe); case 0x09: return a->colubk == (v & 0xfe);
    case 0x0a: return a->ctrlpf == v; case 0x0b: return a->refp0 == v; case 0x0c: return a->refp1 == v;
    case 0x0d: return a->pf0 == v; case 0x0e: return a->pf1 == v; case 0x0f: return a->pf2 == v;
    case 0x1d: return a->enam0 == v; case 0x1e: return a->enam1 == v; case 0x1f: return a->enabl == v;
    case 0x25: return a->vdelp0 == v; case 0x26: return a->vdelp1 == v; case 0x27: return a->vdelbl == v;
    case 0x1b: return a->grp0 == v && a->dgrp1 == a->grp1;
    case 0x1c: return a->grp1 == v && a->dgrp0 == a->grp0 && a->denabl == a->enabl;
    case 0x00: case 0x02: case 0x03: case 0x15: case 0x16: case 0x17: case 0x18: case 0x19: case 0x1a:
    case 0x20: case 0x21: case 0x22: case 0x23: case 0x24: case 0x2b: return 2;
    default: return 0;
  }
}
static void count_span(Atari* a) {
  const int32_t c0 = frame_clock0(a);
  const int32_t start = c0 + CLOCKS_PER_LINE * ATARI_YSTART, stop = start + CLOCKS_PER_LINE * ATARI_H;
  int32_t clock = a->cyc * 3; if (clock > stop) clock = stop;
  int32_t last = g_dev_last; if (last < start || last > stop) last = start;
  if (last >= clock) { if (clock >= start) g_dev_last = clock; return; }
  g_spans++;
  int nfull = 0;
  while (last < clock) {
    const int hpos = (last - c0) % CLOCKS_PER_LINE;
    const int line_end = last + (CLOCKS_PER_LINE - hpos);
    const int seg_end = clock < line_end ? clock : line_end;
    const int h0 = hpos < HBLANK ? HBLANK : hpos, h1 = hpos + (seg_end - last);
    if (h1 > h0) { if (h0 == HBLANK && h1 == CLOCKS_PER_LINE) { g_seg_full++; if (nfull) g_seg_repl++; nfull++; } else g_seg_partial++; }
    last = seg_end;
  }
  g_dev_last = clock;
}
'''

MAIN = r'''
#include <stdio.h>
#include <stdlib.h>
#include "atari_oracle.h"
extern long g_ret_n; extern struct { unsigned short from, to; long n; } g_ret[4096];
extern long g_nins, g_tiaw, g_real, g_real_reg[64], g_tot_reg[64], g_real_pc[65536], g_opc[256], g_pc_hist[65536], g_ind_tgt[65536];
extern long g_seg_partial, g_seg_full, g_seg_repl, g_spans; extern FILE* g_trace;
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); static uint8_t rom[4096]; int n = (int)fread(rom, 1, 4096, f); fclose(f);
  const int game = atoi(argv[2]);
  static Ale e; static uint8_t fb[ATARI_FRAME_BYTES];
  ale_init(&e, rom, n, game); ale_reset(&e, fb);
  int acts[18]; const int na = ale_minimal_actions(game, acts);
  srand(1);
  for (int i = 0; i < 300; i++) { ale_act(&e, acts[rand() % na], fb); if (e.terminal) ale_reset(&e, fb); }
  g_nins = g_tiaw = g_real = g_seg_partial = g_seg_full = g_seg_repl = g_spans = 0;
  for (int i = 0; i < 64; i++) g_real_reg[i] = g_tot_reg[i] = 0;
  for (int i = 0; i < 65536; i++) g_real_pc[i] = g_pc_hist[i] = 0;
  for (int i = 0; i < 256; i++) g_opc[i] = 0;
  int F = argc > 5 ? atoi(argv[5]) : 400;   /* agent steps; long runs (coverage: cart_branch_profile.json "executed") write no trace */
  g_trace = F <= 1000 ? fopen(argv[4], "wb") : 0;
  for (int i = 0; i < F; i++) { const int a = acts[rand() % na]; for (int k = 0; k < 4; k++) ale_act(&e, a, fb); if (e.terminal) ale_reset(&e, fb); }
  F *= 4; if (g_trace) fclose(g_trace); g_trace = 0;
  printf("segments per frame: catch-ups %.1f partial %.1f whole lines %.1f (of which replicas %.1f)\n", (double)g_spans / F, (double)g_seg_partial / F, (double)g_seg_full / F, (double)g_seg_repl / F);
  printf("per frame: instructions %.0f  TIA writes %.0f  real picture changes %.0f\n", (double)g_nins / F, (double)g_tiaw / F, (double)g_real / F);
  for (int r = 0; r < 64; r++) if (g_tot_reg[r]) printf("  reg %02x: writes %.1f real %.1f\n", r, (double)g_tot_reg[r] / F, (double)g_real_reg[r] / F);
  printf("opcodes per frame: JSR %.1f RTS %.1f BRK %.1f RTI %.1f PLA %.1f PLP %.1f PHA %.1f PHP %.1f\n", (double)g_opc[0x20] / F, (double)g_opc[0x60] / F, (double)g_opc[0] / F, (double)g_opc[0x40] / F, (double)g_opc[0x68] / F, (double)g_opc[0x28] / F, (double)g_opc[0x48] / F, (double)g_opc[0x08] / F);
  FILE* o = fopen(argv[3], "w");
  for (int p = 0; p < 65536; p++) if (g_pc_hist[p] || g_real_pc[p]) fprintf(o, "%04x %.3f %.3f%s\n", p, (double)g_pc_hist[p] / F, (double)g_real_pc[p] / F, g_ind_tgt[p] ? " J" : "");
  for (long k = 0; k < g_ret_n; k++) fprintf(o, "R %04x %04x %ld\n", g_ret[k].from, g_ret[k].to, g_ret[k].n);
  fclose(o);
  return 0;
}
'''


def main(out, steps=None):
    os.makedirs(out, exist_ok=True)
    src = open(os.path.join(ROOT, 'oracle', 'atari_oracle.c')).read()

    def rep(old, new):
        nonlocal src
        assert src.count(old) == 1, old
        src = src.replace(old, new)

    rep('static void tia_write(Atari* a, uint8_t reg, uint8_t v) {\n  reg &= 0x3f;',
        PATCH_DECL + 'static void tia_write(Atari* a, uint8_t reg, uint8_t v) {\n  reg &= 0x3f; g_tiaw++; g_tot_reg[reg]++;\n'
        '  if (!is_nop_write(a, reg, v)) { g_trace_real = 1; count_span(a); g_real++; g_real_reg[reg]++; g_real_pc[g_cur_pc]++; }')
    rep('static void cpu_step(Atari* a) {\n  const uint8_t op = fetch(a);',
        'static void cpu_step(Atari* a) {\n'
        '  if (g_trace && g_nins > 0) { unsigned short r[2] = {(unsigned short)g_cur_pc, (unsigned short)g_trace_real}; fwrite(r, 2, 2, g_trace); }\n'
        '  g_trace_real = 0; g_cur_pc = a->PC; g_nins++; g_pc_hist[a->PC]++;\n  if (g_prev_op == 0x6c) g_ind_tgt[a->PC]++;  /* where a JMP () went */\n  if (g_prev_op == 0x60 || g_prev_op == 0x40) { long k = 0; for (; k < g_ret_n; k++) if (g_ret[k].from == g_prev_pc && g_ret[k].to == a->PC) break; if (k == g_ret_n && g_ret_n < 4096) { g_ret[k].from = (unsigned short)g_prev_pc; g_ret[k].to = a->PC; g_ret[k].n = 0; g_ret_n++; } if (k < 4096) g_ret[k].n++; }  /* RTS / RTI -> return site */\n  g_prev_pc = a->PC;\n  const uint8_t op = fetch(a); g_opc[op]++; g_prev_op = op;')
    src += PATCH_TAIL
    open(os.path.join(out, 'atari_prof.c'), 'w').write(src)
    open(os.path.join(out, 'main.c'), 'w').write(MAIN)
    subprocess.check_call(['cp', os.path.join(ROOT, 'oracle', 'atari_oracle.h'), out])
    subprocess.check_call(['gcc', '-O2', '-o', 'prof', 'atari_prof.c', 'main.c', '-lm'], cwd=out)
    for name, game in (('pong', 1), ('breakout', 2)):
        rom = os.path.join(ROOT, 'roms', name + '.bin')
        if os.path.exists(rom):
            print('==', name)
            subprocess.check_call(['./prof', rom, str(game), name + ('.hist' if steps is None else '.cov'), name + '.trace'] +
                                  ([] if steps is None else [str(steps)]), cwd=out)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/tmp/prof', int(sys.argv[2]) if len(sys.argv) > 2 else None)

// Test infrastructure (CPU only): the statically translated cartridge code
// (parl_amd/csrc/gen_cart_native.py -> cart_native.gen.hpp) compiled for the HOST and run against
// the pure CPU oracle, frame by frame.
//
// The generated blocks only touch the emulator through a small surface (registers, lazy flags,
// ram_rd / ram_wr / rom_byte, adc / sbc / cmp / bit, tia_store_is_nop, wsync, tia_read, riot_read,
// the `pend` hand-over, the instruction counter).  This file implements that surface on top of the
// oracle's Atari struct — flag arithmetic restated from atari_core.hpp — and mirrors Emu::frame:
//   native_run -> (decoded TIA store ? the oracle's bus write : one oracle instruction) -> repeat.
// A twin machine runs the oracle's own atari_frame().  After every frame the two machines must be
// identical (CPU registers, RAM, cycle counters, TIA / RIOT state, collision latches, the frame
// buffer).  That checks, without a GPU: every translated addressing mode / operation / cycle count /
// branch target, the dispatch-entry set, the hand-over protocol, and the premise of
// tia_store_is_nop (a write classified as a no-op must have no effect in the oracle).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

extern "C" {
#include "../../../oracle/atari_oracle.h"
void host_frame_begin(Atari* a, uint8_t* fb);
void host_cpu_step(Atari* a);
void host_wr(Atari* a, uint16_t addr, uint8_t v);
uint8_t host_tia_read(Atari* a, uint16_t addr, uint8_t noise);
uint8_t host_riot_read(Atari* a, uint16_t addr);
uint16_t host_alu(int op, uint8_t A, uint8_t m, uint8_t P);
}

#define DEVI inline
enum : int { JAM_NATIVE = 0x4000 };
enum : int { FN = 0x80, FV = 0x40, FU = 0x20, FB = 0x10, FD = 0x08, FI = 0x04, FZ = 0x02, FC = 0x01 };
constexpr int kMaxInstrPerFrame = 25000;
constexpr int kNativeInstrLimit = kMaxInstrPerFrame - 8192;
constexpr int kCyclesPerLine = 76;

struct Emu {
  Atari* a;
  int A, X, Y, S, PC;
  int P, nv, zv, cf;   // lazy flags as in atari_core.hpp
  int pend;
  int jam = 0;
  int& cyc;
  int& stop;   // VSYNC released: the frame is over (the generated code leaves right after that store)
  // RIOT timer state, read by the generated timer-wait loops to skip the iterations whose outcome is known
  uint8_t& timer;
  uint8_t& timer_shift;
  int32_t& timer_set_cyc;
  explicit Emu(Atari* m) : a(m), cyc(m->cyc), stop(m->stop), timer(m->timer), timer_shift(m->timer_shift), timer_set_cyc(m->timer_set_cyc) {}

  void load() { A = a->A; X = a->X; Y = a->Y; S = a->S; PC = a->PC; pset(a->P); pend = -1; }
  void store() { a->A = (uint8_t)A; a->X = (uint8_t)X; a->Y = (uint8_t)Y; a->S = (uint8_t)S; a->PC = (uint16_t)PC; a->P = (uint8_t)pfull(); }

  void set_nz(int v) { nv = v; zv = v; }
  int pfull() const { return P | (nv & FN) | ((zv & 0xff) ? 0 : FZ) | cf; }
  void pset(int v) { P = v & ~(FN | FZ | FC); nv = v; zv = (v & FZ) ^ FZ; cf = v & FC; }
  void bit(int m) { P = (P & ~FV) | (m & FV); nv = m; zv = A & m; }
  void adc(int m) {
    const int old = A, c = cf;
    int carry;
    if (P & FD) {
      const int sum = ((A >> 4) * 10 + (A & 15)) + ((m >> 4) * 10 + (m & 15)) + c;
      carry = sum > 99 ? FC : 0;
      const int tt = sum & 0xff;
      A = (((tt % 100) / 10) << 4) | (tt % 10);
    } else {
      const int sum = A + m + c;
      carry = (sum >> 8) & 1;
      A = sum & 0xff;
    }
    cf = carry;
    P = (P & ~FV) | (((~(old ^ m) & (old ^ A)) >> 1) & FV);
    set_nz(A);
  }
  void sbc(int m) {
    const int old = A, borrow = cf ^ 1;
    const int bin = old - m - borrow;
    if (P & FD) {
      int diff = ((A >> 4) * 10 + (A & 15)) - ((m >> 4) * 10 + (m & 15)) - borrow;
      diff += diff < 0 ? 100 : 0;
      A = ((((diff % 100) / 10) << 4) | (diff % 10)) & 0xff;
    } else {
      A = bin & 0xff;
    }
    cf = ((bin >> 8) & 1) ^ 1;
    P = (P & ~FV) | ((((old ^ m) & (old ^ A)) >> 1) & FV);
    set_nz(A);
  }
  // the translated code's forms with a statically dead V / C left alone: the device's text (Emu::adc_bin_f ... in
  // atari_core.hpp, compared by tests/test_cart_translator.py)
  template <bool KV, bool KC> void adc_bin_f(int m) {
    const int old = A;
    const int sum = A + m + cf;              // 0 .. 511
    if (KC) cf = (sum >> 8) & 1;
    A = sum & 0xff;
    if (KV) P = (P & ~FV) | (((~(old ^ m) & (old ^ A)) >> 1) & FV);
    set_nz(A);
  }
  template <bool KV, bool KC> void sbc_bin_f(int m) {
    const int old = A;
    const int bin = old - m - (cf ^ 1);      // -256 .. 255
    A = bin & 0xff;
    if (KC) cf = ((bin >> 8) & 1) ^ 1;
    if (KV) P = (P & ~FV) | ((((old ^ m) & (old ^ A)) >> 1) & FV);
    set_nz(A);
  }
  template <bool KV, bool KC> void adc_f(int m) {
    const int sp = P, sc = cf;
    adc(m);
    if (!KV) P = (P & ~FV) | (sp & FV);
    if (!KC) cf = sc;
  }
  template <bool KV, bool KC> void sbc_f(int m) {
    const int sp = P, sc = cf;
    sbc(m);
    if (!KV) P = (P & ~FV) | (sp & FV);
    if (!KC) cf = sc;
  }
  void adc_bin(int m) { adc_bin_f<true, true>(m); }
  void sbc_bin(int m) { sbc_bin_f<true, true>(m); }
  template <bool KC> void cmp_f(int r, int m) {
    const int d = r - m;                     // -255 .. 255
    if (KC) cf = ((d >> 8) & 1) ^ 1;         // r >= m
    set_nz(d & 0xff);
  }
  void cmp(int r, int m) { cmp_f<true>(r, m); }
  template <bool KV> void bit_f(int m) { if (KV) P = (P & ~FV) | (m & FV); nv = m; zv = A & m; }

  // the lanes of the device's TIA register file a translated loop keeps scalar shadows of (Emu::t / tset):
  // the write registers its stores are compared with, and the delayed-graphics latches (atari_defs.hpp TiaLane)
  uint8_t* tia_field(int r) const {
    switch (r) {
      case 0x01: return &a->vblank;  case 0x04: return &a->nusiz0;  case 0x05: return &a->nusiz1;
      case 0x06: return &a->colup0;  case 0x07: return &a->colup1;  case 0x08: return &a->colupf;
      case 0x09: return &a->colubk;  case 0x0a: return &a->ctrlpf;  case 0x0b: return &a->refp0;
      case 0x0c: return &a->refp1;   case 0x0d: return &a->pf0;     case 0x0e: return &a->pf1;
      case 0x0f: return &a->pf2;     case 0x1b: return &a->grp0;    case 0x1c: return &a->grp1;
      case 0x1d: return &a->enam0;   case 0x1e: return &a->enam1;   case 0x1f: return &a->enabl;
      case 0x25: return &a->vdelp0;  case 0x26: return &a->vdelp1;  case 0x27: return &a->vdelbl;
      case 0x35: return &a->dgrp0;   case 0x36: return &a->dgrp1;   case 0x37: return &a->denabl;
      default: fprintf(stderr, "host harness: TIA lane %02x has no shadow mapping\n", r); exit(1);
    }
  }
  int t(int r) const { return *tia_field(r); }
  void tset(int r, int v) { *tia_field(r) = (uint8_t)v; }

  int ram_rd(int i) const { return a->ram[i & 0x7f]; }
  void ram_wr(int i, int v) { a->ram[i & 0x7f] = (uint8_t)v; }
  int rom_byte(int ea) const { return a->rom[ea & a->rom_mask]; }
  int tia_read(int ea, int noise) { return host_tia_read(a, (uint16_t)ea, (uint8_t)noise); }
  int riot_read(int ea) { return host_riot_read(a, (uint16_t)ea); }
  // the device waits for its picture wave before a collision-latch read; the oracle's read catches its own picture
  // up to the cycle of the read (`c` = the cycle before it)
  int tia_read_zp(int ea, int noise, int c) { const int c0 = cyc; cyc = c + 1; const int v = tia_read(ea, noise); cyc = c0; return v; }
  void riot_write(int ea, int v) { cyc -= 1; host_wr(a, (uint16_t)ea, (uint8_t)v); }   // (host_wr counts the write cycle itself)
  void wsync(int cw) {
    const int into = (cw - a->cyc0) % kCyclesPerLine;
    cyc = cw + (into ? kCyclesPerLine - into : 0);
  }
  // atari_core.hpp tia_store_is_nop on the oracle's register fields
  bool tia_store_is_nop(int reg, int v) const {
    switch (reg) {
      case 0x01: return a->vblank == v;
      case 0x04: return a->nusiz0 == v;
      case 0x05: return a->nusiz1 == v;
      case 0x06: return a->colup0 == (v & 0xfe);
      case 0x07: return a->colup1 == (v & 0xfe);
      case 0x08: return a->colupf == (v & 0xfe);
      case 0x09: return a->colubk == (v & 0xfe);
      case 0x0a: return a->ctrlpf == v;
      case 0x0b: return a->refp0 == v;
      case 0x0c: return a->refp1 == v;
      case 0x0d: return a->pf0 == v;
      case 0x0e: return a->pf1 == v;
      case 0x0f: return a->pf2 == v;
      case 0x1d: return a->enam0 == v;
      case 0x1e: return a->enam1 == v;
      case 0x1f: return a->enabl == v;
      case 0x25: return a->vdelp0 == v;
      case 0x26: return a->vdelp1 == v;
      case 0x27: return a->vdelbl == v;
      case 0x1b: return a->grp0 == v && a->dgrp1 == a->grp1;
      case 0x1c: return a->grp1 == v && a->dgrp0 == a->grp0 && a->denabl == a->enabl;
      default: return false;
    }
  }
  int inpt_read(int reg, int noise) { return tia_read(reg, noise); }
  int tc(int r) const { return t(r); }   // the oracle writes eagerly: its register fields ARE the CPU-side file
  // The device records real changes in its write log and replays them later (atari_core.hpp); the oracle renders
  // eagerly, so here the write simply happens — at the cycle the device stamps the entry with (`cw` = the cycle
  // before the write cycle; the generated block adds the instruction's cycles to e.cyc afterwards).  Every fifth
  // request is declined so that the hand-over arm of the generated blocks stays exercised.
  int log_calls = 0;
  void write_at(int reg, int v, int cw) { const int c0 = cyc; cyc = cw; host_wr(a, (uint16_t)reg, (uint8_t)v); cyc = c0; }
  // the device publishes its local records to the picture wave at a trace's back edge (Emu::rq_flush); on the host a
  // record IS the oracle's register write, so nothing ever waits
  int wqn = 0;
  void rq_flush() {}
  bool tia_log(int reg, int v, int cw, bool quiet = false) {
    (void)quiet;  // D1 unchanged: the oracle's own write renders first and then changes no pixel
    if ((log_calls++ % 5) == 4) return false;
    write_at(reg, v, cw);
    return true;
  }
  // atari_core.hpp Emu::tia_store
  bool tia_store(int reg, int v, int cw, bool quiet_ok) {
    const bool plain = reg == 0x04 || reg == 0x05 || (reg >= 0x06 && reg <= 0x0f) || (reg >= 0x1d && reg <= 0x1f) || (reg >= 0x25 && reg <= 0x27);
    if (plain || reg == 0x1b || reg == 0x1c) {
      if (tia_store_is_nop(reg, v)) return true;
      const bool quiet = quiet_ok && reg >= 0x1d && reg <= 0x1f && !((t(reg) ^ v) & 0x02);
      return tia_log(reg, (reg >= 0x06 && reg <= 0x09) ? (v & 0xfe) : v, cw, quiet);
    }
    if (reg == 0x01 && tia_store_is_nop(reg, v)) return true;
    if (reg == 0x00 || reg == 0x01) return tia_log(reg, v, cw);   // the oracle's write does the CPU-side part too
    if ((reg >= 0x10 && reg <= 0x14) || (reg >= 0x20 && reg <= 0x24) || (reg >= 0x28 && reg <= 0x2c))
      return tia_log(reg, v, cw);   // strobes, HMxx / HMCLR: logged with the clock of the write
    if (reg == 0x03 || (reg >= 0x15 && reg <= 0x1a) || reg >= 0x2d) {
      write_at(reg, v, cw);   // audio / RSYNC / unmapped: no state on the device
      return true;
    }
    return false;   // WSYNC through a run-time address: handed over
  }
};

static long g_trace_iters[65536], g_defer_pc[65536];
#define PARLHIP_TRACE_ITER(head) (++g_trace_iters[head])
template <int GAME> struct NativeCart { static constexpr bool present = false; static constexpr uint32_t rom_crc32 = 0; };
template <int GAME> inline void native_run(Emu&, int&) {}
#include "cart_native.gen.hpp"

template <int GAME>
static void frame_translated(Atari* a, uint8_t* fb, long* native_instr, long* deferred) {
  host_frame_begin(a, fb);
  Emu e(a);
  int n = 0;
  while (n < kMaxInstrPerFrame && !a->stop) {
    e.load();
    const int n0 = n;
    native_run<GAME>(e, n);
    e.store();
    if (e.jam) { fprintf(stderr, "translated code raised JAM_NATIVE at PC %04x\n", e.PC); exit(1); }
    *native_instr += n - n0;
    if (e.pend == -2) continue;   // translated RTS / RTI: PC set, dispatch again (Emu::frame)
    if (e.pend < 0 && n >= kMaxInstrPerFrame) break;
    if (e.pend >= 0) host_wr(a, (uint16_t)(e.pend & 0xff), (uint8_t)(e.pend >> 8));   // Emu::step, pending path
    else { ++g_defer_pc[a->PC]; host_cpu_step(a); }
    ++*deferred;
    ++n;
  }
  a->fb = nullptr;
}

static int differs(const Atari& x, const Atari& y, const uint8_t* fx, const uint8_t* fy, int frame) {
  Atari p = x, q = y;
  p.bus = q.bus = 0;                 // the data-bus latch is modelled explicitly (noise operands)
  p.last_clock = q.last_clock = 0;   // how far the picture is rendered, not machine state
  p.fb = q.fb = nullptr;
  if (memcmp(&p, &q, sizeof(Atari)) != 0) {
    fprintf(stderr, "frame %d: machine state differs (PC %04x/%04x A %02x/%02x X %02x/%02x Y %02x/%02x S %02x/%02x P %02x/%02x cyc %d/%d cx %x/%x)\n",
            frame, x.PC, y.PC, x.A, y.A, x.X, y.X, x.Y, y.Y, x.S, y.S, x.P, y.P, x.cyc, y.cyc, x.cx, y.cx);
    for (int i = 0; i < 128; ++i) if (x.ram[i] != y.ram[i]) fprintf(stderr, "  ram[%02x] %02x/%02x\n", i, x.ram[i], y.ram[i]);
    return 1;
  }
  if (memcmp(fx, fy, ATARI_FRAME_BYTES) != 0) {
    for (int i = 0; i < ATARI_FRAME_BYTES; ++i) if (fx[i] != fy[i]) { fprintf(stderr, "frame %d: pixel (%d,%d) %02x/%02x\n", frame, i % ATARI_W, i / ATARI_W, fx[i], fy[i]); break; }
    return 1;
  }
  return 0;
}

template <int GAME>
static int run(const uint8_t* rom, int rom_size, int frames) {
  static Atari ref, tr;
  static uint8_t fref[ATARI_FRAME_BYTES], ftr[ATARI_FRAME_BYTES];
  atari_init(&ref, rom, (uint32_t)rom_size); atari_system_reset(&ref);
  atari_init(&tr, rom, (uint32_t)rom_size); atari_system_reset(&tr);
  memset(fref, 0, sizeof(fref)); memset(ftr, 0, sizeof(ftr));
  srand(3);
  long native_instr = 0, deferred = 0;
  int32_t paddle = 408823;
  for (int f = 0; f < frames; ++f) {
    // controller / console inputs as the ALE layer drives them: paddle resistance +/- 23000 per
    // frame inside [27450, 790196], fire button, RESET switch during the first frames
    const int r = rand();
    paddle += ((r & 3) == 1) ? 23000 : (((r & 3) == 2) ? -23000 : 0);
    paddle = paddle < 27450 ? 27450 : (paddle > 790196 ? 790196 : paddle);
    const uint8_t fire = (uint8_t)((r >> 4) & 1), reset = (uint8_t)(f >= 60 && f < 64);
    for (Atari* m : {&ref, &tr}) { m->paddle_res[0] = paddle; m->paddle_res[1] = 408823; m->paddle_fire[0] = fire; m->paddle_fire[1] = 0; m->sw_reset = reset; }
    atari_frame(&ref, fref);
    frame_translated<GAME>(&tr, ftr, &native_instr, &deferred);
    if (differs(ref, tr, fref, ftr, f)) return 1;
  }
  printf("ok: %d frames identical; %.0f translated instructions and %.1f deferrals per frame%s\n", frames,
         (double)native_instr / frames, (double)deferred / frames, ref.jam ? " (jam set)" : "");
  for (int h = 0; h < 65536; ++h)
    if (g_trace_iters[h]) printf("trace %04x: %.1f iterations per frame\n", h, (double)g_trace_iters[h] / frames);
  if (getenv("CART_HOST_DEFERRALS"))   // which instructions still go to the interpreter (dev aid)
    for (int h = 0; h < 65536; ++h)
      if (g_defer_pc[h] * 10 >= frames) printf("interpreted %04x: %.1f per frame (opcode %02x)\n", h, (double)g_defer_pc[h] / frames, rom[h & (rom_size - 1)]);
  memset(g_defer_pc, 0, sizeof(g_defer_pc)); memset(g_trace_iters, 0, sizeof(g_trace_iters));
  return 0;
}

// Exhaustive check of the lazy-flag ALU arithmetic restated above from atari_core.hpp (shifts and
// masks instead of compares, separate nv / zv / cf) against the oracle's ADC / SBC / CMP for every
// accumulator, operand, carry and decimal-mode combination (the games barely touch decimal mode).
static int alu_check() {
  static Atari dummy;
  long bad = 0;
  for (int op = 0; op < 3; ++op)
    for (int dec = 0; dec < 2; ++dec)
      for (int c = 0; c < 2; ++c)
        for (int A = 0; A < 256; ++A)
          for (int m = 0; m < 256; ++m) {
            const int P0 = FU | (dec ? FD : 0) | c | ((A * 7 + m) & (FN | FV | FZ));  // junk in the result flags
            Emu e(&dummy);
            e.A = A; e.pset(P0);
            if (op == 0) e.adc(m); else if (op == 1) e.sbc(m); else e.cmp(A, m);
            if (!dec && op < 2) {  // the forms translated loops use once binary mode is established at their head
              Emu b(&dummy);
              b.A = A; b.pset(P0);
              if (op == 0) b.adc_bin(m); else b.sbc_bin(m);
              if (b.pfull() != e.pfull() || b.A != e.A) { if (bad++ < 10) fprintf(stderr, "alu op %d (binary form) c %d A %02x m %02x differs\n", op, c, A, m); }
            }
            const uint16_t want = host_alu(op, (uint8_t)A, (uint8_t)m, (uint8_t)P0);
            const int got = (e.pfull() << 8) | (e.A & 0xff);
            if (got != want && bad++ < 10)
              fprintf(stderr, "alu op %d dec %d c %d A %02x m %02x: got P %02x A %02x want P %02x A %02x\n", op, dec, c, A, m,
                      got >> 8, got & 0xff, want >> 8, want & 0xff);
          }
  if (bad) return 1;
  printf("ok: ADC / SBC / CMP identical to the oracle for all 3 x 2 x 2 x 65536 cases\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 2 && !strcmp(argv[1], "--alu")) return alu_check();
  if (argc < 4) { fprintf(stderr, "usage: cart_host <rom.bin> <1=pong|2=breakout> <frames>\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  static uint8_t rom[4096];
  const int n = (int)fread(rom, 1, sizeof(rom), f);
  fclose(f);
  const int game = atoi(argv[2]), frames = atoi(argv[3]);
  return game == 2 ? run<GAME_BREAKOUT>(rom, n, frames) : run<GAME_PONG>(rom, n, frames);
}

#!/usr/bin/env python3
"""gen_cart_native.py — static translation of an Atari 2600 cartridge into straight-line gfx950
code for the one-env-per-wavefront emulator (atari_core.hpp).

Why: the 6507 interpreter is wave-uniform scalar code whose time goes into instruction DISPATCH,
not execution.  Measured on MI355X with one wavefront per SIMD (tools/issue_microbench.hip):
a dependent SALU op costs 4.5 clk, a NOT-taken conditional branch 12 clk, a taken one 30 clk, an
LDS round trip 90 clk; the compiler lowers every `switch` to a compare/branch tree (AMDGPU has no
jump tables), so decoding mode / kind / operation costs ~15 branches = ~400 of the ~680 clk an
interpreted 6507 instruction takes.  The cartridge is read-only and known when the library is
built, so each reachable instruction is emitted here as its own block of C++ with mode, operation,
operand bytes, cycle count, page-crossing penalty and branch target folded to constants;
control flow between instructions is the program's own (fall-through / goto).

Exactness: a block implements precisely what Emu::step_fast does for that instruction (same
helpers, same cycle accounting).  Anything else — real TIA register changes (the picture must be
caught up first), collision-latch reads, JSR/RTS/BRK/RTI, stack-in-TIA tricks with a changing
value, undocumented opcodes — sets PC and returns to the caller, which executes that ONE
instruction with the interpreter and re-enters through the dispatch switch.  The result is
bit-identical to pure interpretation (tests/test_gpu_env.py compare against the CPU oracle).

Output: cart_native.gen.hpp (not committed: derived from user-supplied ROM data).
Usage: gen_cart_native.py <out.hpp> [name=path.bin ...]   (name in {pong, breakout})
"""
import os
import re
import sys
import zlib

# PARLHIP_CART_MARKERS=1: emit an asm comment at every block start (profiling builds only:
# tools/cart_profile.py maps the compiled ISA back to 6507 addresses)
MARKERS = bool(os.environ.get('PARLHIP_CART_MARKERS'))

# Real changes of the plain TIA registers and GRP0 / GRP1 are RECORDED in the emulator's write log by the translated
# code (Emu::tia_store -> tia_log: effective colour clock, register, value) instead of being handed to the
# interpreter one by one; the interpreter replays the log before anything that needs the picture (atari_core.hpp,
# "The WRITE LOG").  This supersedes round 2's six-entry playfield queue of the Breakout build.
# Cartridges that read collision latches through zp,X / zp,Y (Pong: `LDA CXM0P,X`, ~11 times a frame): for
# them that arm is an ordinary hand-over whose successor is a dispatch entry; elsewhere it is `/*rare*/`.
ZPX_LATCH_GAMES = ('pong', )
# games whose ENAMx / ENABL bytes with an unchanged D1 (the one wired bit) are logged WITHOUT a catch-up request:
# Pong writes the processor status into them with PHP.  Breakout never does and only pays for the extra test at
# every store site (PMC: 158.6 k -> 161.3 k instructions per frame with it on).
QUIET_STORE_GAMES = ('pong', )
# games whose innermost 6507 loops get ONE way in (Cart.find_loops).  MEASURED AND SWITCHED OFF: it does
# what it says — the guard-flag chains disappear from the hot path (Pong: 113.0 k -> 106.4 k instructions
# per frame, SALU 79 k -> 71 k) — but the branch count does not move (14.1 k), SGPR spills go 580 -> 920,
# every re-entry after a hand-over now takes dispatch -> loop head -> second switch, and the kernel is
# bound by fetch latency after taken branches, not by instruction count: SQ_WAVE_CYCLES -1.9 % in the
# reset-phase micro-benchmark, but the whole pipeline in steady play 2.66 -> 2.59 M frames/s.
# The cartridge's own conditional branches carry the taken / not-taken counts of a CPU oracle run
# (cart_branch_profile.json, by cartridge CRC-32: tests/tools/oracle_profile.py writes the trace,
# tools/make_branch_profile.py counts it) as __builtin_expect_with_probability, so that LLVM lays the likely
# successor out as the fall-through.  Measured on MI355X, E=1024: Pong 1.14 -> 1.12 ms per agent step,
# Breakout 1.86 -> 1.84 (PARLHIP_BRANCH_PGO=0 switches it off).  Small because the 6507's own Bcc are only
# 783 of the 14,100 ISA branches a Pong frame executes (Breakout: 928 of 19,300): the rest are address-class,
# no-op-store, flag and renderer conditionals inside the blocks.
BRANCH_PGO = bool(int(os.environ.get('PARLHIP_BRANCH_PGO', '1')))
# Hot loops as TRACES (round 4; PARLHIP_TRACE_LOOPS=0 switches it off).  A single-stream innermost 6507 loop
# (Cart.find_loops) whose back edge the profile says is hot is emitted a second time, specialised, in front
# of its generic blocks:
#   * ONE way in (the loop head, after a precondition test), any number of ways out, forward edges only inside:
#     reducible control flow.  The generic copy of the same instructions loses its back edge (it returns to the
#     dispatcher, which re-enters the trace), so it is acyclic: LLVM's FixIrreducible / UnifyLoopExits guard-flag
#     chains (~15 scalar moves + a branch on every block with a hand-over path, a dozen blocks per iteration of
#     Pong's scanline loop) disappear from both.
#   * facts that hold for the whole loop are established ONCE at the head instead of per instruction: the stack
#     pointer (S == its dataflow value: every PHP / PLA inside has a constant address), binary mode (no decimal
#     arm in ADC / SBC), zero-page bytes the loop only reads (held in scalar registers: a v_readlane round trip
#     is ~33 clocks against 4.5 for a scalar op, and Pong's loop reads 18 such bytes per iteration), pointers of
#     (zp),Y loads that stay inside the cartridge (no address-class tree), and scalar shadows of the TIA
#     registers its stores are compared with (the no-op test of a GRP1 store reads five lanes of the register
#     file).  A real TIA change still leaves through `pend` exactly like the generic block — after the
#     interpreter's TIA stages the rest of that iteration runs in the generic blocks, the next one here.
# An oracle trace of Pong shows 79 % of the 91 iterations per frame complete without any hand-over.
# Timer wait loops (`L: LDA INTIM; BNE L` — both cartridges burn the vertical blank this way: 116 iterations per Pong
# frame, 399 per Breakout frame = 17 % of its instructions) run in place, and the iterations whose outcome is known
# from the RIOT's state — the timer value stays >= 2 — are skipped in one step (PARLHIP_WAIT_LOOPS=0: off).
WAIT_LOOPS = bool(int(os.environ.get('PARLHIP_WAIT_LOOPS', '1')))
TRACE_LOOPS = bool(int(os.environ.get('PARLHIP_TRACE_LOOPS', '1')))
# translate only what a long oracle run executed + its static successors (Cart.discover); PARLHIP_PRUNE=0: every address
PRUNE = bool(int(os.environ.get('PARLHIP_PRUNE', '1')))
# Dead flags (round 6; PARLHIP_DEAD_FLAGS=0: off).  V and C cost the scalar unit six and two instructions per ADC / SBC
# (two per compare / shift) and almost nothing ever reads them: a backward liveness pass over the cartridge's static control
# flow (Cart.flag_liveness: readers BVC / BVS / PHP / BRK resp. ADC / SBC / ROL / ROR / BCC / BCS / PHP / BRK; RTS, RTI,
# JMP (), JAM and edges out of the translated set count as readers) says where a flag is dead, and the defining
# instruction is then emitted without it.  Inside a trace the walk is over the trace's own instructions, where a PHP
# onto ENAM0 / ENAM1 / ENABL (stack pointer a fact) reads only Z (Emu::tia_wired), and every possible exit falls back on
# the static answer for the generic code behind it.  The stale bit is unobservable by construction; the state blob's P
# differs from the interpreter's in those bits (tests/test_gpu_env.py masks V and C in that one comparison).
DEAD_FLAGS = bool(int(os.environ.get('PARLHIP_DEAD_FLAGS', '1')))
# RTS / RTI jumping straight to the return sites the profile run saw (cart_branch_profile.json "returns") instead of
# leaving for a re-dispatch.  MEASURED AND OFF (PARLHIP_RET_PREDICT=1: on): with every site's usual targets the return
# edges close cycles through the subroutines and the kernel no longer compiles in 15 minutes; with one target per site
# (share >= 0.6: PARLHIP_RET_PREDICT_MAX=1 PARLHIP_RET_PREDICT_MIN=0.6) it compiles and is bit-exact, but Pong reads
# 7.56 M frames/s against 7.62 M without it, Breakout 4.71 = 4.68 M: a re-dispatch costs less than what the extra
# edges cost the code around them.
# records are published to wave B at a trace's back edge once this many are waiting (0: only when the translated code
# returns to the frame loop, i.e. for Pong's display kernel when the 64-entry local log is full)
TRACE_FLUSH = int(os.environ.get('PARLHIP_TRACE_FLUSH', '8'))
RET_PREDICT = bool(int(os.environ.get('PARLHIP_RET_PREDICT', '0')))
RET_PREDICT_MAX = int(os.environ.get('PARLHIP_RET_PREDICT_MAX', '4'))
RET_PREDICT_MIN = float(os.environ.get('PARLHIP_RET_PREDICT_MIN', '0.02'))
F_READS = {'V': {'BVC', 'BVS', 'PHP', 'BRK'},
           'C': {'ADC', 'SBC', 'ROL', 'ROR', 'ROL_A', 'ROR_A', 'BCC', 'BCS', 'PHP', 'BRK'}}
F_KILLS = {'V': {'ADC', 'SBC', 'BIT', 'CLV', 'PLP'},
           'C': {'ADC', 'SBC', 'CMP', 'CPX', 'CPY', 'ASL', 'LSR', 'ROL', 'ROR', 'ASL_A', 'LSR_A', 'ROL_A', 'ROR_A', 'SEC', 'CLC', 'PLP'}}
# Measured on MI355X, E=1024, after reset (profiles/r04_trace_loops.log): Pong 1.12 -> 1.02 ms per agent step (PMC per
# frame: 111.8 k -> 100.5 k instructions, 13.9 k -> 12.7 k branches; the translated code's share of a frame 400 k -> 295 k
# clocks).  Breakout, whose loops index RAM with X (`LDA zp,X`, `DEC zp,X`: nothing to hoist): 1.81 -> 1.98 ms while its
# blocks still carried round 2's playfield queue inline (669 -> 1313 SGPR spills); with the write log instead
# 1.79 -> 1.61 ms.  Per game:
TRACE_GAMES = tuple(x for x in os.environ.get('PARLHIP_TRACE_GAMES', 'pong,breakout').split(',') if x)
TRACE_MIN_TAKEN = 8000   # back-edge "taken" count in cart_branch_profile.json (1600 profiled frames: >= 5 per frame)
LOOP_REENTRY_GAMES = tuple(x for x in os.environ.get('PARLHIP_LOOP_REENTRY', '').split(',') if x)  # default: none

# ---- mirrors atari_defs.hpp (decode_opcode) ------------------------------------------------------
M_IMP, M_IMM, M_ZP, M_ZPX, M_ZPY, M_ABS, M_ABX, M_ABY, M_IZX, M_IZY, M_REL, M_PUSH, M_PULL = range(13)
K_NONE, K_READ, K_WRITE, K_RMW = range(4)
OPS = ('JAM NOP ORA AND EOR ADC SBC CMP CPX CPY LDA LDX LDY STA STX STY BIT ASL LSR ROL ROR INC DEC ASL_A LSR_A '
       'ROL_A ROR_A INX INY DEX DEY TAX TAY TXA TYA TSX TXS CLC SEC CLI SEI CLV CLD SED PHA PHP PLA PLP BPL BMI BVC '
       'BVS BCC BCS BNE BEQ JMP JMPI JSR RTS RTI BRK').split()
O = {n: i for i, n in enumerate(OPS)}


def decode_opcode(op):
    cc, bbb, aaa = op & 3, (op >> 2) & 7, op >> 5
    m01 = [M_IZX, M_ZP, M_IMM, M_ABS, M_IZY, M_ZPX, M_ABY, M_ABX]
    if cc == 1:
        o01 = ['ORA', 'AND', 'EOR', 'ADC', 'STA', 'LDA', 'CMP', 'SBC']
        if op == 0x89:
            return (M_IMP, K_NONE, 'JAM')
        return (m01[bbb], K_WRITE if aaa == 4 else K_READ, o01[aaa])
    single = {
        0x00: (M_IMP, K_NONE, 'BRK'), 0x20: (M_IMP, K_NONE, 'JSR'), 0x40: (M_IMP, K_NONE, 'RTI'),
        0x60: (M_IMP, K_NONE, 'RTS'), 0x4c: (M_IMP, K_NONE, 'JMP'), 0x6c: (M_IMP, K_NONE, 'JMPI'),
        0x08: (M_PUSH, K_WRITE, 'PHP'), 0x28: (M_PULL, K_READ, 'PLP'), 0x48: (M_PUSH, K_WRITE, 'PHA'),
        0x68: (M_PULL, K_READ, 'PLA'), 0x88: (M_IMP, K_NONE, 'DEY'), 0xa8: (M_IMP, K_NONE, 'TAY'),
        0xc8: (M_IMP, K_NONE, 'INY'), 0xe8: (M_IMP, K_NONE, 'INX'), 0x18: (M_IMP, K_NONE, 'CLC'),
        0x38: (M_IMP, K_NONE, 'SEC'), 0x58: (M_IMP, K_NONE, 'CLI'), 0x78: (M_IMP, K_NONE, 'SEI'),
        0x98: (M_IMP, K_NONE, 'TYA'), 0xb8: (M_IMP, K_NONE, 'CLV'), 0xd8: (M_IMP, K_NONE, 'CLD'),
        0xf8: (M_IMP, K_NONE, 'SED'), 0x8a: (M_IMP, K_NONE, 'TXA'), 0x9a: (M_IMP, K_NONE, 'TXS'),
        0xaa: (M_IMP, K_NONE, 'TAX'), 0xba: (M_IMP, K_NONE, 'TSX'), 0xca: (M_IMP, K_NONE, 'DEX'),
        0xea: (M_IMP, K_NONE, 'NOP'), 0x0a: (M_IMP, K_NONE, 'ASL_A'), 0x2a: (M_IMP, K_NONE, 'ROL_A'),
        0x4a: (M_IMP, K_NONE, 'LSR_A'), 0x6a: (M_IMP, K_NONE, 'ROR_A'), 0x10: (M_REL, K_NONE, 'BPL'),
        0x30: (M_REL, K_NONE, 'BMI'), 0x50: (M_REL, K_NONE, 'BVC'), 0x70: (M_REL, K_NONE, 'BVS'),
        0x90: (M_REL, K_NONE, 'BCC'), 0xb0: (M_REL, K_NONE, 'BCS'), 0xd0: (M_REL, K_NONE, 'BNE'),
        0xf0: (M_REL, K_NONE, 'BEQ'), 0x24: (M_ZP, K_READ, 'BIT'), 0x2c: (M_ABS, K_READ, 'BIT'),
        0x84: (M_ZP, K_WRITE, 'STY'), 0x94: (M_ZPX, K_WRITE, 'STY'), 0x8c: (M_ABS, K_WRITE, 'STY'),
        0xa0: (M_IMM, K_READ, 'LDY'), 0xa4: (M_ZP, K_READ, 'LDY'), 0xb4: (M_ZPX, K_READ, 'LDY'),
        0xac: (M_ABS, K_READ, 'LDY'), 0xbc: (M_ABX, K_READ, 'LDY'), 0xc0: (M_IMM, K_READ, 'CPY'),
        0xc4: (M_ZP, K_READ, 'CPY'), 0xcc: (M_ABS, K_READ, 'CPY'), 0xe0: (M_IMM, K_READ, 'CPX'),
        0xe4: (M_ZP, K_READ, 'CPX'), 0xec: (M_ABS, K_READ, 'CPX'), 0x86: (M_ZP, K_WRITE, 'STX'),
        0x96: (M_ZPY, K_WRITE, 'STX'), 0x8e: (M_ABS, K_WRITE, 'STX'), 0xa2: (M_IMM, K_READ, 'LDX'),
        0xa6: (M_ZP, K_READ, 'LDX'), 0xb6: (M_ZPY, K_READ, 'LDX'), 0xae: (M_ABS, K_READ, 'LDX'),
        0xbe: (M_ABY, K_READ, 'LDX'),
    }
    if op in single:
        return single[op]
    if cc == 2 and bbb in (1, 3, 5, 7) and aaa not in (4, 5):
        o10 = ['ASL', 'ROL', 'LSR', 'ROR', None, None, 'DEC', 'INC']
        m10 = [0, M_ZP, 0, M_ABS, 0, M_ZPX, 0, M_ABX]
        return (m10[bbb], K_RMW, o10[aaa])
    return (M_IMP, K_NONE, 'JAM')


def length(mode):
    if mode in (M_IMP, M_PUSH, M_PULL):
        return 1
    if mode in (M_ABS, M_ABX, M_ABY):
        return 3
    return 2


FLAGS = dict(FN=0x80, FV=0x40, FU=0x20, FB=0x10, FD=0x08, FI=0x04, FZ=0x02, FC=0x01)


PLAIN_REGS = {0x01, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b, 0x0c, 0x0d, 0x0e, 0x0f, 0x1d, 0x1e, 0x1f, 0x25,
              0x26, 0x27}  # Emu::kPlainRegs
T_DGRP0, T_DGRP1, T_DENABL = 0x35, 0x36, 0x37   # atari_defs.hpp TiaLane (derived lanes of the register file)


class Trace(object):
    """What holds for every iteration of a hot single-stream loop, found by a forward pass over its
    instruction stream (forward branches only, plus the back edge(s) to the head)."""

    def __init__(self, cart, head, stream):
        self.cart = cart
        self.head, self.stream, self.sset = head, stream, set(stream)
        code = cart.code
        ops = [code[a][2] for a in stream]
        # ---- stack pointer / X as constants (same transfer function as Cart.stack_hints, now as FACTS under
        # the precondition S == S_head): state before each instruction, met over fall-through and branch edges
        UNK = None
        s_head = cart.s_hint.get(head)
        st_in = {head: (s_head, UNK)}
        self.S = {}
        ok = s_head is not None

        def meet(a, b):
            return tuple(x if x == y else UNK for x, y in zip(a, b))

        back = []
        for i, a in enumerate(stream):
            if a not in st_in:
                st_in[a] = (UNK, UNK)  # unreachable by fall-through (cannot happen in a linear stream)
            S, X = st_in[a]
            self.S[a] = S
            mode, kind, op, b1, b2 = code[a]
            if op == 'LDX':
                X = b1 if mode == M_IMM else UNK
            elif op == 'TAX':
                X = UNK
            elif op == 'TSX':
                X = S
            elif op in ('INX', 'DEX'):
                X = UNK if X is UNK else (X + (1 if op == 'INX' else -1)) & 0xff
            elif op == 'TXS':
                S = X
            elif op in ('PHA', 'PHP'):
                S = UNK if S is UNK else (S - 1) & 0xff
            elif op in ('PLA', 'PLP'):
                S = UNK if S is UNK else (S + 1) & 0xff
            elif op == 'JSR':
                S = UNK  # leaves the trace
            out = (S, X)
            nxt = (a + length(mode)) & 0xffff
            if mode == M_REL:
                t = (a + 2 + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff
                if t == head:
                    back.append(out)
                elif t in self.sset:
                    st_in[t] = meet(st_in[t], out) if t in st_in else out
            if i + 1 < len(stream):
                n2 = stream[i + 1]
                st_in[n2] = meet(st_in[n2], out) if n2 in st_in else out
        if not back or any(o[0] != s_head for o in back):
            ok = False
        self.use_S = ok
        if not ok:
            self.S = {a: None for a in stream}
        # ---- binary mode
        self.has_alu = any(o in ('ADC', 'SBC') for o in ops)
        self.d_clear = self.has_alu and not any(o in ('SED', 'PLP') for o in ops)
        # ---- RAM bytes the loop only reads at static addresses
        reads, written, dyn_store = set(), set(), False
        for a in stream:
            mode, kind, op, b1, b2 = code[a]
            if kind == K_READ:
                if mode == M_ZP and b1 >= 0x80:
                    reads.add(b1 & 0x7f)
                elif mode == M_ABS and not ((b1 | (b2 << 8)) & 0x1000) and ((b1 | (b2 << 8)) & 0x280) == 0x80:
                    reads.add(b1 & 0x7f)
                elif mode == M_IZY and 0x80 <= b1 < 0xff:
                    reads |= {b1 & 0x7f, (b1 + 1) & 0x7f}
            elif kind in (K_WRITE, K_RMW):
                if mode == M_ZP:
                    if b1 >= 0x80:
                        written.add(b1 & 0x7f)
                elif mode == M_PUSH:
                    S = self.S[a]
                    if S is None:
                        dyn_store = True
                    elif S & 0x80:
                        written.add(S & 0x7f)
                else:
                    dyn_store = True   # zp,X / zp,Y / absolute / indirect stores: the address is a run-time value
            if op in ('JSR', 'BRK'):
                pass  # pushes, then leaves the trace: nothing after it runs here
        self.hoist = set() if dyn_store else (reads - written)
        # ---- (zp),Y loads whose pointer bytes are hoisted: one range test at the head instead of a tree per load
        self.rom_ptrs = sorted({code[a][3] for a in stream if code[a][0] == M_IZY and code[a][1] == K_READ
                                and 0x80 <= code[a][3] < 0xff and {code[a][3] & 0x7f, (code[a][3] + 1) & 0x7f} <= self.hoist})
        # ---- scalar shadows of the TIA registers the loop's stores are compared with
        self.shadows = set()
        quiet = cart.name in QUIET_STORE_GAMES
        for a in stream:
            mode, kind, op, b1, b2 = code[a]
            if kind != K_WRITE:
                continue
            reg = None
            if mode == M_ZP and b1 < 0x80:
                reg = b1 & 0x3f
            elif mode == M_PUSH and self.S[a] is not None and not (self.S[a] & 0x80):
                reg = self.S[a] & 0x3f
            if reg is None:
                continue
            if reg in PLAIN_REGS and not (0x0d <= reg <= 0x0f):
                self.shadows.add(reg)
            elif reg == 0x1b:
                self.shadows |= {0x1b, 0x1c, T_DGRP1}
            elif reg == 0x1c:
                self.shadows |= {0x1b, 0x1c, 0x1f, T_DGRP0, T_DENABL}
        self.quiet = quiet

    def precondition(self):
        c = []
        if self.use_S:
            c.append('e.S == 0x%02x' % self.S[self.head])
        if self.d_clear:
            c.append('!(e.P & FD)')
        for b1 in self.rom_ptrs:
            c.append('(hp_%02x & 0x1000) && ((hp_%02x + 0xff) & 0x1000)' % (b1, b1))
        return ' && '.join('(%s)' % x for x in c) if c else '1'

    def prologue(self):
        L = ['const int h_%02x = e.ram_rd(0x%02x);' % (x, x) for x in sorted(self.hoist)]
        L += ['const int hp_%02x = h_%02x | (h_%02x << 8);' % (b1, b1 & 0x7f, (b1 + 1) & 0x7f) for b1 in self.rom_ptrs]
        L += ['int ts_%02x = e.tc(0x%02x);' % (r, r) for r in sorted(self.shadows)]
        L += ['int rc_%04x_ea = -1, rc_%04x_m = 0;' % (a, a) for a in self.stream
              if self.cart.code[a][0] == M_IZY and self.cart.code[a][1] == K_READ and self.cart.code[a][3] in self.rom_ptrs]
        return L


class Cart(object):
    def __init__(self, name, rom):
        assert len(rom) in (2048, 4096)
        self.name, self.rom, self.mask = name, rom, len(rom) - 1
        self.quiet_ok = 'true' if name in QUIET_STORE_GAMES else 'false'
        self.branch_prob = {}
        if BRANCH_PGO:
            import json
            prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cart_branch_profile.json')
            ent = json.load(open(prof)).get('%08x' % (zlib.crc32(rom) & 0xffffffff)) if os.path.exists(prof) else None
            if ent:
                for a, (nt, tk) in ent['branches'].items():
                    if nt + tk >= 16:
                        self.branch_prob[int(a, 16)] = tk / float(nt + tk)
        self.code = {}  # 16-bit address -> (mode, kind, op, b1, b2)
        self.executed = set()
        self.indirect_targets = set()   # where the profile run's JMP () went: dispatch entries (Cart.entries)
        self.returns = {}               # RTS / RTI address -> [(count, return site)]
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cart_branch_profile.json')):
            import json
            ent = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cart_branch_profile.json'))).get(
                '%08x' % (zlib.crc32(rom) & 0xffffffff))
            if ent and ent.get('executed'):
                self.executed = {int(x, 16) for x in ent['executed'].split()}
            self.indirect_targets = {int(x, 16) for x in ent.get('indirect_targets', '').split()} if ent else set()
            for item in (ent.get('returns', '').split() if ent else []):
                ft, cnt = item.split(':')
                f, t = ft.split('>')
                self.returns.setdefault(int(f, 16), []).append((int(cnt), int(t, 16)))
        self.discover()
        self.s_hint = self.stack_hints()
        self.cur = None       # block being emitted (goto() needs the source of an edge)
        self.tc = None        # trace being emitted (None: generic blocks)
        self.branch_taken = {}
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cart_branch_profile.json')):
            import json
            ent = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cart_branch_profile.json'))).get(
                '%08x' % (zlib.crc32(rom) & 0xffffffff))
            if ent:
                self.branch_taken = {int(a, 16): tk for a, (nt, tk) in ent['branches'].items()}
        all_loops = self.find_loops()
        self.loops = all_loops if name in LOOP_REENTRY_GAMES else []
        self.wait_loops = self.find_wait_loops() if WAIT_LOOPS else {}
        self.traces = {}      # loop head -> Trace
        self.trace_of = {}    # instruction start inside a traced loop -> its head
        if TRACE_LOOPS and name in TRACE_GAMES and name not in LOOP_REENTRY_GAMES:
            for h, stream in all_loops:
                if set(stream) & set(self.wait_loops):
                    continue  # a timer wait loop: run in place with its iterations skipped (emit_wait_loop)
                if max(self.branch_taken.get(x, 0) for x in stream) >= TRACE_MIN_TAKEN:
                    self.traces[h] = Trace(self, h, stream)
                    for x in stream:
                        self.trace_of[x] = h
        self.flag_live = self.flag_liveness() if DEAD_FLAGS else None   # flag -> addresses where it is live BEFORE the instruction
        self._can_exit = {}
        self._in_analysis = False
        self.loop_of = {}     # instruction start inside a re-entry loop -> index of the loop
        for i, (h, stream) in enumerate(self.loops):
            for x in stream:
                self.loop_of[x] = i

    def byte(self, a):
        return self.rom[a & self.mask]

    def word(self, a):
        return self.byte(a) | (self.byte(a + 1) << 8)

    def discover(self):
        """The instruction starts we emit.  A block is a faithful translation of whatever bytes sit
        at its address, so over-approximating the set of entry points is harmless (a block that is
        never reached is never fetched) while a missing one only means that instruction gets
        interpreted.  So: EVERY address of the cartridge mirror the reset vector points into
        (computed jumps, RTS tricks and BRK/RTI calls need no analysis), plus recursive descent
        from the vectors through branches / JMP / JSR for code reached through another mirror."""
        reset = self.word(0xfffc)
        lo = reset & ~self.mask & 0xffff
        work = [reset, self.word(0xfffe)] + list(range(lo, lo + len(self.rom)))
        if PRUNE and self.executed:
            # Round 6: with the coverage of a long oracle run at hand (cart_branch_profile.json "executed"), only what that
            # run executed plus everything reachable from it by the program's static edges is translated — data bytes and
            # misaligned decodes are not (Pong: 2,116 -> ~900 blocks).  An address outside the set is still safe: a jump to
            # it is a hand-over and the interpreter steps until it reaches a translated entry.  The kernel loses the dead
            # blocks LLVM cannot prove dead (each is a dispatch entry's fall-through) and with them code size, long
            # branches and compile time.
            work = [reset, self.word(0xfffe)] + sorted(self.executed)
        while work:
            a = work.pop() & 0xffff
            while True:
                if a in self.code or not (a & 0x1000):
                    break
                opc = self.byte(a)
                mode, kind, op = decode_opcode(opc)
                b1, b2 = self.byte(a + 1), self.byte(a + 2)
                self.code[a] = (mode, kind, op, b1, b2)
                if op in ('JMP', 'JSR'):
                    work.append(b1 | (b2 << 8))
                if mode == M_REL:
                    work.append((a + 2 + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff)
                if op in ('JAM', 'BRK', 'RTS', 'RTI', 'JMPI', 'JMP'):
                    break
                a = (a + length(mode)) & 0xffff

    def find_wait_loops(self):
        """{address of `LDA abs INTIM`: (INTIM address, cycles of one iteration)} for `LDA INTIM; BNE back`"""
        out = {}
        for a, (mode, kind, op, b1, b2) in self.code.items():
            ea = b1 | (b2 << 8)
            if op != 'LDA' or mode != M_ABS or (ea & 0x1000) or (ea & 0x285) != 0x284:
                continue
            nx = (a + 3) & 0xffff
            if nx not in self.code or self.code[nx][2] != 'BNE':
                continue
            c1 = self.code[nx][3]
            npc = (nx + 2) & 0xffff
            if (npc + (c1 - 256 if c1 & 0x80 else c1)) & 0xffff != a:
                continue
            out[a] = (ea, 4 + (4 if ((a ^ npc) & 0xff00) else 3))
        return out

    def emit_wait_loop(self, a):
        ea, it = self.wait_loops[a]
        nxt = (a + 5) & 0xffff
        return [
            '// timer wait loop (LDA INTIM; BNE back): in place; iterations during which the timer value stays >= 2 are skipped',
            'for (;;) {',
            '  if (__builtin_expect(n > kNativeInstrLimit, 0)) { --n; e.PC = 0x%04x; return; }' % a,
            '  {',
            '    const int delta = (e.cyc + 3) - e.timer_set_cyc;  // Emu::riot_read at this iteration\'s read cycle',
            '    const int bound = (e.timer - 2) << e.timer_shift;   // the value read is >= 2 while delta < bound',
            '    if (bound > delta) {',
            '      int k = (bound - delta + %d) / %d;' % (it - 1, it),
            '      const int room = (kNativeInstrLimit - n) >> 1;',
            '      k = k > room ? room : k;',
            '      n += 2 * k; e.cyc += %d * k;' % it,
            '    }',
            '  }',
            '  e.cyc += 4;',
            '  const int m = e.riot_read(0x%04x);' % ea,
            '  e.A = m; e.set_nz(e.A);',
            '  ++n;  // the BNE',
            '  if (m == 0) { e.cyc += 2; break; }',
            '  e.cyc += %d;' % (it - 4),
            '  ++n;  // the next LDA',
            '}',
            self.goto(nxt),
        ]

    def find_loops(self):
        """Innermost 6507 loops that get a single way in.  The dispatch switch of native_run enters the
        code at every instruction that follows a possible hand-over — a dozen places inside Pong's
        62-instruction scanline loop.  A loop that can be entered in the middle is irreducible control
        flow; LLVM repairs it (FixIrreducible) by routing EVERY edge into such a block, also the
        fall-through from the instruction before it, through a guard header: ~15 flag moves, a chain of
        flag tests and several taken branches per edge, a dozen times per loop iteration (ISA of the
        block after Pong's `PHP` at $F629).  Here such a loop is entered only at its head: the dispatch
        case of an inner instruction sets `sel` and jumps to the head, whose first statement is
        `if (sel) switch (sel) { ... goto inner; }` — forward edges inside the loop, reducible.  Edges
        into the loop's interior from blocks outside it (overlapping decodes, jumps into the middle)
        become returns to the dispatcher.
        A loop qualifies if the linear decode from the target of a backward branch reaches the branch
        (one instruction stream), every other backward branch inside targets the same head, and the
        stream has no BRK / RTS / RTI / JMP () / JAM."""
        heads = {}
        for a in self.code:
            mode, kind, op, b1, b2 = self.code[a]
            if mode == M_REL:
                t = (a + 2 + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff
                if t <= a and t in self.code:
                    heads.setdefault(t, []).append(a)
        cands = []
        for h, latches in heads.items():
            for end in sorted(latches, reverse=True):  # the widest region whose stream is aligned
                stream, a, ok = [], h, True
                while a <= end:
                    if a not in self.code:
                        ok = False
                        break
                    mode, kind, op, b1, b2 = self.code[a]
                    stream.append(a)
                    if op in ('BRK', 'RTS', 'RTI', 'JMPI', 'JAM', 'JMP'):
                        ok = False
                        break
                    a = (a + length(mode)) & 0xffff
                if not ok or stream[-1] != end:
                    continue
                sset = set(stream)
                for x in stream:  # innermost: no backward branch to another head
                    mode, kind, op, b1, b2 = self.code[x]
                    if mode == M_REL:
                        t = (x + 2 + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff
                        if t <= x and t != h:
                            ok = False
                if ok and len(stream) >= 2:
                    cands.append((len(stream), h, stream))
                    break
        # overlapping decodes produce overlapping candidates: the longer stream wins, disjoint ones only
        taken, out = set(), []
        for _, h, stream in sorted(cands, reverse=True):
            if not (set(range(stream[0], stream[-1] + 3)) & taken):
                taken |= set(range(stream[0], stream[-1] + 3))
                out.append((h, stream))
        return out

    def static_succ(self, a):
        """static successors of the instruction at `a`, or None when control can go somewhere unknown"""
        mode, kind, op, b1, b2 = self.code[a]
        nxt = (a + length(mode)) & 0xffff
        if op in ('RTS', 'RTI', 'JMPI', 'BRK', 'JAM'):
            return None
        if op in ('JMP', 'JSR'):   # (a subroutine's RTS is "unknown": the flags a caller defines are live through a call)
            succ = [b1 | (b2 << 8)]
        elif mode == M_REL:
            succ = [(a + 2 + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff, nxt]
        else:
            succ = [nxt]
        return None if any(t not in self.code for t in succ) else succ

    def flag_liveness(self):
        live = {}
        for f in ('V', 'C'):
            rd, kl = F_READS[f], F_KILLS[f]
            li = {a: False for a in self.code}   # least fixed point: grow from the readers
            changed = True
            while changed:
                changed = False
                for a in self.code:
                    if li[a]:
                        continue
                    op = self.code[a][2]
                    if op in rd:
                        v = True
                    elif op in kl:
                        v = False
                    else:
                        succ = self.static_succ(a)
                        v = True if succ is None else any(li[t] for t in succ)
                    if v:
                        li[a] = True
                        changed = True
            live[f] = {a for a, v in li.items() if v}
        return live

    def live_out(self, f, a):
        succ = self.static_succ(a)
        return True if succ is None else any(t in self.flag_live[f] for t in succ)

    def trace_can_exit(self, a):
        """can the trace's copy of the instruction at `a` leave the trace (hand-over, pend, instruction budget, a jump to a
        generic block)?  Read off its emitted text, like entries() does."""
        key = (self.tc.head, a)
        if key not in self._can_exit:
            cur, self._in_analysis = self.cur, True
            try:
                body = ' '.join(self.emit(a))
            finally:
                self.cur, self._in_analysis = cur, False
            self._can_exit[key] = ('return' in body) or ('goto L_' in body)
        return self._can_exit[key]

    def flag_dead_after(self, f, a):
        """may the instruction at `a` (being emitted: generic block or trace copy) leave flag `f` stale?"""
        if not DEAD_FLAGS or self._in_analysis:
            return False
        if self.tc is None:
            return not self.live_out(f, a)
        tc, rd, kl = self.tc, F_READS[f], F_KILLS[f]
        seen = set()

        def dead_from(j):   # is f dead before the trace's copy of instruction j?
            if j not in tc.sset:
                return j in self.code and j not in self.flag_live[f]   # left the trace: the generic code's answer
            if j not in self.flag_live[f]:
                return True     # dead on every static path from here: exits included
            if j in seen:
                return True     # a cycle that met no reader
            seen.add(j)
            mode, kind, op, b1, b2 = self.code[j]
            nxt = (j + length(mode)) & 0xffff
            tS = tc.S.get(j)
            php_tia = op == 'PHP' and tS is not None and not (tS & 0x80) and 0x1d <= (tS & 0x3f) <= 0x1f
            if php_tia:         # reads Z only; its pend exit continues in the generic code behind it
                return nxt in self.code and nxt not in self.flag_live[f] and dead_from(nxt)
            if op in rd:
                return False
            if op in kl:
                return True
            if self.trace_can_exit(j):
                return False    # (live in the generic code from j on, and j can hand over)
            succ = self.static_succ(j)
            return succ is not None and all(dead_from(t) for t in succ)

        succ = self.static_succ(a)
        return succ is not None and all(dead_from(t) for t in succ)

    def stack_hints(self):
        """Likely value of the stack pointer BEFORE each instruction, by an optimistic forward dataflow
        (`LDX #imm; TXS` makes it known, pushes / pulls / JSR move it; at a join the unique known value
        wins, two different ones cancel).  Only a HINT: the emitted fast path is guarded by
        `e.S == hint` and the generic code stays behind it, so a wrong hint costs a compare.  Both
        cartridges point S into TIA space inside their display kernels (`LDX #$1F; TXS`) and `PHP` the
        result of a compare onto ENABL / ENAMx — 273 (Pong) / 95 (Breakout) pushes per frame whose
        address class, register number and no-op test fold to constants once S is known."""
        UNK, CONFLICT = None, -1
        s_in, x_in = {}, {}

        def meet(old, new):
            if new is UNK or old == new:
                return old
            if old is UNK:
                return new
            return CONFLICT

        work = list(self.code)
        for a in work:
            s_in[a], x_in[a] = UNK, UNK
        work = sorted(self.code)  # every address is a block: facts arise wherever `TXS` follows `LDX #imm`
        seen_iter = 0
        while work and seen_iter < 2000000:
            seen_iter += 1
            a = work.pop()
            if a not in self.code:
                continue
            mode, kind, op, b1, b2 = self.code[a]
            S, X = s_in[a], x_in[a]
            if op == 'LDX':
                X = b1 if mode == M_IMM else UNK
            elif op in ('TAX', 'TSX'):
                X = UNK if op == 'TAX' else S
            elif op in ('INX', 'DEX') and X not in (UNK, CONFLICT):
                X = (X + (1 if op == 'INX' else -1)) & 0xff
            elif op == 'TXS':
                S = X
            elif op in ('PHA', 'PHP') and S not in (UNK, CONFLICT):
                S = (S - 1) & 0xff
            elif op in ('PLA', 'PLP') and S not in (UNK, CONFLICT):
                S = (S + 1) & 0xff
            succ = []
            nxt = (a + length(mode)) & 0xffff
            if op == 'JSR':
                succ.append((b1 | (b2 << 8), ((S - 2) & 0xff) if S not in (UNK, CONFLICT) else S, X))
                succ.append((nxt, S, UNK))
            elif op == 'JMP':
                succ.append((b1 | (b2 << 8), S, X))
            elif mode == M_REL:
                succ.append(((a + 2 + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff, S, X))
                succ.append((nxt, S, X))
            elif op not in ('RTS', 'RTI', 'BRK', 'JMPI', 'JAM'):
                succ.append((nxt, S, X))
            for t, ns, nx in succ:
                if t not in self.code:
                    continue
                ms, mx = meet(s_in[t], ns), meet(x_in[t], nx)
                if ms != s_in[t] or mx != x_in[t]:
                    s_in[t], x_in[t] = ms, mx
                    work.append(t)
        return {a: v for a, v in s_in.items() if v not in (UNK, CONFLICT)}

    def entries(self):
        """Addresses native_run can be ENTERED at (the cases of its dispatch switch).  Control only
        comes back from the interpreter after it executed one deferred instruction, so the entry
        points are: the vectors, the successor of every instruction with a deferral path (static
        or dynamic), JSR targets and JSR / BRK return sites.  Every other block is reached only by
        the program's own fall-through / goto edges, which lets the compiler optimise across
        instruction boundaries (flag updates that are overwritten, cycle / counter adds that
        merge) — with every address a switch case, each block had the dispatch as a predecessor.
        A PC outside this set (computed JMP (), RTS tricks) is always safe: the switch returns, the
        interpreter executes that instruction and dispatch is tried again at the next one."""
        ent = {self.word(0xfffc), self.word(0xfffe)} | set(self.traces) | self.indirect_targets
        for a in self.code:
            mode, kind, op, b1, b2 = self.code[a]
            if mode == M_REL or op == 'JMP':
                continue  # their only `return` is the instruction-budget guard
            body = self.emit(a)
            # a `/*rare*/ return` (an operand class the cartridges hardly ever produce: an indexed read
            # landing on a collision latch, (zp),Y into TIA / RIOT space) does NOT make the successor a
            # dispatch entry: after it the interpreter simply keeps stepping until it reaches one.  Every
            # entry inside a 6507 loop costs guard-flag bookkeeping on the loop's hot path (LLVM's
            # FixIrreducible): Breakout 1.97 -> 1.81 ms per agent step.
            if sum(ln.count('return') - ln.count('/*rare*/ return') for ln in body) == 0:
                continue
            if op == 'JSR':
                ent |= {b1 | (b2 << 8), (a + 3) & 0xffff}
            elif op == 'BRK':
                ent.add((a + 2) & 0xffff)
            elif op in ('RTS', 'RTI', 'JMPI', 'JAM'):
                pass
            else:
                ent.add((a + length(mode)) & 0xffff)
        return {a for a in ent if a in self.code}

    # ---- emission ----------------------------------------------------------------------------
    def label(self, a):
        return 'L_%04X' % a

    def inside_edge_ok(self, dst):
        """may the block being emitted jump straight to `dst`?  Not into the interior of a re-entry loop
        from outside its instruction stream (find_loops): that goes back through the dispatcher."""
        i = self.loop_of.get(dst)
        if i is None or dst == self.loops[i][0]:
            return True
        return self.loop_of.get(self.cur) == i

    def goto(self, a):
        if a not in self.code:
            return '{ e.PC = 0x%04x; return; }' % a
        if self.tc is not None:
            if a == self.tc.head:
                return 'goto TL_%04X;' % a
            if a in self.tc.sset and a > self.cur:
                return 'goto TL_%04X;' % a
        elif a in self.traces and self.trace_of.get(self.cur) == a:
            # the generic copy of a traced loop has no back edge: the dispatcher re-enters through the trace's
            # precondition test (which makes this copy acyclic — no irreducible loop for LLVM to repair)
            return '{ e.PC = 0x%04x; e.pend = -2; return; }' % a
        if self.inside_edge_ok(a):
            return 'goto %s;' % self.label(a)
        return '{ e.PC = 0x%04x; /*rare*/ return; }' % a

    def predicted_returns(self, a):
        if not RET_PREDICT or self.tc is not None:
            return []
        sites = sorted(self.returns.get(a, []), reverse=True)
        tot = float(sum(c for c, t in sites)) or 1.0
        out = []
        for c, t in sites[:RET_PREDICT_MAX]:
            if c / tot >= RET_PREDICT_MIN and t in self.code:
                out.append('if (e.PC == 0x%04x) { if (__builtin_expect(n > kNativeInstrLimit, 0)) { e.pend = -2; return; } %s }' % (t, self.goto(t)))
        return out

    def fallback(self, a):
        return ['{ e.PC = 0x%04x; return; }' % a]

    def rd(self, a7):
        """static RAM read (7-bit index): the trace's scalar copy when the loop only reads that byte"""
        if self.tc is not None and a7 in self.tc.hoist:
            return 'h_%02x' % a7
        return 'e.ram_rd(0x%02x)' % a7

    def store_stmt(self, reg, val, pend, dc):
        """a TIA register store with a CONSTANT register number: complete in place (no-op rewrite, logged change,
        HMxx ...: Emu::tia_store) or handed over through `pend` (a block that leaves); `dc` = the instruction's
        cycles, e.cyc still at its start"""
        tc = self.tc
        cw = 'e.cyc + %d' % (dc - 1)
        if tc is None or reg not in tc.shadows or reg in (T_DGRP0, T_DGRP1, T_DENABL):
            return 'if (__builtin_expect(!e.tia_store(0x%02x, %s, %s, %s), 0)) %s' % (reg, val, cw, self.quiet_ok, pend)
        # inside a trace: the no-op test against the loop's scalar shadows of the CPU-side register file
        v = '((%s) & 0xfe)' % val if 0x06 <= reg <= 0x09 else ('((%s) & 0x02)' % val if 0x1d <= reg <= 0x1f else '(%s)' % val)  # Emu::tia_wired
        upd = 'ts_%02x = v_;' % reg
        if reg == 0x1b:      # Emu::tia_store_is_nop: GRP0 also latches the delayed GRP1
            nop = '(ts_1b == v_) & (ts_%02x == ts_1c)' % T_DGRP1
            upd += ' ts_%02x = ts_1c;' % T_DGRP1
        elif reg == 0x1c:    # GRP1 also latches the delayed GRP0 and ENABL
            nop = '(ts_1c == v_) & (ts_%02x == ts_1b) & (ts_%02x == ts_1f)' % (T_DGRP0, T_DENABL)
            upd += ' ts_%02x = ts_1b; ts_%02x = ts_1f;' % (T_DGRP0, T_DENABL)
        else:
            nop = 'ts_%02x == v_' % reg
        quiet = 'false'
        if tc.quiet and 0x1d <= reg <= 0x1f:  # D1 unchanged: the byte is logged without a catch-up request
            quiet = '!((ts_%02x ^ v_) & 0x02)' % reg
        return ('{ const int v_ = %s; if (__builtin_expect(!(%s), 0)) { if (__builtin_expect(e.tia_log(0x%02x, v_, %s, %s), 1)) { %s } else %s } }'
                % (v, nop, reg, cw, quiet, upd, pend))

    def emit_read_op(self, op):
        a = self.cur
        kv, kc = ('false' if self.flag_dead_after('V', a) else 'true'), ('false' if self.flag_dead_after('C', a) else 'true')
        if op in ('ADC', 'SBC'):
            binary = self.tc is not None and self.tc.d_clear
            return 'e.%s%s_f<%s, %s>(m);' % (op.lower(), '_bin' if binary else '', kv, kc)
        if op in ('CMP', 'CPX', 'CPY'):
            return 'e.cmp_f<%s>(e.%s, m);' % (kc, {'CMP': 'A', 'CPX': 'X', 'CPY': 'Y'}[op])
        if op == 'BIT':
            return 'e.bit_f<%s>(m);' % kv
        return {
            'LDA': 'e.A = m; e.set_nz(e.A);', 'LDX': 'e.X = m; e.set_nz(e.X);', 'LDY': 'e.Y = m; e.set_nz(e.Y);',
            'ORA': 'e.A |= m; e.set_nz(e.A);', 'AND': 'e.A &= m; e.set_nz(e.A);', 'EOR': 'e.A ^= m; e.set_nz(e.A);',
            'ADC': 'e.adc(m);', 'SBC': 'e.sbc(m);', 'CMP': 'e.cmp(e.A, m);', 'CPX': 'e.cmp(e.X, m);',
            'CPY': 'e.cmp(e.Y, m);',
            'BIT': 'e.bit(m);',
        }[op]

    def emit(self, a):
        self.cur = a
        mode, kind, op, b1, b2 = self.code[a]
        if self.tc is None and a in self.wait_loops:
            return self.emit_wait_loop(a)
        L = []
        nxt = (a + length(mode)) & 0xffff
        fb = self.fallback(a)
        if kind == K_READ:
            pre = []
            dc = None  # None: dynamic `dc` variable
            if mode == M_IMM:
                pre, dc = ['const int m = 0x%02x;' % b1], 2
            elif mode == M_ZP:
                if b1 < 0x80:
                    # a TIA read register: the collision latches wait for the picture wave to get there
                    # (Emu::tia_read_zp), the input ports are sampled at the read cycle; no interpreter either way
                    pre, dc = ['const int m = e.tia_read_zp(0x%02x, 0x%02x, e.cyc + 2);' % (b1, b1)], 3
                else:
                    pre, dc = ['const int m = %s;' % self.rd(b1 & 0x7f)], 3
            elif mode in (M_ZPX, M_ZPY):
                idx = 'e.X' if mode == M_ZPX else 'e.Y'
                # RAM, or an input port (INPTx: does not depend on the picture; Breakout polls `LDA $38,X`
                # on every line of its kernel); collision latches go to the interpreter
                pre = ['const int ea = (0x%02x + %s) & 0xff;' % (b1, idx), 'int dc = 4, m;',
                       'if (ea & 0x80) m = e.ram_rd(ea & 0x7f);',
                       'else if ((ea & 0x0f) >= 8) { e.cyc += 4; dc = 0; m = e.inpt_read(ea, 0x%02x); }' % b1,
                       'else m = e.tia_read_zp(ea, 0x%02x, e.cyc + 3);' % b1]   # a collision latch (Pong: `LDA CXM0P,X`)
                dc = None
            elif mode == M_ABS:
                ea = b1 | (b2 << 8)
                dc = 4
                if ea & 0x1000:
                    pre = ['const int m = 0x%02x;' % self.byte(ea)]
                elif (ea & 0x280) == 0x80:
                    pre = ['const int m = %s;' % self.rd(ea & 0x7f)]
                elif (ea & 0x280) == 0x280:
                    pre, dc = ['e.cyc += 4;', 'const int m = e.riot_read(0x%04x);' % ea], 0
                elif (ea & 0x0f) >= 8:
                    pre, dc = ['e.cyc += 4;', 'const int m = e.tia_read(0x%04x, 0x%02x);' % (ea, b2)], 0
                else:
                    return fb
            elif mode in (M_ABX, M_ABY):
                idx = 'e.X' if mode == M_ABX else 'e.Y'
                base = b1 | (b2 << 8)
                pre = ['const int ea = (0x%04x + %s) & 0xffff;' % (base, idx),
                       'int dc = 4 + (((ea ^ 0x%04x) & 0xff00) ? 1 : 0);' % base, 'int m;']
                always_rom = (base & 0x1000) and base + 255 <= 0xffff and ((base + 255) >> 12) == (base >> 12)
                if always_rom:
                    pre += ['m = e.rom_byte(ea);']
                else:
                    # only the address classes the 256-byte window base .. base+255 can reach
                    span = [(base + i) & 0xffff for i in range(256)]
                    can_rom = any(x & 0x1000 for x in span)
                    can_ram = any(not (x & 0x1000) and (x & 0x280) == 0x80 for x in span)
                    can_riot = any(not (x & 0x1000) and (x & 0x280) == 0x280 for x in span)
                    arms = []
                    if can_rom:
                        arms.append('if (ea & 0x1000) m = e.rom_byte(ea);')
                    if can_ram:
                        arms.append('if ((ea & 0x%x) == 0x80) m = e.ram_rd(ea & 0x7f);' % (0x1280 if not can_rom else 0x280))
                    if can_riot:
                        arms.append('if ((ea & 0x%x) == 0x280) { e.cyc += dc; dc = 0; m = e.riot_read(ea); }' %
                                    (0x1280 if not can_rom else 0x280))
                    inpt = 'if ((ea & 0x0f) >= 8) { e.cyc += dc; dc = 0; m = e.inpt_read(ea, 0x%02x); }' % b2
                    if not (base & 0x1280) and (base & 0x0f) >= 8:
                        # the base itself is an input port (Pong: `LDA $0038,Y` / `$003a,Y` on every line pair):
                        # that arm first, as the fall-through path
                        arms.insert(0, 'if (__builtin_expect(!(ea & 0x1280) && (ea & 0x08), 1)) { e.cyc += dc; dc = 0; '
                                       'm = e.inpt_read(ea, 0x%02x); }' % b2)
                    arms.append(inpt)
                    pre += [arms[0]] + ['else ' + x for x in arms[1:]] + ['else { --n; e.PC = 0x%04x; /*rare*/ return; }' % a]
            elif mode == M_IZY:
                if b1 < 0x80 or b1 == 0xff:
                    return fb
                pre = [
                    'const int base = %s | (%s << 8);' % (self.rd(b1 & 0x7f), self.rd((b1 + 1) & 0x7f)),
                    'const int ea = (base + e.Y) & 0xffff;', 'const int dc = 5 + (((ea ^ base) & 0xff00) ? 1 : 0);']
                if self.tc is not None and b1 in self.tc.rom_ptrs:
                    # the trace's precondition: base .. base+255 inside the cartridge.  The byte comes from LDS
                    # (ds_read + wait + v_readfirstlane: ~130 clocks for one wave); a table walked by a slowly moving
                    # index (Pong's playfield rows: Y = scanline / 8) is read again and again at the same address, so
                    # each site remembers its last (address, byte) in two scalar registers
                    pre += ['int m;', 'if (__builtin_expect(ea == rc_%04x_ea, 1)) m = rc_%04x_m;' % (a, a),
                            'else { m = e.rom_byte(ea); rc_%04x_ea = ea; rc_%04x_m = m; }' % (a, a)]
                else:
                    pre += ['int m;', 'if (ea & 0x1000) m = e.rom_byte(ea);',
                            'else if ((ea & 0x280) == 0x80) m = e.ram_rd(ea & 0x7f);',
                            'else { --n; e.PC = 0x%04x; /*rare*/ return; }' % a]
            elif mode == M_PULL and op == 'PLA':
                # pull from a stack in RAM (a pull from TIA space reads collision latches: interpreter)
                # ... or from an input-port address in TIA space (Breakout pulls from $1F inside its
                # kernel: the read returns the bus noise = the next opcode byte and needs no picture)
                tS = self.tc.S.get(a) if self.tc is not None else None
                if tS is not None:
                    k1 = (tS + 1) & 0xff
                    if k1 & 0x80:
                        L += ['const int m = e.ram_rd(0x%02x);' % (k1 & 0x7f), 'e.S = 0x%02x;' % k1, 'e.A = m; e.set_nz(e.A);',
                              'e.cyc += 4;']
                        return L
                    if (k1 & 0x0f) >= 8:
                        L += ['e.cyc += 4;', 'const int m = e.tia_read(0x%02x, 0x%02x);' % (k1, b1), 'e.S = 0x%02x;' % k1,
                              'e.A = m; e.set_nz(e.A);']
                        return L
                    return fb
                pre = ['const int s1 = (e.S + 1) & 0xff;', 'int dc = 4, m;']
                h = self.s_hint.get(a)
                hs1 = None if h is None else (h + 1) & 0xff
                if hs1 is not None and not (hs1 & 0x80) and (hs1 & 0x0f) >= 8:
                    # known (guarded) pull from an input-port address: Emu::tia_read folds to constants
                    pre.append('if (__builtin_expect(s1 == 0x%02x, 1)) { e.cyc += 4; dc = 0; m = e.tia_read(0x%02x, 0x%02x); } else'
                               % (hs1, hs1, b1))
                pre += ['if (s1 & 0x80) m = e.ram_rd(s1 & 0x7f);',
                        'else if ((s1 & 0x0f) >= 8) { e.cyc += 4; dc = 0; m = e.tia_read(s1, 0x%02x); }' % b1,
                        'else { --n; e.PC = 0x%04x; return; }' % a, 'e.S = s1;']
                dc = None
                op = 'LDA'
            else:
                return fb  # (zp,X), PLP
            L += pre
            L.append(self.emit_read_op(op))
            if dc is None:
                L.append('e.cyc += dc;')
            elif dc:
                L.append('e.cyc += %d;' % dc)
            return L
        if kind == K_NONE:
            simple = {
                'ASL_A': 'e.cf = e.A >> 7; e.A = (e.A << 1) & 0xff; e.set_nz(e.A);',
                'LSR_A': 'e.cf = e.A & 1; e.A = e.A >> 1; e.set_nz(e.A);',
                'ROL_A': '{ const int c = e.cf; e.cf = e.A >> 7; e.A = ((e.A << 1) | c) & 0xff; e.set_nz(e.A); }',
                'ROR_A': '{ const int c = e.cf; e.cf = e.A & 1; e.A = (e.A >> 1) | (c << 7); e.set_nz(e.A); }',
                'INX': 'e.X = (e.X + 1) & 0xff; e.set_nz(e.X);', 'INY': 'e.Y = (e.Y + 1) & 0xff; e.set_nz(e.Y);',
                'DEX': 'e.X = (e.X - 1) & 0xff; e.set_nz(e.X);', 'DEY': 'e.Y = (e.Y - 1) & 0xff; e.set_nz(e.Y);',
                'TAX': 'e.X = e.A; e.set_nz(e.X);', 'TAY': 'e.Y = e.A; e.set_nz(e.Y);',
                'TXA': 'e.A = e.X; e.set_nz(e.A);', 'TYA': 'e.A = e.Y; e.set_nz(e.A);',
                'TSX': 'e.X = e.S; e.set_nz(e.X);', 'TXS': 'e.S = e.X;', 'CLC': 'e.cf = 0;', 'SEC': 'e.cf = 1;',
                'CLI': 'e.P &= ~FI;', 'SEI': 'e.P |= FI;', 'CLV': 'e.P &= ~FV;', 'CLD': 'e.P &= ~FD;',
                'SED': 'e.P |= FD;', 'NOP': '',
            }
            if op in simple:
                txt = simple[op]
                if op in F_KILLS['C'] and self.flag_dead_after('C', a):
                    txt = re.sub(r'e\.cf = [^;]*; ?', '', txt)
                if op == 'CLV' and self.flag_dead_after('V', a):
                    txt = ''
                return [txt, 'e.cyc += 2;']
            if mode == M_REL:
                flag, want = {
                    'BPL': ('FN', 0), 'BMI': ('FN', 1), 'BVC': ('FV', 0), 'BVS': ('FV', 1), 'BCC': ('FC', 0),
                    'BCS': ('FC', 1), 'BNE': ('FZ', 0), 'BEQ': ('FZ', 1)
                }[op]
                npc = (a + 2) & 0xffff
                tgt = (npc + (b1 - 256 if b1 & 0x80 else b1)) & 0xffff
                dc = 2 + (2 if ((tgt ^ npc) & 0xff00) else 1)
                test = {'FN': '(e.nv & 0x80)', 'FZ': '((e.zv & 0xff) == 0)', 'FC': 'e.cf', 'FV': '(e.P & FV)'}[flag]
                cond = test if want else '!' + test
                body = 'e.cyc += %d; ' % dc
                # (measured: leaving on every backward edge and dispatching again — which makes native_run
                # acyclic and spares the loop-entry guard flags LLVM's FixIrreducible / UnifyLoopExits add to
                # blocks inside 6507 loops that are also dispatch entries — is slower: Pong 1.20 -> 1.43 ms)
                if tgt <= a:  # backward edge: the only place a frame can loop without bound
                    body += 'if (__builtin_expect(n > kNativeInstrLimit, 0)) { e.PC = 0x%04x; return; } ' % tgt
                    if self.tc is not None and TRACE_FLUSH and tgt == self.tc.head:
                        # the trace runs a whole display kernel without leaving: hand wave B what has been recorded so far
                        body += 'if (__builtin_expect(e.wqn >= %d, 0)) e.rq_flush(); ' % TRACE_FLUSH
                body += self.goto(tgt)
                pt = self.branch_prob.get(a)
                if pt is not None:
                    cond = '__builtin_expect_with_probability((long)(%s), 1, %.4f)' % ('(%s) != 0' % cond, min(max(pt, 0.0001), 0.9999))
                return ['if (%s) { %s }' % (cond, body), 'e.cyc += 2;']
            if op == 'JMP':
                tgt = b1 | (b2 << 8)
                return ['e.cyc += 3;', 'if (__builtin_expect(n > kNativeInstrLimit, 0)) { e.PC = 0x%04x; return; }' % tgt, self.goto(tgt)]
            # ---- subroutine linkage with the stack in RAM (cycle totals as in Emu::step; no bus access
            # in between that could observe the intermediate cycles).  RTS / RTI have a run-time target:
            # they set PC and return with pend = -2 ("dispatch again, nothing to interpret"); an
            # in-function `goto` back to the dispatch switch made the object 4.5x larger and the
            # compile 20x slower (the switch becomes a loop header inside a 2000-block function).  With the stack pointer in TIA
            # space (Breakout's `LDX #$1F; TXS` kernel trick) the interpreter does it.
            if op == 'JSR':
                tgt, ret = b1 | (b2 << 8), (a + 2) & 0xffff
                return ['if (__builtin_expect(e.S < 0x81, 0)) { --n; e.PC = 0x%04x; return; }' % a,
                        'e.ram_wr(e.S & 0x7f, 0x%02x); e.ram_wr((e.S - 1) & 0x7f, 0x%02x); e.S = (e.S - 2) & 0xff;' %
                        (ret >> 8, ret & 0xff), 'e.cyc += 6;',
                        'if (__builtin_expect(n > kNativeInstrLimit, 0)) { e.PC = 0x%04x; return; }' % tgt, self.goto(tgt)]
            if op == 'RTS':
                return ['if (__builtin_expect(e.S < 0x7f || e.S > 0xfd, 0)) { --n; e.PC = 0x%04x; return; }' % a,
                        'e.PC = ((e.ram_rd((e.S + 1) & 0x7f) | (e.ram_rd((e.S + 2) & 0x7f) << 8)) + 1) & 0xffff;',
                        'e.S = (e.S + 2) & 0xff; e.cyc += 6;'] + self.predicted_returns(a) + ['e.pend = -2; return;']
            if op == 'RTI':
                return ['if (__builtin_expect(e.S < 0x7f || e.S > 0xfc, 0)) { --n; e.PC = 0x%04x; return; }' % a,
                        'e.pset((e.ram_rd((e.S + 1) & 0x7f) & ~FB) | FU);',
                        'e.PC = e.ram_rd((e.S + 2) & 0x7f) | (e.ram_rd((e.S + 3) & 0x7f) << 8);',
                        'e.S = (e.S + 3) & 0xff; e.cyc += 6;'] + self.predicted_returns(a) + ['e.pend = -2; return;']
            if op == 'BRK':
                ret, vec = (a + 2) & 0xffff, self.word(0xfffe)
                return ['if (__builtin_expect(e.S < 0x82, 0)) { --n; e.PC = 0x%04x; return; }' % a,
                        'e.ram_wr(e.S & 0x7f, 0x%02x); e.ram_wr((e.S - 1) & 0x7f, 0x%02x);' % (ret >> 8, ret & 0xff),
                        'e.ram_wr((e.S - 2) & 0x7f, e.pfull() | FB | FU); e.S = (e.S - 3) & 0xff; e.P |= FI; e.cyc += 7;',
                        'if (__builtin_expect(n > kNativeInstrLimit, 0)) { e.PC = 0x%04x; return; }' % vec, self.goto(vec)]
            return fb  # JMP () / JAM
        # ---- stores and read-modify-writes (zero-page class only, as in step_fast) ----
        if mode == M_ZP:
            ea, dc, static = '0x%02x' % b1, 3, b1
        elif mode == M_ZPX:
            ea, dc, static = '((0x%02x + e.X) & 0xff)' % b1, 4, None
        elif mode == M_ZPY:
            ea, dc, static = '((0x%02x + e.Y) & 0xff)' % b1, 4, None
        elif mode == M_PUSH:
            ea, dc, static = 'e.S', 3, None
        elif mode in (M_ABS, M_ABX, M_ABY) and kind == K_WRITE:
            # absolute stores (the RIOT timer, RAM / TIA through their mirrors): the address classes in the order of
            # Emu::step's stage D; an indexed store always takes its extra cycle
            val = {'STA': 'e.A', 'STX': 'e.X', 'STY': 'e.Y'}[op]
            base = b1 | (b2 << 8)
            dc = 4 if mode == M_ABS else 5
            eaexp = '0x%04x' % base if mode == M_ABS else '((0x%04x + %s) & 0xffff)' % (base, 'e.X' if mode == M_ABX else 'e.Y')
            pend = '{ --n; e.cyc += %d; e.pend = (ea & 0xff) | ((%s) << 8); e.PC = 0x%04x; return; }' % (dc - 1, val, nxt)
            return ['const int ea = %s;' % eaexp,
                    'if (ea & 0x1000) { e.cyc += %d; }' % dc,
                    'else if (!(ea & 0x80)) { if (__builtin_expect(!e.tia_store(ea & 0x3f, %s, e.cyc + %d, %s), 0)) %s e.cyc += %d; '
                    'if (__builtin_expect(e.stop, 0)) { e.PC = 0x%04x; e.pend = -2; return; } }' % (val, dc - 1, self.quiet_ok, pend, dc, nxt),
                    'else if (!(ea & 0x200)) { e.ram_wr(ea & 0x7f, %s); e.cyc += %d; }' % (val, dc),
                    'else { e.cyc += %d; e.riot_write(ea, %s); }' % (dc, val)]
        else:
            return fb
        if kind == K_WRITE:
            val = {'STA': 'e.A', 'STX': 'e.X', 'STY': 'e.Y', 'PHA': 'e.A', 'PHP': '(e.pfull() | FB | FU)'}[op]
            dec_s = ' e.S = (e.S - 1) & 0xff;' if mode == M_PUSH else ''
            # A real TIA register change: the block finishes what step_fast does for the store (S,
            # cycles up to the write cycle, PC) and hands (address, value) to the interpreter's TIA
            # stages through e.pend — no opcode fetch / decode there (Emu::step).
            pend = '{ --n;%s e.cyc += %d; e.pend = %%s | ((%s) << 8); e.PC = 0x%04x; return; }' % (dec_s, dc - 1, val, nxt)
            if static is not None:
                if static & 0x80:
                    return ['e.ram_wr(0x%02x, %s);' % (static & 0x7f, val), 'e.cyc += %d;' % dc]
                reg = static & 0x3f
                if reg == 0x02:  # WSYNC
                    return ['e.wsync(e.cyc + %d);' % dc]
                out = [self.store_stmt(reg, val, pend % ('0x%02x' % static), dc), 'e.cyc += %d;' % dc]
                if reg == 0x00:  # VSYNC may have ended the frame (Emu::tia_store sets `stop`): leave, nothing to interpret
                    out.append('if (e.stop) { e.PC = 0x%04x; e.pend = -2; return; }' % nxt)
                return out
            generic = [
                'const int ea = %s;' % ea,
                'if (ea & 0x80) e.ram_wr(ea & 0x7f, %s);' % val,
                'else if (__builtin_expect(!e.tia_store(ea & 0x3f, %s, e.cyc + %d, %s), 0)) %s' % (val, dc - 1, self.quiet_ok, pend % 'ea'),
                ('%s e.cyc += %d;' % (dec_s, dc)).strip(),
                'if (__builtin_expect(e.stop, 0)) { e.PC = 0x%04x; e.pend = -2; return; }' % nxt   # a run-time address can be VSYNC
            ]
            tS = self.tc.S.get(a) if (self.tc is not None and mode == M_PUSH) else None
            if tS is not None and not ((tS & 0x3f) == 0x02 and not (tS & 0x80)):
                # inside a trace the stack pointer is a FACT (precondition at the head): no guard, no generic arm
                if tS & 0x80:
                    return ['e.ram_wr(0x%02x, %s);' % (tS & 0x7f, val), 'e.S = 0x%02x; e.cyc += %d;' % ((tS - 1) & 0xff, dc)]
                return [self.store_stmt(tS & 0x3f, val, pend % ('0x%02x' % tS), dc), 'e.S = 0x%02x; e.cyc += %d;' % ((tS - 1) & 0xff, dc)]
            h = self.s_hint.get(a) if mode == M_PUSH else None
            if h is None or (h & 0x3f) == 0x02 and not (h & 0x80):
                return generic
            # stack pointer known (guarded): address class, register number and the no-op test fold
            if h & 0x80:
                fast = ['e.ram_wr(0x%02x, %s);' % (h & 0x7f, val)]
            else:
                fast = [self.store_stmt(h & 0x3f, val, pend % ('0x%02x' % h), dc)]
            fast.append('e.S = 0x%02x; e.cyc += %d;' % ((h - 1) & 0xff, dc))
            return ['if (__builtin_expect(e.S == 0x%02x, 1)) { %s } else { %s }' % (h, ' '.join(fast), ' '.join(generic))]
        # K_RMW
        rmw = {
            'ASL': 'e.cf = m >> 7; wv = (m << 1) & 0xff;',
            'LSR': 'e.cf = m & 1; wv = m >> 1;',
            'ROL': '{ const int c = e.cf; e.cf = m >> 7; wv = ((m << 1) | c) & 0xff; }',
            'ROR': '{ const int c = e.cf; e.cf = m & 1; wv = (m >> 1) | (c << 7); }',
            'INC': 'wv = (m + 1) & 0xff;', 'DEC': 'wv = (m - 1) & 0xff;',
        }[op]
        if static is not None and not (static & 0x80):
            return fb
        if op in F_KILLS['C'] and self.flag_dead_after('C', a):
            rmw = re.sub(r'e\.cf = [^;]*; ?', '', rmw)
        return [
            'const int ea = %s;' % ea, 'if (!(ea & 0x80)) { --n; e.PC = 0x%04x; return; }' % a,
            'const int m = e.ram_rd(ea & 0x7f);', 'int wv;', rmw, 'e.set_nz(wv);', 'e.ram_wr(ea & 0x7f, wv);',
            'e.cyc += %d;' % (dc + 2)
        ]

    def emit_trace(self, tr):
        """the specialised copy of a hot loop (see TRACE_LOOPS), placed at the loop head's label"""
        out = ['  %s: {  // ---- trace of the loop %04x .. %04x (%d instructions)' % (self.label(tr.head), tr.head, tr.stream[-1], len(tr.stream))]
        for ln in tr.prologue():
            out.append('    ' + ln)
        out.append('    if (__builtin_expect(%s, 1)) {' % tr.precondition())
        self.tc = tr
        try:
            for i, a in enumerate(tr.stream):
                mode, kind, op, b1, b2 = self.code[a]
                self.cur = a
                body = self.emit(a)
                is_fb = len(body) == 1 and body[0].startswith('{ e.PC')
                out.append('      TL_%04X: {  // %s mode %d' % (a, op, mode))
                if a == tr.head:
                    out.append('        PARLHIP_TRACE_ITER(0x%04x);' % a)  # empty on the device; the host harness counts
                if MARKERS:
                    out.append('        asm volatile("; @@TRC %04x");' % a)
                if is_fb:
                    out.append('        ' + body[0])
                else:
                    out.append('        ++n;')
                    for ln in body:
                        if ln:
                            out.append('        ' + ln)
                out.append('      }')
                terminal = op in ('JMP', 'JAM', 'BRK', 'RTS', 'RTI', 'JMPI', 'JSR') or is_fb
                nxt = (a + length(mode)) & 0xffff
                if i + 1 == len(tr.stream) and not terminal:
                    self.tc = None  # the edge that leaves the trace at its end is a generic edge
                    self.cur = None
                    out.append('      ' + self.goto(nxt))
        finally:
            self.tc = None
        out.append('    }')
        out.append('  }')
        return out

    def source(self, game_const):
        addrs = sorted(self.code)
        out = []
        crc = zlib.crc32(bytes(self.rom)) & 0xffffffff
        out.append('// ---- %s: %d instruction blocks, rom crc32 %08x ----' % (self.name, len(addrs), crc))
        out.append('template <> struct NativeCart<%s> { static constexpr bool present = true; '
                   'static constexpr uint32_t rom_crc32 = 0x%08xu; };' % (game_const, crc))
        out.append('template <> DEVI void native_run<%s>(Emu& e, int& n) {' % game_const)
        out.append('  if (n > kNativeInstrLimit) return;')
        if self.loops:
            out.append('  int sel = 0;  // re-entry into the middle of a loop goes through its head (find_loops)')
        hot = sorted(self.traces, key=lambda h: -max(self.branch_taken.get(x, 0) for x in self.traces[h].stream))
        for h in hot[:3]:  # the hottest loop heads before the compare tree of the switch (every dirty iteration re-enters here)
            out.append('  if (e.PC == 0x%04x) goto %s;' % (h, self.label(h)))
        out.append('  switch (e.PC) {')
        entries = self.entries()
        sel_cases = {}  # loop index -> [(sel value, address)]
        for a in sorted(entries):
            i = self.loop_of.get(a)
            if i is not None and a != self.loops[i][0]:
                lst = sel_cases.setdefault(i, [])
                lst.append((len(lst) + 1, a))
                out.append('    case 0x%04x: sel = %d; goto %s;' % (a, len(lst), self.label(self.loops[i][0])))
            else:
                out.append('    case 0x%04x: goto %s;' % (a, self.label(a)))
        out.append('    default: return;')
        out.append('  }')
        head_switch = {}
        for i, lst in sel_cases.items():
            head_switch[self.loops[i][0]] = ('    if (__builtin_expect(sel != 0, 0)) { const int s_ = sel; sel = 0; switch (s_) { %s default: break; } }'
                                             % ' '.join('case %d: goto %s;' % (k, self.label(a)) for k, a in lst))
        native = 0
        for i, a in enumerate(addrs):
            mode, kind, op, b1, b2 = self.code[a]
            body = self.emit(a)
            is_fb = len(body) == 1 and body[0].startswith('{ e.PC')
            native += 0 if is_fb else 1
            if a in self.traces:
                out += self.emit_trace(self.traces[a])
                out.append('  {  // %s mode %d (generic copy of the trace head)' % (op, mode))
            else:
                out.append('  %s: {  // %s mode %d' % (self.label(a), op, mode))
            if a in head_switch:
                out.append(head_switch[a])
            if MARKERS:
                out.append('    asm volatile("; @@BLK %04x");' % a)
            if is_fb:
                out.append('    ' + body[0])
            else:
                out.append('    ++n;')
                for ln in body:
                    if ln:
                        out.append('    ' + ln)
            out.append('  }')
            nxt = (a + length(mode)) & 0xffff
            terminal = op in ('JMP', 'JAM', 'BRK', 'RTS', 'RTI', 'JMPI', 'JSR') or is_fb
            self.cur = a
            if not terminal and nxt not in self.code:
                out.append('  { e.PC = 0x%04x; return; }' % nxt)
            elif not terminal and not self.inside_edge_ok(nxt):
                out.append('  { e.PC = 0x%04x; return; }' % nxt)  # fall-through into a loop's interior from outside it
            elif not terminal and (i + 1 >= len(addrs) or addrs[i + 1] != nxt):
                out.append('  goto %s;' % self.label(nxt))
        out.append('}')
        out.insert(1, '// natively translated: %d, deferred to the interpreter: %d' % (native, len(addrs) - native))
        return '\n'.join(out)


HEADER = '''// cart_native.gen.hpp — GENERATED by gen_cart_native.py from the cartridges in roms/ (do not edit,
// do not commit: derived from user-supplied ROM data).  Included by atari_core.hpp.
#pragma once
'''


def main():
    out_path = sys.argv[1]
    parts = [HEADER]
    consts = {'pong': 'GAME_PONG', 'breakout': 'GAME_BREAKOUT'}
    for spec in sys.argv[2:]:
        name, path = spec.split('=', 1)
        try:
            rom = open(path, 'rb').read()
        except OSError:
            continue
        parts.append(Cart(name, rom).source(consts[name]))
    text = '\n'.join(parts) + '\n'
    try:
        if open(out_path).read() == text:
            os.utime(out_path)  # make must see the target as remade, or it stays stale for ever
            return
    except OSError:
        pass
    open(out_path, 'w').write(text)


if __name__ == '__main__':
    main()

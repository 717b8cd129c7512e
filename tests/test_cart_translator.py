"""CPU checks of the static cartridge translator (parl_amd/csrc/gen_cart_native.py): structural
properties of the dispatch-entry set, and — by replaying an instruction trace of the CPU oracle —
that restricting the dispatch switch to those entries keeps execution inside translated code
(a PC outside the set is always safe but interpreted).  Needs the cartridges
(roms/, provisioned by __graft_entry__.build()); skipped without them."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'parl_amd', 'csrc'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))


def _cart(name):
    import gen_cart_native as g
    path = os.path.join(ROOT, 'roms', name + '.bin')
    if not os.path.exists(path):
        pytest.skip('cartridge %s not provisioned' % name)
    return g, g.Cart(name, open(path, 'rb').read())


@pytest.mark.parametrize('name', ['pong', 'breakout'])
def test_entry_set_structure(name):
    g, c = _cart(name)
    ent = c.entries()
    assert ent <= set(c.code)                       # every case label has a block
    assert c.word(0xfffc) in ent                    # reset vector
    assert len(ent) < len(c.code) // 2              # the point: most blocks are NOT dispatch targets
    for a, (mode, kind, op, b1, b2) in c.code.items():
        body = c.emit(a)
        if op == 'JSR':
            assert (b1 | (b2 << 8)) in ent or (b1 | (b2 << 8)) not in c.code
            assert ((a + 3) & 0xffff) in ent or ((a + 3) & 0xffff) not in c.code
        elif mode != g.M_REL and op not in ('JMP', 'BRK', 'RTS', 'RTI', 'JMPI', 'JAM') and \
                sum(ln.count('return') - ln.count('/*rare*/ return') for ln in body) > 0:
            # (a `/*rare*/ return` deliberately creates no entry: the interpreter keeps stepping to the next one)
            nxt = (a + g.length(mode)) & 0xffff     # execution resumes here after the deferral
            assert nxt in ent or nxt not in c.code, hex(a)
    src = c.source('GAME_PONG' if name == 'pong' else 'GAME_BREAKOUT')
    assert src.count('case 0x') == len(ent)
    assert 'e.pend = ' in src                        # real TIA stores are handed over, not re-decoded
    assert set(c.traces) <= ent                      # a hot loop's trace is entered through the dispatcher


def test_hot_loops_are_emitted_as_traces():
    """gen_cart_native TRACE_LOOPS: Pong's scanline loop ($F5E0 .. $F63C, 91 iterations per frame) gets a specialised
    single-entry copy in front of its generic blocks — stack pointer, binary mode and the (zp),Y pointers established
    once at the head, the zero-page bytes the loop only reads hoisted, scalar shadows of the TIA registers its stores
    are compared with — and the generic copy's back edge returns to the dispatcher (no cycle left in it)"""
    g, c = _cart('pong')
    assert 0xf5e0 in c.traces
    tr = c.traces[0xf5e0]
    assert tr.stream[0] == 0xf5e0 and tr.stream[-1] == 0xf63c and len(tr.stream) == 57
    assert tr.use_S and tr.S[0xf5e0] == 0x1d and tr.S[0xf61e] == 0x1f and tr.S[0xf629] == 0x1e   # PHP -> ENAM0 / ENABL / ENAM1
    assert tr.d_clear and tr.rom_ptrs == [0x9b, 0x9d, 0x9f]
    assert {0x33, 0x26, 0x34, 0x27, 0x00} <= tr.hoist and not ({0x04, 0x05} & tr.hoist)   # $84 / $85 are written in the loop
    assert {0x1b, 0x1c, 0x1d, 0x1e, 0x1f} <= tr.shadows
    src = c.source('GAME_PONG')
    assert 'trace of the loop f5e0 .. f63c' in src and 'TL_F5E0:' in src and 'goto TL_F5E0;' in src
    assert 'e.sbc_bin_f<false, false>(m);' in src and 'const int m = h_33;' in src   # (V and C dead after it: round 6's flag liveness)
    assert '{ e.PC = 0xf5e0; e.pend = -2; return; }' in src          # the generic copy's back edge
    assert 'if (e.PC == 0xf5e0) goto L_F5E0;' in src                  # the hottest head ahead of the switch's compare tree
    i0 = src.index('TL_F5E0:')
    body = src[i0:src.index('generic copy of the trace head', i0)]
    assert 'e.S ==' not in body and body.count('TL_') > 57           # no stack-pointer guards inside the trace
    # breakout: its display loops index RAM with X and write through X (`DEC zp,X`): nothing is hoisted there, the
    # stack-pointer / binary-mode facts and the single entry remain
    g, b = _cart('breakout')
    assert {0xf040, 0xf0b1} <= set(b.traces) and not b.traces[0xf040].hoist and b.traces[0xf0b1].use_S


def test_tia_stores_are_recorded_and_timer_wait_loops_run_in_place():
    """round 4: a TIA store of the translated code is Emu::tia_store (no-op rewrite, or a record for the picture wave);
    only a full local log hands over through `pend`, and a VSYNC store leaves when it ended the frame; the
    `LDA INTIM; BNE back` wait loops of both cartridges are emitted as in-place loops that skip the iterations whose
    outcome is known"""
    g, c = _cart('pong')
    src = c.source('GAME_PONG')
    assert 'e.tia_store(0x' in src and 'tia_store_quiet' not in src and 'pf_enqueue' not in src
    assert 'if (e.stop) { e.PC = 0x' in src
    assert set(c.wait_loops) == {0xf094, 0xf21f} and c.wait_loops[0xf21f] == (0x0284, 7)
    assert 'timer wait loop (LDA INTIM; BNE back)' in src and 'const int bound = (e.timer - 2) << e.timer_shift;' in src
    g, b = _cart('breakout')
    assert {0xf33a, 0xf611} <= set(b.wait_loops)


@pytest.mark.parametrize('name,max_wait', [('pong', 10.0), ('breakout', 10.0)])
def test_restricted_dispatch_keeps_execution_translated(tmp_path, name, max_wait):
    """replay 1600 frames of the oracle's instruction trace against the entry set: instructions that
    would be interpreted only because the PC is not an entry must stay negligible"""
    g, c = _cart(name)
    import oracle_profile
    oracle_profile.main(str(tmp_path))
    tr = np.fromfile(os.path.join(str(tmp_path), name + '.trace'), dtype=np.uint16).reshape(-1, 2)
    frames = 1600
    bodies = {a: c.emit(a) for a in c.code}
    static_fb = {a for a, b in bodies.items() if len(b) == 1 and b[0].startswith('{ e.PC')}
    dyn = {a for a, b in bodies.items() if a not in static_fb and c.code[a][0] != g.M_REL and c.code[a][2] != 'JMP'
           and any('return' in ln for ln in b)}
    ent = c.entries()
    native, deferrals, waiting, translated = False, 0, 0, 0
    for pc, real in tr.tolist():
        if not native:
            if pc in ent:
                native = True
            else:
                waiting += 1
                continue
        if pc in static_fb or (real and pc in dyn):
            deferrals += 1
            native = False
            continue
        translated += 1
    total = len(tr)
    assert waiting / frames <= max_wait, (waiting / frames, deferrals / frames)
    assert translated / total > 0.9, translated / total


def test_single_entry_loops_are_found():
    """gen_cart_native.Cart.find_loops (switched off in the product build, kept alive here): Pong's
    scanline loop and Breakout's two display loops qualify, streams are aligned, heads are loop targets"""
    import importlib
    os.environ['PARLHIP_LOOP_REENTRY'] = 'pong,breakout'
    try:
        import gen_cart_native as g
        g = importlib.reload(g)
        for name, want in (('pong', [(0xf5e0, 0xf63c)]), ('breakout', [(0xf040, 0xf098), (0xf0b1, 0xf0eb)])):
            path = os.path.join(ROOT, 'roms', name + '.bin')
            if not os.path.exists(path):
                pytest.skip('cartridge %s not provisioned' % name)
            c = g.Cart(name, open(path, 'rb').read())
            got = {(h, st[-1]) for h, st in c.loops}
            for w in want:
                assert w in got, (name, [(hex(a), hex(b)) for a, b in sorted(got)])
            for h, st in c.loops:
                assert st[0] == h and all(c.loop_of[x] == c.loop_of[h] for x in st)
            src = c.source('GAME_PONG' if name == 'pong' else 'GAME_BREAKOUT')
            assert 'sel = 1; goto L_%04X;' % want[0][0] in src and 'switch (s_)' in src
    finally:
        del os.environ['PARLHIP_LOOP_REENTRY']
        importlib.reload(g)


@pytest.mark.parametrize('name,game,loops', [('pong', 1, ''), ('breakout', 2, ''), ('pong', 1, 'pong,breakout'),
                                             ('breakout', 2, 'pong,breakout')])
def test_translated_cartridge_on_host_equals_oracle(tmp_path, name, game, loops):
    """The generated cartridge code compiled for the HOST (tests/tools/cart_host: the generator's
    emulator surface implemented on the oracle's machine state) against the oracle's own
    atari_frame(), 600 frames with paddle / fire / RESET inputs: CPU registers, RAM, cycle counters,
    TIA / RIOT state, collision latches and the frame buffer identical after every frame.  Checks
    every translated addressing mode / operation / cycle count / branch target, the dispatch-entry
    set, the TIA-store hand-over and the no-op-write classification without a GPU."""
    import subprocess
    g, c = _cart(name)
    d = str(tmp_path)
    src = os.path.join(ROOT, 'tests', 'tools', 'cart_host')
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'parl_amd', 'csrc', 'gen_cart_native.py'),
                           os.path.join(d, 'cart_native.gen.hpp'),
                           'pong=' + os.path.join(ROOT, 'roms', 'pong.bin'),
                           'breakout=' + os.path.join(ROOT, 'roms', 'breakout.bin')],
                          env=dict(os.environ, PARLHIP_LOOP_REENTRY=loops))  # '' = the product build's setting
    subprocess.check_call(['gcc', '-O1', '-std=c11', '-ffp-contract=off', '-c', os.path.join(src, 'shim.c'), '-o',
                           os.path.join(d, 'shim.o')])
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-I', d, '-c', os.path.join(src, 'main.cpp'), '-o',
                           os.path.join(d, 'main.o')])
    subprocess.check_call(['g++', os.path.join(d, 'shim.o'), os.path.join(d, 'main.o'), '-lm', '-o',
                           os.path.join(d, 'cart_host')])
    p = subprocess.run([os.path.join(d, 'cart_host'), os.path.join(ROOT, 'roms', name + '.bin'), str(game), '600'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, CART_HOST_DEFERRALS='1'))
    assert p.returncode == 0, p.stdout[-2000:]
    assert 'frames identical' in p.stdout
    translated = float(p.stdout.split('identical;')[1].split()[0])
    assert translated > 3000          # execution really goes through the translated blocks
    # what is left to the interpreter (round 4: absolute stores and zero-page TIA reads are translated): the
    # instructions behind Pong's one `JMP ()` per game tick, nothing in Breakout
    interpreted = sum(float(ln.split(':')[1].split()[0]) for ln in p.stdout.splitlines() if ln.startswith('interpreted '))
    if not loops:                     # the product build's dispatch-entry set
        assert interpreted <= (5.0 if name == 'pong' else 0.5), p.stdout[-1500:]
    if name == 'pong' and not loops:  # the product build: the scanline loop runs as a trace (counted by the harness)
        it = [float(ln.split(':')[1].split()[0]) for ln in p.stdout.splitlines() if ln.startswith('trace f5e0')]
        assert it and it[0] > 80, p.stdout[-500:]


def _fn_body(text, signature):
    """source of the function starting at `signature` (brace matched), normalised"""
    import re
    i = text.index(signature)
    j = text.index('{', i)
    depth, k = 0, j
    while True:
        depth += text[k] == '{'
        depth -= text[k] == '}'
        k += 1
        if depth == 0:
            break
    body = text[j:k]
    body = re.sub(r'//[^\n]*', '', body)
    body = re.sub(r'__builtin_expect\((.*?), 0\)', r'\1', body)
    return re.sub(r'\s+', '', body)


def test_host_harness_alu_is_the_device_text_and_matches_oracle(tmp_path):
    """(1) the flag arithmetic of tests/tools/cart_host/main.cpp is textually the device's
    (atari_core.hpp adc / sbc / cmp / bit / set_nz / pfull / pset) — the harness cannot drift;
    (2) that arithmetic equals the oracle's ADC / SBC / CMP for every accumulator, operand, carry and
    decimal-mode combination (3 x 2 x 2 x 65,536 cases)."""
    import subprocess
    dev = open(os.path.join(ROOT, 'parl_amd', 'csrc', 'atari_core.hpp')).read()
    host = open(os.path.join(ROOT, 'tests', 'tools', 'cart_host', 'main.cpp')).read()
    for sig in ('void adc(int m)', 'void sbc(int m)', 'void cmp(int r, int m)', 'void bit(int m)', 'void set_nz(int v)',
                'int pfull() const', 'void pset(int v)', 'void adc_bin_f(int m)', 'void sbc_bin_f(int m)', 'void adc_f(int m)',
                'void sbc_f(int m)', 'void cmp_f(int r, int m)', 'void bit_f(int m)', 'void adc_bin(int m)', 'void sbc_bin(int m)'):
        assert _fn_body(dev, sig) == _fn_body(host, sig), sig
    d = str(tmp_path)
    src = os.path.join(ROOT, 'tests', 'tools', 'cart_host')
    open(os.path.join(d, 'cart_native.gen.hpp'), 'w').write('// no cartridge needed for the ALU check\n')
    subprocess.check_call(['gcc', '-O1', '-std=c11', '-ffp-contract=off', '-c', os.path.join(src, 'shim.c'), '-o',
                           os.path.join(d, 'shim.o')])
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-I', d, '-c', os.path.join(src, 'main.cpp'), '-o',
                           os.path.join(d, 'main.o')])
    subprocess.check_call(['g++', os.path.join(d, 'shim.o'), os.path.join(d, 'main.o'), '-lm', '-o',
                           os.path.join(d, 'cart_host')])
    p = subprocess.run([os.path.join(d, 'cart_host'), '--alu'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and 'identical to the oracle' in p.stdout, p.stdout[-1500:]

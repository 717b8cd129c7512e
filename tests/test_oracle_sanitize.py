"""The CPU oracle under AddressSanitizer + UBSan (SURVEY §5 sanitizer row): emulator, wrapper chain,
frame pipeline and the scans run for a few hundred steps on both cartridges with no report."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('game,gid', [('pong', 1), ('breakout', 2)])
def test_oracle_runs_clean_under_asan_ubsan(game, gid):
    rom = os.path.join(ROOT, 'roms', game + '.bin')
    if not os.path.exists(rom):
        pytest.skip('cartridge not provisioned')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'])
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1')
    p = subprocess.run([os.path.join(ROOT, 'oracle', 'oracle_asan'), rom, str(gid), '150'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert 'scans rc 0' in p.stdout and 'runtime error' not in p.stderr and 'AddressSanitizer' not in p.stderr

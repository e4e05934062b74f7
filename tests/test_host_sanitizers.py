"""ASAN + UBSAN build of the engine's pure-host translation-unit pieces (host-side sanitizer build).

csrc/tsm_host_util.h holds everything of tsm_engine.hip that needs no HIP type -- BatchNorm folding and weight
packing, the bf16 / split-bf16 converters, the segment rule and the TSM_TUNE_CACHE line parser; tests/host_sanitize.cpp
fuzzes the parser with malformed lines and checks the packers' invariants.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='g++ not available')
def test_host_pieces_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'host_sanitize')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
           '-fno-omit-frame-pointer', '-Wall', '-Wextra', '-Werror', os.path.join(ROOT, 'tests', 'host_sanitize.cpp'),
           '-o', exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    env.pop('LD_PRELOAD', None)
    run = subprocess.run([exe, '20000'], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert 'host sanitize ok' in run.stdout

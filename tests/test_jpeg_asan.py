"""The host JPEG entropy decoder (csrc_host/jpeg_coef.cpp: the one piece of this build that parses untrusted bytes) under
AddressSanitizer on the CPU: the committed fixtures corrupted, truncated and spliced with stray markers a few thousand times must
never read or write outside its buffers. Runs in a subprocess (the sanitizer runtime has to be loaded before Python's allocator)."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entropy_decoder_fuzz_under_address_sanitizer(tmp_path):
    gxx = shutil.which('g++')
    asan = subprocess.run([gxx, '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip() if gxx else ''
    if not gxx or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('no g++ / libasan on this box')
    lib = str(tmp_path / 'libjc_asan.so')
    subprocess.check_call([gxx, '-O1', '-g', '-fPIC', '-shared', '-std=c++17', '-fsanitize=address', '-o', lib,
                           os.path.join(ROOT, 'witw_amd', 'csrc_host', 'jpeg_coef.cpp')])
    script = textwrap.dedent("""
        import ctypes, glob, numpy as np
        lib = ctypes.CDLL(%r)
        g = np.random.default_rng(3)
        n = 0
        for f in sorted(glob.glob(%r)):
            data = np.fromfile(f, dtype=np.uint8)
            for trial in range(120):
                b = data.copy()
                if trial %% 4 == 0:
                    for _ in range(6):
                        b[g.integers(2, len(b))] = g.integers(0, 256)
                elif trial %% 4 == 1:
                    b = b[:g.integers(4, len(b))]
                elif trial %% 4 == 2:
                    i = g.integers(2, len(b) - 2)
                    b[i], b[i + 1] = 0xFF, g.choice([0xD0, 0xD9, 0xC4, 0xDB, 0xDA, 0xDD])
                b = np.ascontiguousarray(b)
                info = np.zeros(22, np.int32)
                rc = lib.witw_jpeg_info(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(b.size), ctypes.c_void_p(info.ctypes.data))
                if rc == 0 and 0 < info[5] < 10 ** 6:
                    coef = np.zeros((info[5], 64), np.int16)
                    qt = np.zeros((info[2], 64), np.uint16)
                    lib.witw_jpeg_decode_coef(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(b.size), ctypes.c_void_p(coef.ctypes.data),
                                              ctypes.c_void_p(qt.ctypes.data))
                n += 1
        # crafted tables: code lengths that over-subscribe the code space (round-3 advisor finding: 255 codes of length 1 passed
        # the symbol-count check and were written ~130 KB past the look-up tables), alone and in front of a real file's segments
        real = np.fromfile(sorted(glob.glob(%r))[0], dtype=np.uint8).tobytes()
        m = 0
        for pos in range(16):
            for cnt in (255, 200, 129, 3):
                for tc in (0x00, 0x10, 0x03, 0x13):
                    counts = [0] * 16
                    counts[pos] = cnt
                    body = bytes([tc]) + bytes(counts) + bytes(i & 255 for i in range(cnt))
                    seg = b'\\xff\\xc4' + (len(body) + 2).to_bytes(2, 'big') + body
                    for blob in (b'\\xff\\xd8' + seg, b'\\xff\\xd8' + seg + b'\\xff\\xd9', real[:2] + seg + real[2:]):
                        b = np.frombuffer(blob, dtype=np.uint8).copy()
                        info = np.zeros(22, np.int32)
                        rc = lib.witw_jpeg_info(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(b.size), ctypes.c_void_p(info.ctypes.data))
                        if rc == 0 and 0 < info[5] < 10 ** 6:
                            coef = np.zeros((info[5], 64), np.int16)
                            qt = np.zeros((info[2], 64), np.uint16)
                            lib.witw_jpeg_decode_coef(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(b.size), ctypes.c_void_p(coef.ctypes.data),
                                                      ctypes.c_void_p(qt.ctypes.data))
                        m += 1
        print('fuzzed', n, 'crafted', m)
        """) % (lib, os.path.join(ROOT, 'tests', 'golden', 'jpeg', '*.jpg'), os.path.join(ROOT, 'tests', 'golden', 'jpeg', 's4*.jpg'))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1')
    p = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'fuzzed 1440 crafted 768' in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])

"""Host-side AddressSanitizer + UBSan run of libffk's host logic (SURVEY section 5
"race detection / sanitizers").  Needs build/libffk_asan.so (tools/build_asan.sh, ~4 min) -- skipped
when it has not been built.  Everything here runs WITHOUT a GPU: argument validation of every entry
point, the workspace-size queries, and ffk_selftest_host (arena growth, block-pool reuse and
eviction, every workspace layout sliced and written end to end with the allocator stubbed to the C
heap).  A sanitizer finding aborts the child process with a report on stderr."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, 'build', 'libffk_asan.so')
RUNTIME = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))



def _stale():
    """the sanitizer build is older than a source it was made from (a new entry point would be missing from it)"""
    built = os.path.getmtime(LIB)
    sources = glob.glob(os.path.join(ROOT, 'filter_functions_amd', 'csrc', '*.h*')) + \
        glob.glob(os.path.join(ROOT, 'filter_functions_amd', 'csrc', '*.inc')) + \
        glob.glob(os.path.join(ROOT, 'include', '*.h'))
    return any(os.path.getmtime(f) > built for f in sources)


pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and RUNTIME) or _stale(),
                                reason='sanitizer variant not built or older than the sources (tools/build_asan.sh)')

CHILD = r'''
import ctypes, itertools, os, random, sys
sys.path.insert(0, os.environ['FFK_ROOT'])
from filter_functions_amd import _lib
lib = _lib.load()
assert lib.ffk_version() == 100
rng = random.Random(7)
# 1. workspace queries over the whole argument range, incl. invalid shapes (must return 0, not crash)
n = 0
for name, (res, args) in _lib.SIGNATURES.items():
    if not name.endswith('_workspace_bytes'):
        continue
    fn = getattr(lib, name)
    for _ in range(300):
        vals = [rng.choice([-1, 0, 1, 2, 3, 4, 7, 8, 16, 17, 64, 257, 1000, 4096]) for _ in args]
        fn(*vals)
        n += 1
print('workspace queries:', n)
# 2. every entry point with NULL / zero arguments: FFK_EINVAL or FFK_EHIP, never a crash
calls = bad = 0
for name, (res, args) in _lib.SIGNATURES.items():
    if res is not ctypes.c_int or name in ('ffk_version',):
        continue
    if name in ('ffk_device_synchronize', 'ffk_release_arena', 'ffk_resident_release_pools'):
        continue
    fn = getattr(lib, name)
    zero = []
    for a in args:
        if a in (ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int64):
            zero.append(0)
        elif a in (ctypes.c_double, ctypes.c_float):
            zero.append(0.0)
        else:
            zero.append(None)
    rc = fn(*zero)
    calls += 1
    if rc == 0 and name not in ('ffk_free', 'ffk_ipc_close_handle', 'ffk_graph_destroy', 'ffk_resident_destroy',
                                'ffk_set_accumulate_events', 'ffk_set_accumulate_gate', 'ffk_set_segment_chunks',
                                'ffk_set_accumulate_variant', 'ffk_stream_destroy', 'ffk_event_destroy',
                                'ffk_memset', 'ffk_memcpy_h2d', 'ffk_memcpy_d2h', 'ffk_memcpy_d2d',
                                'ffk_get_device', 'ffk_set_device', 'ffk_stream_synchronize'):
        bad += 1
        print('accepted an all-zero call:', name)
    lib.ffk_last_error()
print('entry points called with null arguments:', calls, 'unexpectedly accepted:', bad)
# 3. the host-logic self test
lib.ffk_selftest_host.restype = ctypes.c_int
lib.ffk_selftest_host.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int]
report = ctypes.create_string_buffer(256)
rc = lib.ffk_selftest_host(int(os.environ.get('FFK_SELFTEST_ROUNDS', '150')), 12345, report, 256)
print('selftest rc:', rc, report.value.decode())
sys.exit(0 if rc == 0 and bad == 0 else 3)
'''


def test_host_logic_under_asan_and_ubsan(tmp_path):
    env = dict(os.environ, FFK_LIBRARY=LIB, FFK_ROOT=ROOT, LD_PRELOAD=RUNTIME[-1],
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=86',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1:exitcode=87')
    res = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=900)
    out = res.stdout + res.stderr
    with open(os.path.join(ROOT, 'gpurun_out', 'asan_report.txt') if os.path.isdir(os.path.join(ROOT, 'gpurun_out'))
              else tmp_path / 'asan_report.txt', 'w') as fh:
        fh.write(out)
    assert 'AddressSanitizer' not in out and 'runtime error:' not in out, out[-4000:]
    assert res.returncode == 0, out[-4000:]
    assert 'selftest rc: 0' in res.stdout


def test_sanitizer_is_armed():
    """Negative control: ffk_selftest_host(-1, ...) writes one byte past a 16-byte heap block; the
    run must die with an AddressSanitizer report (else the clean run above proves nothing)."""
    child = ("import ctypes, os, sys\nsys.path.insert(0, os.environ['FFK_ROOT'])\n"
             "from filter_functions_amd import _lib\nlib = _lib.load()\n"
             "lib.ffk_selftest_host.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int]\n"
             "lib.ffk_selftest_host(-1, 0, None, 0)\nprint('survived')\n")
    env = dict(os.environ, FFK_LIBRARY=LIB, FFK_ROOT=ROOT, LD_PRELOAD=RUNTIME[-1],
               ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:exitcode=86')
    res = subprocess.run([sys.executable, '-c', child], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and 'heap-buffer-overflow' in res.stderr and 'survived' not in res.stdout

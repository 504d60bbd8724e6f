"""Config 3's rule kernel (from_atomic_block_kernel<1,4>) alone: HIP events recorded by the library around its launch
(ffk_set_accumulate_events), over 20 device calls.    python tools/time_rule_kernel.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402
from filter_functions_amd import _lib, numeric  # noqa: E402
from filter_functions_amd import pulse_sequence as ps  # noqa: E402
from filter_functions_amd._resident import ResidentResult  # noqa: E402

cfg = wl.CONFIG3
omega = wl.rb_omega(cfg['W'], cfg['T'])
_, cliffords = wl.rb_cliffords(ff, omega, cfg['T'])
seq = [cliffords[k] for k in wl.rb_draw(cfg['n_gates'], cfg['seed'])]
_, distinct, _, index = ps._validated_sequence(seq)
residents, taus = [p._resident for p in distinct], [p.tau for p in distinct]
lib = _lib.load()
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
for e in (e0, e1):
    _lib.check(lib.ffk_event_create(ctypes.byref(e)))
ms, out = ctypes.c_float(), []
for rep in range(25):
    keep = ResidentResult()
    _lib.check(lib.ffk_set_accumulate_events(e0, e1))
    numeric.concatenate_sequence_resident(residents, taus, index, distinct[0].basis, which='total',
                                          return_filter_function=True, keep=keep)
    _lib.check(lib.ffk_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    if rep >= 5:
        out.append(ms.value*1e3)
_lib.check(lib.ffk_set_accumulate_events(None, None))
print(f'rule kernel: median {np.median(out):.1f} us, min {min(out):.1f} us, max {max(out):.1f} us over {len(out)} calls')

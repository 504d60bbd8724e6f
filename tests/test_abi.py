"""CPU-only: the C-ABI library loads and exports every symbol include/ffk.h declares, the
Python binding table matches the header, and the product fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'ffk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ffk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from filter_functions_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'libffk.so not built (run __graft_entry__.build())'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) > 30
    for name in names:
        assert hasattr(lib, name), f'{name} declared in ffk.h but not exported'


def test_binding_table_matches_header():
    from filter_functions_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    assert lib.ffk_version() == 100


def test_workspace_queries_need_no_gpu():
    from filter_functions_amd import _lib
    lib = _lib.load()
    assert lib.ffk_diagonalize_workspace_bytes(256, 4) > 256*16*16
    assert lib.ffk_control_matrix_workspace_bytes(4096, 16, 3, 256, 4) > 3*16*4096*16
    assert lib.ffk_control_matrix_workspace_bytes(64, 289, 3, 8, 17) > 3*289*64*16   # runtime-d kernels
    assert lib.ffk_control_matrix_workspace_bytes(4096, 16, 3, 256, 65) == 0     # above FFK_MAX_D
    assert lib.ffk_diagonalize_workspace_bytes(8, 64) > 0 and lib.ffk_diagonalize_workspace_bytes(8, 65) == 0
    assert lib.ffk_pipeline_workspace_bytes(64, 289, 3, 8, 17, 3, 1) == 0        # per-dimension kernels only
    assert lib.ffk_liouville_workspace_bytes(1, 4, 16) > 0
    assert lib.ffk_infidelity_workspace_bytes(4096, 3, 3) > 0
    assert lib.ffk_pipeline_workspace_bytes(4096, 16, 3, 256, 4, 3, 1) > 0


def test_no_cpu_fallback():
    """Without a GPU every numeric entry point must raise, not silently compute on the host."""
    import numpy as np
    from filter_functions_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('GPU present')
    import filter_functions_amd as ff
    X, Z = ff.util.paulis[1], ff.util.paulis[3]
    pulse = ff.PulseSequence([[X/2, [1.0]]], [[Z/2, [1.0]]], [1.0])
    with pytest.raises(_lib.FFKError):
        pulse.get_filter_function(np.linspace(0.1, 1, 5))
    with pytest.raises(_lib.FFKError):
        ff.liouville_representation(np.eye(2), ff.Basis.pauli(1))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'filter_functions_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'ff_oracle' not in src and 'oracle/' not in src, f


@pytest.mark.parametrize('generator, committed', [('gen_pq_consumer.py', 'ctrl_pq_consumer.inc'),
                                                  ('gen_pcr_consumer.py', 'ctrl_pcr_consumer.inc')])
def test_generated_consumers_are_current(generator, committed):
    """The d = 4 and d = 8 accumulate kernels' consumer loops (filter_functions_amd/csrc/ctrl_p*_consumer.inc, inline
    assembly) are generated: each committed file must be what its generator under tools/ prints."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith('GEN_PCR_')}
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', generator)], capture_output=True,
                         text=True, check=True, env=env).stdout
    with open(os.path.join(root, 'filter_functions_amd', 'csrc', committed)) as fh:
        assert fh.read() == out

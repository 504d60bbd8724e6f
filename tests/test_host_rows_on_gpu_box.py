"""SURVEY section 8 rows a15 (basis ordering / index maps) and a16 (`get_sample_frequencies`) are host
logic with a bit-exact bar; their tests live in test_host_logic.py and run in the CPU suite.  The same
functions are run once more here under the `gpu` marker so that they are checked against the GPU box's
own NumPy / BLAS as well -- they need no GPU and take under a second."""
import pytest

import test_host_logic as host


@pytest.mark.gpu
def test_basis_bit_exact_against_reference_on_the_gpu_box():
    host.test_basis_bit_exact_against_reference()


@pytest.mark.gpu
def test_basis_index_maps_bit_exact_on_the_gpu_box():
    host.test_basis_index_maps_bit_exact()


@pytest.mark.gpu
def test_util_against_reference_vectors_on_the_gpu_box():
    host.test_util_against_reference_vectors()

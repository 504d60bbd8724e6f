"""CPU-only tests of the host-side mirror: basis construction (bit-exact against the reference's
arrays), argument validation, cache bookkeeping that needs no kernel, host math harness."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import filter_functions_amd as ff
from conftest import ROOT, load_golden, rel_err
from filter_functions_amd import util
from filter_functions_amd.basis import (Basis, equivalent_pauli_basis_elements, expand, ggm_expand,
                                        remap_pauli_basis_elements)


def test_basis_bit_exact_against_reference():
    g = load_golden('basis')
    for n in (1, 2, 3):
        b = Basis.pauli(n)
        assert np.array_equal(np.asarray(b), g[f'pauli{n}'])
        assert b.labels == list(g[f'pauli{n}_labels'])
        assert b.btype == 'Pauli' and b.d == 2**n
    for d in range(2, 9):
        b = Basis.ggm(d)
        assert np.array_equal(np.asarray(b), g[f'ggm{d}'])
        assert b.labels == list(g[f'ggm{d}_labels'])
    sha = hashlib.sha256(np.ascontiguousarray(np.asarray(Basis.pauli(4)) + 0.0).tobytes()).hexdigest()
    assert sha == str(g['pauli4_sha256'])
    sha = hashlib.sha256(np.ascontiguousarray(np.asarray(Basis.ggm(16)) + 0.0).tobytes()).hexdigest()
    assert sha == str(g['ggm16_sha256'])


def test_basis_index_maps_bit_exact():
    g = load_golden('basis')
    for N in (1, 2, 3, 4):
        for q in range(N):
            assert np.array_equal(equivalent_pauli_basis_elements(q, N), g[f'equiv_N{N}_q{q}'])
        if N > 1:
            assert np.array_equal(equivalent_pauli_basis_elements([0, 1], N), g[f'equiv_N{N}_q01'])
            assert np.array_equal(remap_pauli_basis_elements(list(range(N))[::-1], N),
                                  g[f'remap_N{N}_rev'])
    assert np.array_equal(remap_pauli_basis_elements([1, 2, 0], 3), g['remap_N3_120'])


def test_basis_properties_and_expand():
    g = load_golden('basis')
    for b in (Basis.pauli(2), Basis.ggm(3), Basis.ggm(4)):
        assert b.isherm and b.isnorm and b.isorthogonal and b.isorthonorm
        assert b.istraceless and b.iscomplete
        assert b == b and not (b != b)
        assert b.H == b
    assert rel_err(Basis.ggm(5).expand(g['expand_M']), g['expand_ggm5']) < 1e-15
    Mh = g['expand_M'] + g['expand_M'].conj().transpose(0, 2, 1)
    got = Basis.ggm(5).expand(Mh, hermitian=True)
    assert got.dtype == np.float64 and rel_err(got, g['expand_ggm5_herm']) < 1e-15
    assert rel_err(Basis.pauli(2).expand(g['expand_M4']), g['expand_pauli2']) < 1e-15
    assert rel_err(expand(g['expand_M'], Basis.ggm(5)), ggm_expand(g['expand_M'])) < 1e-15
    nonherm = Basis(np.array([[[0, 1], [0, 0]], [[0, 0], [1, 0]]]))
    assert not nonherm.isherm and not nonherm.iscomplete and nonherm.btype == 'Custom'


def test_util_against_reference_vectors():
    g = load_golden('util')
    assert np.array_equal(util.integrate(g['f'], g['x']), g['integral'])
    assert np.array_equal(util.cexp(g['cexp_in']), g['cexp_out'])
    assert np.array_equal(util.cexpm1(g['cexp_in'] - 5e3), g['cexpm1_out'])
    h = load_golden('hadamard')
    X, Y, Z = util.paulis[1:]
    p = ff.PulseSequence([[X/2, [0, np.pi], 'X'], [Y/2, [np.pi/2, 0], 'Y']], [[Z/2, [1, 1], 'Z']],
                         [1, 1])
    assert np.array_equal(util.get_sample_frequencies(p, 200), h['omega'])
    assert np.array_equal(util.get_sample_frequencies(p), h['omega_default'])
    assert np.array_equal(np.asarray(p.basis), h['basis'])
    assert np.array_equal(p.c_opers, h['c_opers']) and np.array_equal(p.c_coeffs, h['c_coeffs'])
    assert list(p.c_oper_identifiers) == list(h['c_oper_identifiers'])


def test_pulse_sequence_constructor_contract():
    """Error types of the constructor, reference tests/test_core.py:42-221."""
    X, Z = util.paulis[1], util.paulis[3]
    with pytest.raises(TypeError):
        ff.PulseSequence([[X, [1]]], [[Z, [1]]], 1.0)                  # dt not a sequence
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1]]], [[Z, [1]]], [-1.0])               # negative dt
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1j])                 # complex dt
    with pytest.raises(TypeError):
        ff.PulseSequence(1, [[Z, [1]]], [1.0])
    with pytest.raises(TypeError):
        ff.PulseSequence([[X, 1]], [[Z, [1]]], [1.0])                  # coeffs not a sequence
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1, 2]]], [[Z, [1]]], [1.0])             # wrong coeff length
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1], 'a'], [Z, [1], 'a']], [[Z, [1]]], [1.0])   # duplicate ids
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1]]], [[np.eye(3), [1]]], [1.0])        # dimension mismatch
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1.0], basis=np.eye(2))
    with pytest.raises(ValueError):
        ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1.0], basis=Basis.pauli(2))
    p = ff.PulseSequence([[X, [1, 2], 'b'], [Z, [3, 4], 'a']], [[Z, [1, 1]]], [1.0, 2.0])
    assert list(p.c_oper_identifiers) == ['a', 'b']            # sorted by identifier
    assert np.array_equal(p.c_coeffs, [[3, 4], [1, 2]])
    assert list(p.n_oper_identifiers) == ['B_0']
    assert p.d == 2 and p.basis.btype == 'GGM' and len(p) == 2
    assert np.array_equal(p.t, [0, 1, 3]) and p.tau == 3 and p.duration == 3
    q = p[:1]
    assert len(q) == 1 and q.tau == 1
    with pytest.raises(IndexError):
        p[5:]
    with pytest.raises(ValueError):
        p.cleanup('nonsense')
    with pytest.raises(ValueError):
        p.get_filter_function([1.0], which='bogus')
    with pytest.raises(util.CalculationError):
        p.get_pulse_correlation_filter_function()
    with pytest.raises(util.CalculationError):
        p.get_pulse_correlation_control_matrix()
    c = p.copy()
    assert c._data is not p._data and c.dt is p.dt


def test_cache_bookkeeping_without_kernels():
    """omega setter / cleanup / is_cached semantics (reference pulse_sequence.py:1158-1245)."""
    X, Z = util.paulis[1], util.paulis[3]
    p = ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1.0])
    omega = [1.0, 2.0]
    R = np.zeros((1, 4, 2), complex)
    p._data['total_propagator'] = np.eye(2, dtype=complex)
    p._data['total_propagator_liouville'] = np.eye(4)
    p.cache_control_matrix(omega, R)
    assert p.is_cached('control matrix') and p.is_cached('total_phases') and p.is_cached('omega')
    assert p.get_control_matrix(np.array(omega)) is R            # list vs array omega: same cache
    assert np.allclose(p.get_total_phases(omega), np.exp(1j*np.array(omega)))
    F = np.ones((1, 1, 2), complex)
    p.cache_filter_function(omega, filter_function=F)
    assert p.get_filter_function(omega) is F
    p.omega = [1.0, 3.0]                                          # different grid: wiped
    assert not p.is_cached('control_matrix') and not p.is_cached('filter function')
    assert p.is_cached('total propagator liouville')
    p.cleanup('all')
    assert not p.data and not p.frequency_data
    assert p.nbytes == 0


def test_remembered_grid_is_a_private_copy():
    """The reference copies omega into the cache (pulse_sequence.py:1166).  Only grids this package
    froze itself are shared between pulses; a caller's read-only VIEW of a writable array is copied, so
    changing the base afterwards cannot change the remembered grid under the cached results."""
    X, Z = util.paulis[1], util.paulis[3]
    p = ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1.0])
    base = np.array([1.0, 2.0, 3.0])
    view = base[:]
    view.flags.writeable = False
    p.omega = view
    assert p.omega is not view and not p.omega.flags.writeable
    base[1] = 7.0
    assert np.array_equal(p.omega, [1.0, 2.0, 3.0])
    q = ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1.0])
    q.omega = p.omega                       # the package's own frozen copy: shared as it is
    assert q.omega is p.omega
    r = ff.PulseSequence([[X, [1]]], [[Z, [1]]], [1.0])
    r.omega = [1.0, 2.0, 3.0]               # same content from elsewhere: the one interned object
    assert r.omega is p.omega


def test_spectrum_validation():
    with pytest.raises(ValueError):
        util.parse_spectrum(np.ones((3, 5)), np.arange(5), [0, 1])           # wrong n_idx
    with pytest.raises(ValueError):
        util.parse_spectrum(np.ones((2, 2, 5)) + 1j*np.arange(5), np.arange(5), [0, 1])
    with pytest.raises(ValueError):
        util.parse_spectrum(np.ones((2, 2, 2, 5)), np.arange(5), [0, 1])
    assert util.parse_spectrum(np.ones(5), np.arange(5), [0, 1]).shape == (5,)
    with pytest.raises(ValueError):
        util.get_indices_from_identifiers(['a', 'b'], ['c'])
    assert list(util.get_indices_from_identifiers(['a', 'b'], 'b')) == [1]


def test_device_math_on_host():
    """The __host__ __device__ numerics of csrc/ffk_math.h, compiled for the host by
    tests/csrc (built by __graft_entry__.build()), against NumPy."""
    path = os.path.join(ROOT, 'tests', 'csrc', 'libffk_math_host.so')
    if not os.path.exists(path):
        pytest.skip('host math harness not built')
    lib = ctypes.CDLL(path)
    dp = ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(0)
    for scale in (1.0, 1e3, 1e6, 1e9):
        x = (rng.random(200000)*2 - 1)*scale
        s, c = np.empty_like(x), np.empty_like(x)
        lib.ffk_host_sincos(ctypes.c_long(x.size), x.ctypes.data_as(dp), s.ctypes.data_as(dp),
                            c.ctypes.data_as(dp))
        assert np.max(np.abs(s - np.sin(x))/np.spacing(np.abs(np.sin(x)))) <= 2
        assert np.max(np.abs(c - np.cos(x))/np.spacing(np.abs(np.cos(x)))) <= 2
    # first-order integral incl. the exact-zero mask, omega = 0, 1e-10 and resonance
    omega = np.concatenate([[0.0, 1e-10, -1e-10, 2.5, -2.5], rng.standard_normal(1000)*10])
    dE = np.concatenate([[0.0, 0.0, 0.0, -2.5, 2.5], rng.standard_normal(1000)])
    dt = 0.37
    out = np.empty(2*omega.size)
    lib.ffk_host_first_order_integral(ctypes.c_long(omega.size), omega.ctypes.data_as(dp),
                                      dE.ctypes.data_as(dp), ctypes.c_double(dt),
                                      out.ctypes.data_as(dp))
    got = out[0::2] + 1j*out[1::2]
    x = omega + dE
    ref = np.full(x.shape, dt, dtype=complex)
    m = x != 0
    ref[m] = util.cexpm1(x[m]*dt)/(1j*x[m])
    assert np.max(np.abs(got - ref)) < 5e-16
    assert got[0] == dt and got[3] == dt and got[4] == dt
    # angle-addition form used by the accumulate kernel, including the near-resonance band where
    # it switches to the direct small-angle evaluation (|h| < 2^-5) and the band's edge
    om2 = np.concatenate([omega, 10.0**rng.uniform(-2, 3, 4000)])
    gap = np.concatenate([[0.01, 0.03125/0.185, 0.0313/0.185, 1e-3, 1e-6, 1e-9, -1e-4],
                          rng.standard_normal(2000)*0.2])
    dE2 = np.concatenate([dE, -om2[dE.size:dE.size + gap.size] + gap,
                          rng.standard_normal(om2.size - dE.size - gap.size)*3])
    out2 = np.empty(2*om2.size)
    lib.ffk_host_first_order_integral_aa(ctypes.c_long(om2.size), om2.ctypes.data_as(dp),
                                         dE2.ctypes.data_as(dp), ctypes.c_double(dt),
                                         out2.ctypes.data_as(dp))
    got2 = out2[0::2] + 1j*out2[1::2]
    x2 = om2 + dE2
    ref2 = np.full(x2.shape, dt, dtype=complex)
    m2 = x2 != 0
    ref2[m2] = util.cexpm1(x2[m2]*dt)/(1j*x2[m2])
    # the half-angle a + b differs from the reference's fl(fl(w + dE) dt)/2 by a few ulp of
    # (|w| + |dE|) dt; inside the band the evaluation is direct (relative error of a few ulp)
    eps = np.finfo(float).eps
    tol = 64*eps*np.abs(ref2) + 8*eps*(np.abs(om2) + np.abs(dE2))*dt*dt
    assert np.all(np.abs(got2 - ref2) <= tol)
    band = np.abs(0.5*x2*dt) < 0.03
    assert band.sum() > 1000 and np.all(np.abs(got2 - ref2)[band] <= 8*eps*np.abs(ref2[band]))


def test_concatenation_bookkeeping():
    """Identifier / coefficient-table logic of concatenate_without_filter_function; expected values
    are what the reference (pulse_sequence.py:1340-1483) produces on the same inputs."""
    X, Y, Z = util.paulis[1:]
    a = ff.PulseSequence([[X, [1., 2.], 'c']], [[Z, [1., 1.], 'n']], [1., 1.])
    b = ff.PulseSequence([[Y, [3.], 'c']], [[Z, [1.], 'n']], [2.])
    ab = ff.concatenate_without_filter_function([a, b])
    assert list(ab.c_oper_identifiers) == ['c_0', 'c_1']        # same name, different operator
    assert np.array_equal(ab.c_opers, [X, Y])
    assert np.array_equal(ab.c_coeffs, [[1, 2, 0], [0, 0, 3]])  # control terms zero-filled
    assert list(ab.n_oper_identifiers) == ['n'] and np.array_equal(ab.n_coeffs, [[1, 1, 1]])
    assert np.array_equal(ab.dt, [1, 1, 2]) and ab.tau == 4 and len(ab) == 3
    # a noise operator missing from one pulse inherits its (constant) sensitivity
    c = ff.PulseSequence([[X, [1.], 'c']], [[Z, [2.], 'nz'], [X, [0.5], 'nx']], [1.])
    d = ff.PulseSequence([[X, [3.], 'c']], [[Z, [2.], 'nz']], [2.])
    cd, c_map, n_map = ff.concatenate_without_filter_function([c, d], return_identifier_mappings=True)
    assert list(cd.n_oper_identifiers) == ['nx', 'nz']
    assert np.array_equal(cd.n_coeffs, [[0.5, 0.5], [2, 2]])
    assert n_map == {0: {'nx': 'nx', 'nz': 'nz'}, 1: {'nz': 'nz'}} and c_map[0] == {'c': 'c'}
    # ... but not a time-dependent one
    e = ff.PulseSequence([[X, [1., 1.], 'c']], [[Z, [2., 2.], 'nz'], [X, [0.5, 0.6], 'nx']], [1., 1.])
    with pytest.raises(ValueError):
        ff.concatenate_without_filter_function([e, d])
    # equal operators under different names are refused
    with pytest.raises(ValueError):
        ff.concatenate_without_filter_function([a, ff.PulseSequence([[X, [1.], 'other']],
                                                                    [[Z, [1.], 'n']], [1.])])
    with pytest.raises(TypeError):
        ff.concatenate_without_filter_function(5)
    with pytest.raises(TypeError):
        ff.concatenate_without_filter_function([a, 'b'])
    with pytest.raises(ValueError):
        ff.concatenate_without_filter_function([a, ff.PulseSequence([[np.eye(3), [1.]]],
                                                                    [[np.eye(3), [1.]]], [1.])])
    with pytest.raises(ValueError):
        ff.concatenate_without_filter_function(
            [a, ff.PulseSequence([[X, [1.]]], [[Z, [1.]]], [1.], basis=ff.Basis.pauli(1)[::-1])])
    # no filter function work unless something is cached or asked for
    assert not ff.concatenate([a, b]).is_cached('filter_function')
    assert not ff.concatenate([a, b], calc_filter_function=False).is_cached('omega')
    with pytest.raises(ValueError):          # forces the filter function, but no frequencies known
        ff.concatenate([a, b], calc_second_order_FF=True)


def test_long_sequence_bookkeeping_equals_pairwise_concatenation():
    """A long sequence drawn from a few pulses takes vectorised routes (one scatter for the
    coefficient blocks, np.take gathers, shared ragged columns; the bookkeeping of ONE entry when all
    pulses carry the same operator table): the result must be the pulse that concatenating the
    positions one after the other gives -- ragged and equal segment counts, pulses with and without
    all control operators, operators in different order, a noise operator one pulse lacks."""
    X, Y, Z = util.paulis[1:]
    rng = np.random.default_rng(11)

    def chain(seq):
        out = seq[0]
        for p in seq[1:]:
            out = ff.concatenate_without_filter_function([out, p])
        return out

    def same(a, b):
        assert list(a.c_oper_identifiers) == list(b.c_oper_identifiers)
        assert list(a.n_oper_identifiers) == list(b.n_oper_identifiers)
        assert np.array_equal(a.c_opers, b.c_opers) and np.array_equal(a.n_opers, b.n_opers)
        assert np.array_equal(a.c_coeffs, b.c_coeffs) and np.array_equal(a.n_coeffs, b.n_coeffs)
        assert np.array_equal(a.dt, b.dt) and a.tau == b.tau

    def pulse(n_dt, controls, noises):
        return ff.PulseSequence([[op, rng.random(n_dt), name] for op, name in controls],
                                [[op, [s]*n_dt, name] for op, name, s in noises], rng.random(n_dt) + 0.1)
    families = {
        # same operator table everywhere, ragged lengths (the fast path)
        'same table, ragged': [pulse(n, [(X, 'X'), (Y, 'Y')], [(Z, 'Z', 1.0)]) for n in (1, 3, 2, 4)],
        # same table, equal lengths
        'same table, equal': [pulse(2, [(X, 'X'), (Y, 'Y')], [(Z, 'Z', 1.0), (X, 'nX', 0.5)]) for _ in range(3)],
        # control operators missing from some pulses, other order in others (general path, zero fill)
        'mixed controls': [pulse(2, [(X, 'X')], [(Z, 'Z', 1.0)]), pulse(3, [(Y, 'Y'), (X, 'X')], [(Z, 'Z', 1.0)]),
                           pulse(1, [(Y, 'Y')], [(Z, 'Z', 1.0)])],
        # a noise operator only some pulses know (constant sensitivity: inferred)
        'missing noise': [pulse(2, [(X, 'X')], [(Z, 'Z', 1.0), (Y, 'nY', 0.25)]), pulse(3, [(X, 'X')], [(Z, 'Z', 1.0)])],
    }
    for name, distinct in families.items():
        for n_positions in (2, 7, 40):              # 40 > 4 x len(distinct): the gather routes
            draw = rng.integers(0, len(distinct), n_positions)
            draw[:len(distinct)] = np.arange(len(distinct))[:n_positions]     # every pulse appears
            seq = [distinct[k] for k in draw]
            same(ff.concatenate_without_filter_function(seq), chain(seq))


def test_pulse_sequence_equality():
    """__eq__ merges constant stretches before comparing (reference pulse_sequence.py:363-440)."""
    X, Y, Z = util.paulis[1:]

    def make(dt, cx, ident='X'):
        return ff.PulseSequence([[X, cx, ident], [Y, [0.5]*len(dt), 'Y']],
                                [[Z, [1.0]*len(dt), 'Z']], dt)
    a, b, c = make([1., 1., 2.], [1., 1., 3.]), make([2., 2.], [1., 3.]), make([2., 2.], [1., 4.])
    assert a == b and a == a and not (a == c) and a != c
    assert not (a == make([2., 2.1], [1., 3.])) and not (a == make([2., 2.], [1., 3.], 'Xx'))
    assert (a == 5) is False
    with pytest.raises(NotImplementedError):
        a @= b
    with pytest.raises(TypeError):
        hash(a)


def test_from_arrays_contract():
    """PulseSequence.from_arrays: success and the ValueErrors the reference's
    tests/test_core.py:1124-1222 expect."""
    rng = np.random.default_rng(5)
    c_opers = rng.standard_normal((3, 4, 4)) + 1j*rng.standard_normal((3, 4, 4))
    c_ids = np.array(['c1', 'c2', 'c3'])
    c_coeffs = rng.random((3, 100))
    n_opers = rng.standard_normal((2, 4, 4)) + 1j*rng.standard_normal((2, 4, 4))
    n_ids = np.array(['n1', 'n2'])
    n_coeffs = rng.random((2, 100))
    dt = np.linspace(0, 10, 100)
    p = ff.PulseSequence.from_arrays(c_opers, c_ids, c_coeffs, n_opers, n_ids, n_coeffs, dt)
    assert np.array_equal(p.c_opers, c_opers) and np.array_equal(p.c_oper_identifiers, c_ids)
    assert np.array_equal(p.c_coeffs, c_coeffs) and np.array_equal(p.n_opers, n_opers)
    assert np.array_equal(p.n_oper_identifiers, n_ids) and np.array_equal(p.n_coeffs, n_coeffs)
    assert np.array_equal(p.dt, dt)
    bad = [
        (c_opers[:, :, :3], c_ids, c_coeffs, n_opers, n_ids, n_coeffs, dt),       # not square
        (c_opers, c_ids, c_coeffs, n_opers[:, :, :3], n_ids, n_coeffs, dt),
        (c_opers, c_ids[:-1], c_coeffs, n_opers, n_ids, n_coeffs, dt),            # lengths
        (c_opers, c_ids, c_coeffs, n_opers, n_ids[:-1], n_coeffs, dt),
        (c_opers, c_ids, c_coeffs, n_opers, n_ids, n_coeffs, dt[:-1]),            # time steps
    ]
    for args in bad:
        with pytest.raises(ValueError):
            ff.PulseSequence.from_arrays(*args)
    with pytest.raises(ValueError):
        ff.PulseSequence.from_arrays(c_opers, c_ids, c_coeffs, n_opers, n_ids, n_coeffs, dt,
                                     basis=ff.Basis.ggm(5))


def test_is_cached_aliases():
    """is_cached accepts the human-readable aliases in any case and with underscores (reference
    tests/test_core.py:349-385, pulse_sequence.py:508-538)."""
    X, Z = util.paulis[1], util.paulis[3]
    A = ff.PulseSequence([[X, [1]]], [[Z, [2]]], [3])
    aliases = {'eigenvalues': 'eigvals', 'eigenvectors': 'eigvecs', 'propagators': 'propagators',
               'total propagator': 'total_propagator',
               'total propagator liouville': 'total_propagator_liouville'}
    frequency_aliases = {
        'frequencies': 'omega', 'total phases': 'total_phases', 'filter function': 'filter_function',
        'fidelity filter function': 'filter_function',
        'generalized filter function': 'filter_function_gen',
        'pulse correlation filter function': 'filter_function_pc',
        'fidelity pulse correlation filter function': 'filter_function_pc',
        'generalized pulse correlation filter function': 'filter_function_pc_gen',
        'control matrix': 'control_matrix', 'pulse correlation control matrix': 'control_matrix_pc'}
    for present in (True, False):
        for alias, attr in {**aliases, **frequency_aliases}.items():
            store = A._data if alias in aliases else A._frequency_data
            if present:
                store[attr] = 'foo'
            else:
                store.pop(attr, None)
            for name in (alias, alias.upper(), alias.replace(' ', '_')):
                assert A.is_cached(name) is present
    pulse = ff.PulseSequence([[X, [1, 2, 3]]], [[Z, [1, 1, 1]]], [0.5, 1.0, 1.5])
    assert 't' not in pulse.data and 'tau' not in pulse.data
    assert np.array_equal(pulse.t, [0, *pulse.dt.cumsum()])
    assert pulse.tau == pulse.t[-1] and pulse.duration == pulse.tau


@pytest.mark.parametrize('name', ['pauli1', 'ggm3', 'pauli2'])
def test_choi_matrix_and_complete_positivity(name):
    """liouville_to_choi / liouville_is_CP / liouville_is_cCP against the reference's outputs
    (superoperator.py:87-266): unitary channels (CP), a cumulant function (cCP, not CP), its
    exponential (CP) and the transposition map (not CP)."""
    from conftest import load_golden, rel_err
    from filter_functions_amd import superoperator
    g = load_golden('superoperator')
    btype = 'Pauli' if name.startswith('pauli') else 'GGM'
    basis = ff.Basis(g[f'{name}_basis'], btype=btype)
    S = g[f'{name}_superoperators']
    assert rel_err(superoperator.liouville_to_choi(S, basis), g[f'{name}_choi']) < 1e-14
    CP, (D, V) = superoperator.liouville_is_CP(S, basis, True)
    assert np.array_equal(CP, g[f'{name}_CP'])
    assert np.abs(D - g[f'{name}_CP_eigvals']).max() < 1e-12
    assert np.array_equal(superoperator.liouville_is_CP(S, basis), CP)
    cCP, (D2, _) = superoperator.liouville_is_cCP(S, basis, True)
    assert np.array_equal(cCP, g[f'{name}_cCP'])
    assert np.abs(D2 - g[f'{name}_cCP_eigvals']).max() < 1e-12
    # the physics: unitaries and exp(K) are CP, K itself only conditionally, transposition neither
    assert CP[:3].all() and not CP[3] and CP[4] and not CP[5]
    assert cCP[3]
    # single matrix in, scalar bool out
    assert bool(superoperator.liouville_is_CP(S[0], basis)) is True


def test_remap_and_extend_bookkeeping():
    """Hamiltonian / identifier bookkeeping of remap and extend against the reference's outputs
    (pulse_sequence.py:1976-2625); no frequencies cached, so nothing touches the device."""
    from conftest import load_golden, rel_err
    g = load_golden('register')
    p = {}
    for name in ('p1', 'p1b', 'p2', 'p3'):
        p[name] = ff.PulseSequence.from_arrays(
            g[f'{name}_c_opers'], g[f'{name}_c_oper_identifiers'], g[f'{name}_c_coeffs'],
            g[f'{name}_n_opers'], g[f'{name}_n_oper_identifiers'], g[f'{name}_n_coeffs'],
            g[f'{name}_dt'], ff.Basis(g[f'{name}_basis'], btype='Pauli'))
    ZZ = ff.util.tensor(ff.util.paulis[3], np.eye(2), ff.util.paulis[3])
    cases = {
        'remap_p2_10': ff.remap(p['p2'], (1, 0)),
        'remap_p3_201': ff.remap(p['p3'], (2, 0, 1)),
        'extend_singles': ff.extend([(p['p1'], 0), (p['p1b'], 2)], N=3),
        'extend_multi': ff.extend([(p['p2'], (2, 0)), (p['p1'], 1)], N=4),
        'extend_additional': ff.extend([(p['p1'], 0), (p['p1b'], 2)], N=3,
                                       additional_noise_Hamiltonian=[[ZZ, np.ones(3), 'ZZ']]),
    }
    for prefix, pulse in cases.items():
        assert list(pulse.c_oper_identifiers) == list(g[f'{prefix}_c_oper_identifiers'])
        assert list(pulse.n_oper_identifiers) == list(g[f'{prefix}_n_oper_identifiers'])
        for attr in ('c_opers', 'n_opers', 'c_coeffs', 'n_coeffs'):
            assert rel_err(getattr(pulse, attr), g[f'{prefix}_{attr}']) < 1e-14, (prefix, attr)
        assert not pulse.is_cached('control_matrix')
    assert cases['extend_multi'].basis.btype == 'Pauli' and cases['extend_multi'].d == 16
    # tensor utilities
    X, Y, Z = ff.util.paulis[1:]
    t = ff.util.tensor(X, Y, Z)
    assert np.array_equal(ff.util.tensor_transpose(t, [1, 2, 0], [[2, 2, 2]]*2), ff.util.tensor(Y, Z, X))
    assert np.array_equal(ff.util.embed_in_register(ff.util.tensor(X, Z), [0, 2], 3),
                          ff.util.tensor(X, np.eye(2), Z))
    assert np.array_equal(ff.util.embed_in_register(np.array([1., 2.]), [1], 2, rank=1),
                          [1., 2., 1., 2.])
    ok, phase = ff.util.oper_equiv(X, np.exp(0.25j)*X)
    assert ok and np.isclose(phase, 0.25)
    assert ff.util.all_array_equal([np.arange(3)]*3) and not ff.util.all_array_equal([[1], [2]])
    with pytest.raises(ValueError):
        ff.util.tensor_transpose(t, [0, 0, 1], [[2, 2, 2]]*2)


def test_basis_from_partial_and_analytic_formulas():
    """Basis.from_partial (reference basis.py:492-620): given elements first (after the identity
    for traceless bases), orthonormal Hermitian completion; the closed-form DD filter functions
    (reference analytic.py:59-88) at a few known values."""
    X, Y, Z = ff.util.paulis[1:]
    b = ff.Basis.from_partial([X, Y], labels=['X', 'Y'])
    assert b.shape == (4, 2, 2) and b.isorthonorm and b.isherm and b.istraceless and b.iscomplete
    assert np.allclose(b[0], np.eye(2)/np.sqrt(2)) and np.allclose(b[1], X/np.sqrt(2))
    assert np.allclose(b[2], Y/np.sqrt(2)) and np.allclose(np.abs(b[3]), np.abs(Z)/np.sqrt(2))
    assert b.btype == 'From partial' and list(b.labels)[:2] == ['X', 'Y'] and len(b.labels) == 4
    ggm = ff.Basis.ggm(4)
    b4 = ff.Basis.from_partial(ggm[[3, 7, 9]])
    assert b4.shape == (16, 4, 4) and b4.isorthonorm and b4.iscomplete and b4.istraceless
    assert np.allclose(np.asarray(b4)[1:4], np.asarray(ggm)[[3, 7, 9]])
    nt = ff.Basis.from_partial([np.diag([1.0, 0.0, 0.0])])          # not traceless
    assert nt.shape == (9, 3, 3) and nt.isorthonorm and not nt.istraceless
    with pytest.raises(ValueError):
        ff.Basis.from_partial([X, X + Y])                            # not orthogonal
    with pytest.raises(ValueError):
        ff.Basis.from_partial([np.diag([1.0, 0.0])], traceless=True)
    with pytest.raises(ValueError):
        ff.Basis.from_partial([X, Y], labels=['only one'])
    z = np.array([np.pi, 2*np.pi])
    assert np.allclose(ff.analytic.FID(z), [2, 0]) and np.allclose(ff.analytic.SE(z), [2, 8])
    assert np.allclose(ff.analytic.CPMG(z, 1), ff.analytic.SE(z))
    assert np.allclose(ff.analytic.UDD(z, 1), ff.analytic.SE(z))
    assert np.allclose(ff.analytic.CDD(z, 1), ff.analytic.PDD(z, 1))
    assert np.isscalar(float(ff.analytic.UDD(1.3, 4)))


def test_hermitian_operand_rows_carry_the_whole_trace():
    """The identity behind the Liouville kernels' d^2 operand rows (csrc/ffk_internal.h::
    hermitian_operand_row, restated here): for Hermitian C_i, C_j and ANY U,
    tr(U^dag C_i U C_j) = sum_a Re CB[a,a] Re C[a,a] + sum_{a<b} 2 (Re CB[a,b] Re C[b,a] - Im CB[a,b] Im C[b,a]),
    with the rows numbered a (diagonal), d + 2 p(a,b) (+1 for the imaginary parts)."""
    def row(a, b, imag, d):
        if a == b:
            return -1 if imag else a
        if a > b:
            return -1
        return d + 2*(a*(2*d - a - 1)//2 + b - a - 1) + imag
    rng = np.random.default_rng(5)
    for d in (2, 3, 5, 8):
        basis = np.asarray(ff.Basis.ggm(d))
        U = rng.standard_normal((d, d)) + 1j*rng.standard_normal((d, d))       # not unitary
        CB = np.einsum('ba,ibc,cd->iad', U.conj(), basis, U)
        rows = sorted(row(a, b, im, d) for a in range(d) for b in range(d) for im in (0, 1)
                      if row(a, b, im, d) >= 0)
        assert rows == list(range(d*d))                                        # a bijection onto 0 .. d^2 - 1
        A = np.zeros((d*d, len(basis)))
        B = np.zeros((d*d, len(basis)))
        for a in range(d):
            for b in range(a, d):
                r0, r1 = row(a, b, 0, d), row(a, b, 1, d)
                A[r0] = CB[:, a, b].real
                B[r0] = basis[:, b, a].real*(1 if a == b else 2)
                if r1 >= 0:
                    A[r1] = -CB[:, a, b].imag
                    B[r1] = 2*basis[:, b, a].imag
        full = np.einsum('iab,jba->ij', CB, basis)
        assert np.abs(full.imag).max() < 1e-12
        assert np.abs(A.T @ B - full.real).max() < 1e-12


def test_tensor_product_chains():
    """util.tensor / tensor_insert / tensor_merge / tensor_transpose on product chains with known
    constituents (after the reference's tests/test_util.py tensor tests and the doctest examples of
    util.py:521-556, :695-726); the chain orders for mixed-sign merge positions are the
    reference's own (generated by running it, see the literal table)."""
    rng = np.random.default_rng(11)
    t = util.tensor
    I, X, Y, Z = util.paulis
    assert np.array_equal(t(X, Z), np.kron(X, Z))
    dims = [[2, 2], [2, 2]]
    arr = t(X, I)
    assert np.array_equal(util.tensor_insert(arr, Y, Z, arr_dims=dims, pos=0), t(Y, Z, X, I))
    assert np.array_equal(util.tensor_insert(arr, Y, Z, arr_dims=dims, pos=1), t(X, Y, Z, I))
    assert np.array_equal(util.tensor_insert(arr, Y, Z, arr_dims=dims, pos=2), t(X, I, Y, Z))
    assert np.array_equal(util.tensor_insert(arr, Y, Z, arr_dims=dims, pos=-1), t(X, Y, Z, I))
    A, B, C = rng.standard_normal((2, 3, 1, 2)), rng.standard_normal((2, 2, 2, 2)), \
        rng.standard_normal((3, 2, 1))
    r = util.tensor_insert(t(A, C, rank=3), B, pos=1, rank=3, arr_dims=[[3, 3], [1, 2], [2, 1]])
    assert r.shape == (2, 18, 4, 4) and np.allclose(r, t(A, B, C, rank=3))
    # random chains of random rank with broadcast leading axes: numpy.insert rule
    for _ in range(60):
        rank = int(rng.integers(1, 4))
        n_arr, n_ins = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        arrs = [rng.standard_normal(tuple(rng.integers(1, 4, size=rank))) for _ in range(n_arr)]
        inss = [rng.standard_normal(tuple(rng.integers(1, 4, size=rank))) for _ in range(n_ins)]
        arrs[0] = rng.standard_normal((2,) + arrs[0].shape)
        arr_dims = [[a.shape[a.ndim - rank + ax] for a in arrs] for ax in range(rank)]
        ins_dims = [[a.shape[a.ndim - rank + ax] for a in inss] for ax in range(rank)]
        pos = sorted(int(p) for p in rng.integers(0, n_arr + 1, size=n_ins))
        chain = list(range(n_arr))
        for shift, (p, j) in enumerate(zip(pos, range(n_ins))):
            chain.insert(p + shift, n_arr + j)
        expected = t(*[(arrs + inss)[k] for k in chain], rank=rank)
        arr, ins = t(*arrs, rank=rank), t(*inss, rank=rank)
        got = util.tensor_merge(arr, ins, pos=pos, arr_dims=arr_dims, ins_dims=ins_dims, rank=rank)
        assert got.shape == expected.shape and np.allclose(got, expected)
        got = util.tensor_insert(arr, *inss, pos=pos, arr_dims=arr_dims, rank=rank)
        assert got.shape == expected.shape and np.allclose(got, expected)
        # and back: moving the inserted factors to the end restores arr (x) ins
        all_dims = [[(a + i)[k] for k in chain] for a, i in zip(arr_dims, ins_dims)]
        back = util.tensor_transpose(got, np.argsort(chain), all_dims, rank=rank)
        assert np.allclose(back, t(arr, ins, rank=rank))
    # chain orders of the reference for (n_arr, pos): constituents 0..n_arr-1 are arr's
    reference_chains = [(2, [0, -1], [0, 2, 3, 1]), (2, [1, -1, -2], [4, 0, 3, 2, 1]),
                        (3, [-1, 0, 1], [0, 4, 1, 5, 3, 2]), (3, [2, -1], [0, 1, 4, 3, 2]),
                        (3, [0, 3, -1], [0, 3, 1, 5, 2, 4]), (2, [2, 0], [3, 0, 1, 2]),
                        (3, [1, 1], [0, 3, 4, 1, 2])]
    for n_arr, pos, chain in reference_chains:
        parts = [np.array([1.0, 2.0 + k]) for k in range(n_arr + len(pos))]
        got = util.tensor_merge(t(*parts[:n_arr], rank=1), t(*parts[n_arr:], rank=1), pos=pos,
                                arr_dims=[[2]*n_arr], ins_dims=[[2]*len(pos)], rank=1)
        assert np.array_equal(got, t(*[parts[k] for k in chain], rank=1)), (n_arr, pos)
    # error behaviour (reference util.py:564-577, :610-612, :750-753)
    with pytest.raises(ValueError):
        util.tensor_insert(t(X, I), pos=0, arr_dims=dims)
    with pytest.raises(ValueError):
        util.tensor_insert(t(X, I), Y, Z, pos=(0,), arr_dims=dims)
    with pytest.raises(IndexError):
        util.tensor_insert(t(X, I), Y, pos=(3,), arr_dims=dims)
    with pytest.raises(IndexError):
        util.tensor_merge(t(X, I), t(Y, Z), pos=(0, -3), arr_dims=dims, ins_dims=dims)
    with pytest.raises(ValueError):
        util.tensor_merge(t(X, I), t(Y, Z), pos=(0, 1), arr_dims=[[2, 2]], ins_dims=dims)
    with pytest.raises(ValueError):
        util.tensor_merge(t(X, I), t(Y, Z), pos=(0, 1), arr_dims=[[2, 3], [2, 2]], ins_dims=dims)
    # hashes: one per slice, signed zeros alike (reference util.py:1096-1100)
    h = util.hash_array_along_axis(np.array([[0.0, 1.0], [-0.0, 1.0], [0.0, 2.0]]))
    assert h[0] == h[1] != h[2]
    assert list(util.progressbar(range(3), disable=True)) == [0, 1, 2]


# public names of the reference's modules (v1.2.1; functions, classes, methods and properties)
REFERENCE_API = {'analytic': ['CDD', 'CPMG', 'FID', 'PDD', 'SE', 'UDD'],
 'basis': ['Basis', 'Basis.H', 'Basis.T', 'Basis.expand', 'Basis.four_element_traces',
           'Basis.from_partial', 'Basis.ggm', 'Basis.iscomplete', 'Basis.isherm', 'Basis.isnorm',
           'Basis.isorthogonal', 'Basis.isorthonorm', 'Basis.istraceless', 'Basis.normalize',
           'Basis.pauli', 'Basis.sparse', 'Basis.tidyup', 'equivalent_pauli_basis_elements',
           'expand', 'ggm_expand', 'normalize', 'remap_pauli_basis_elements'],
 'gradient': ['calculate_derivative_of_control_matrix_from_scratch',
              'calculate_filter_function_derivative', 'infidelity_derivative'],
 'numeric': ['calculate_control_matrix_from_atomic', 'calculate_control_matrix_from_scratch',
             'calculate_control_matrix_periodic', 'calculate_cumulant_function',
             'calculate_decay_amplitudes', 'calculate_filter_function',
             'calculate_frequency_shifts', 'calculate_noise_operators_from_atomic',
             'calculate_noise_operators_from_scratch',
             'calculate_pulse_correlation_filter_function',
             'calculate_second_order_filter_function_from_atomic',
             'calculate_second_order_filter_function_from_scratch', 'diagonalize',
             'error_transfer_matrix', 'infidelity'],
 'pulse_sequence': ['PulseSequence', 'PulseSequence.cache_control_matrix',
                    'PulseSequence.cache_filter_function', 'PulseSequence.cache_total_phases',
                    'PulseSequence.cleanup', 'PulseSequence.data', 'PulseSequence.diagonalize',
                    'PulseSequence.duration', 'PulseSequence.eigvals', 'PulseSequence.eigvecs',
                    'PulseSequence.frequency_data', 'PulseSequence.from_arrays',
                    'PulseSequence.get_control_matrix', 'PulseSequence.get_filter_function',
                    'PulseSequence.get_filter_function_derivative',
                    'PulseSequence.get_pulse_correlation_control_matrix',
                    'PulseSequence.get_pulse_correlation_filter_function',
                    'PulseSequence.get_total_phases', 'PulseSequence.intermediates',
                    'PulseSequence.is_cached', 'PulseSequence.nbytes', 'PulseSequence.omega',
                    'PulseSequence.propagator_at_arb_t', 'PulseSequence.propagators',
                    'PulseSequence.t', 'PulseSequence.tau', 'PulseSequence.total_propagator',
                    'PulseSequence.total_propagator_liouville', 'concatenate',
                    'concatenate_periodic', 'concatenate_without_filter_function', 'extend',
                    'remap'],
 'superoperator': ['liouville_is_CP', 'liouville_is_cCP', 'liouville_representation',
                   'liouville_to_choi'],
 'util': ['CalculationError', 'abs2', 'adot', 'all_array_equal', 'cexp', 'cexpm1', 'dot_HS',
          'get_indices_from_identifiers', 'get_sample_frequencies', 'hash_array_along_axis',
          'integrate', 'is_sequence_like', 'mdot', 'oper_equiv', 'parse_operators',
          'parse_optional_parameters', 'parse_spectrum', 'progressbar', 'progressbar_range',
          'remove_float_errors', 'tensor', 'tensor_insert', 'tensor_merge', 'tensor_transpose']}
NOT_BUILT = {'basis': {'Basis.sparse'}}       # needs the `sparse` package


def test_public_api_surface_covers_the_reference():
    """Every public function, class, method and property of the reference's modules (plotting aside,
    SURVEY section 2) exists here under the same name; package-level exports likewise."""
    import importlib
    for mod, names in REFERENCE_API.items():
        m = importlib.import_module(f'filter_functions_amd.{mod}')
        for name in names:
            if name in NOT_BUILT.get(mod, ()):
                continue
            obj = m
            for part in name.split('.'):
                assert hasattr(obj, part), f'{mod}.{name} missing'
                obj = getattr(obj, part)
    for name in ['Basis', 'PulseSequence', 'analytic', 'basis', 'concatenate', 'concatenate_periodic',
                 'error_transfer_matrix', 'extend', 'infidelity', 'liouville_representation',
                 'numeric', 'gradient', 'pulse_sequence', 'remap', 'util', 'superoperator',
                 'infidelity_derivative']:
        assert name in ff.__all__ and hasattr(ff, name), name


def test_qft_pulse_assembly_matches_reference_bit_exact():
    """BASELINE config 5: the 4-qubit QFT of examples/qft.py:42-136 assembled by this package's
    concatenation bookkeeping (identifier sort, zero-filled control amplitudes, noise
    sensitivities filled in from the other pulses, pulse_sequence.py:1340-1483) carries exactly
    the arrays the reference assembles."""
    import workloads as wl
    g = load_golden('qft')
    qft = wl.qft_pulse(ff)
    assert len(qft) == 13 and qft.d == 16
    for key in ('c_opers', 'c_coeffs', 'n_opers', 'n_coeffs', 'dt'):
        assert np.array_equal(getattr(qft, key), g[key]), key
    assert list(qft.c_oper_identifiers) == list(g['c_oper_identifiers'])
    assert list(qft.n_oper_identifiers) == list(g['n_oper_identifiers'])
    assert qft.basis.btype == str(g['btype']) == 'GGM'
    assert qft.tau == 13.0


def test_concatenate_accepts_any_iterable_and_single_pulse():
    """concatenate() takes any iterable exactly once (a generator too) and returns an independent
    copy -- caches included -- for a single pulse (reference pulse_sequence.py:1745-1752)."""
    X, Z = util.paulis[1], util.paulis[3]
    p = ff.PulseSequence([[X, [1.0], 'X']], [[Z, [1.0], 'Z']], [1.0])
    q = ff.PulseSequence([[Z, [0.5, 0.25], 'Zc']], [[Z, [1.0, 1.0], 'Z']], [1.0, 2.0])
    from_list = ff.concatenate([p, q])
    from_generator = ff.concatenate(pulse for pulse in (p, q))
    assert from_list == from_generator and len(from_generator) == 3
    with pytest.raises(TypeError):
        ff.concatenate(5)
    with pytest.raises(TypeError):
        ff.concatenate([p, 'not a pulse'])
    omega = [1.0, 2.0]
    R = np.arange(8, dtype=complex).reshape(1, 4, 2)
    p._data['total_propagator'] = np.eye(2, dtype=complex)
    p._data['total_propagator_liouville'] = np.eye(4)
    p.cache_control_matrix(omega, R)
    p.cache_filter_function(omega, filter_function=np.ones((1, 1, 2), complex))
    single = ff.concatenate(iter([p]))
    assert single is not p and single == p
    assert single.is_cached('control_matrix') and single.is_cached('filter_function')
    assert single.get_control_matrix(omega) is not R
    assert np.array_equal(single.get_control_matrix(omega), R)
    assert np.array_equal(single.get_total_phases(omega), p.get_total_phases(omega))


def test_lazy_cache_semantics():
    """Deferred entries are produced once, on first read; membership and nbytes never produce."""
    from filter_functions_amd._resident import Deferred, LazyCache
    calls = []

    def produce():
        calls.append(1)
        return np.ones(4)
    cache = LazyCache(a=1, b=Deferred(produce, nbytes=32))
    assert 'b' in cache and len(cache) == 2 and cache.stored_nbytes() == 32 and not calls
    twin = cache.copy()
    assert isinstance(twin, LazyCache) and not calls
    assert np.array_equal(cache['b'], np.ones(4)) and calls == [1]
    assert cache['b'] is cache.get('b') and calls == [1]               # produced once
    assert [k for k, _ in cache.items()] == ['a', 'b'] and len(cache.values()) == 2
    assert cache.setdefault('c', 5) == 5 and cache.setdefault('a', 7) == 1
    assert cache.get('missing', 3) == 3 and cache.pop('a') == 1
    import copy
    import pickle
    assert np.array_equal(copy.deepcopy(twin)['b'], np.ones(4))        # produces in the copy
    assert np.array_equal(pickle.loads(pickle.dumps(twin))['b'], np.ones(4))
    from types import MappingProxyType
    view = MappingProxyType(LazyCache(x=Deferred(lambda: 42)))
    assert view['x'] == 42 and dict(view.items()) == {'x': 42}


def test_resident_result_is_released_with_the_pulse(monkeypatch):
    """No reference cycle between a pulse, its deferred cache entries and its resident result: the
    device and pinned blocks go back to the pool as soon as the pulse is dropped, without waiting
    for the cyclic garbage collector (a stand-in result object: no GPU needed)."""
    import gc
    import weakref
    from filter_functions_amd import pulse_sequence

    class StandIn:
        alive = 0

        def __init__(self):
            StandIn.alive += 1
            self.shape = None

        def __del__(self):
            StandIn.alive -= 1

        def evaluate(self, c_opers, dt, t, omega, basis, n_opers, n_coeffs, c_coeffs=None):
            G, d, W, A = len(dt), c_opers.shape[1], len(omega), len(n_opers)
            F = np.zeros((A, A, W), complex)
            self._f = weakref.ref(F)
            return (np.zeros((G, d)), np.zeros((G, d, d), complex),
                    np.tile(np.eye(d, dtype=complex), (G + 1, 1, 1)), F)

        filter_function = property(lambda self: self._f())

        def control_matrix_nbytes(self):
            return 0

        def control_matrix(self):
            return np.zeros((1, 4, 2), complex)

    monkeypatch.setattr(pulse_sequence, 'ResidentResult', StandIn)
    X, Z = util.paulis[1], util.paulis[3]
    gc.collect()
    gc.disable()
    try:
        for _ in range(4):
            pulse = ff.PulseSequence([[X, [1.0]]], [[Z, [1.0]]], [1.0])
            F = pulse.get_filter_function([1.0, 2.0])
            assert StandIn.alive == 1                      # the previous pulse's result is gone
        assert pulse.is_cached('total_phases') and pulse.is_cached('control_matrix')
        assert np.allclose(pulse.get_total_phases([1.0, 2.0]), np.exp(1j*np.array([1.0, 2.0])))
        del pulse
        assert StandIn.alive == 0 and F.shape == (1, 1, 2)
    finally:
        gc.enable()


def test_basis_and_pulse_survive_pickling():
    import copy
    import pickle
    for basis in (Basis.pauli(2), Basis.ggm(3)):
        for clone in (pickle.loads(pickle.dumps(basis)), copy.deepcopy(basis)):
            assert clone.btype == basis.btype and clone.d == basis.d and clone.labels == basis.labels
            assert clone == basis and np.array_equal(np.asarray(clone), np.asarray(basis))
    X, Z = util.paulis[1], util.paulis[3]
    pulse = ff.PulseSequence([[X, [1.0, 2.0], 'X']], [[Z, [1.0, 1.0], 'Z']], [1.0, 0.5])
    clone = pickle.loads(pickle.dumps(pulse))
    assert clone == pulse and list(clone.n_oper_identifiers) == ['Z'] and clone.tau == 1.5


def test_shallow_copy_outlives_the_original_with_by_products_still_due(monkeypatch):
    """total phases and the Liouville propagator are produced on first read; a shallow
    copy must be able to produce them after the original pulse is gone (the reference computes
    them eagerly, so its copies always can)."""
    import copy
    import gc
    from filter_functions_amd import pulse_sequence
    X, Z = util.paulis[1], util.paulis[3]
    omega = np.array([0.5, 1.0, 2.0])
    R = np.arange(1*4*3, dtype=complex).reshape(1, 4, 3)
    seen = []
    monkeypatch.setattr(pulse_sequence, 'liouville_representation',
                        lambda U, basis: seen.append((U, basis)) or np.eye(len(basis)))
    # (a) control matrix handed in, pulse not diagonalised: phases from captured values
    pulse = ff.PulseSequence([[X, [1.0, 2.0]]], [[Z, [1.0, 1.0]]], [1.0, 0.5])
    pulse.cache_control_matrix(omega, R)
    twin = copy.copy(pulse)
    del pulse
    gc.collect()
    assert np.allclose(twin.get_total_phases(omega), np.exp(1j*omega*1.5))
    assert twin.get_control_matrix(omega) is R
    # (b) total propagator known when the by-products were deferred: captured by value
    pulse = ff.PulseSequence([[X, [1.0, 2.0]]], [[Z, [1.0, 1.0]]], [1.0, 0.5])
    U = np.array([[0, 1], [1, 0]], complex)
    pulse.total_propagator = U
    pulse.cache_control_matrix(omega, R)
    twin = copy.copy(pulse)
    del pulse
    gc.collect()
    assert np.array_equal(twin.total_propagator_liouville, np.eye(4))
    assert seen[-1][0] is U
    # (c) not diagonalised: the deferred entry of the twin diagonalises the TWIN
    pulse = ff.PulseSequence([[X, [1.0, 2.0]]], [[Z, [1.0, 1.0]]], [1.0, 0.5])
    pulse.cache_control_matrix(omega, R)
    twin = copy.copy(pulse)
    del pulse
    gc.collect()
    monkeypatch.setattr(pulse_sequence.PulseSequence, 'diagonalize',
                        lambda self: self._data.update(total_propagator=2*U))
    assert np.array_equal(twin.total_propagator_liouville, np.eye(4))
    assert np.array_equal(seen[-1][0], 2*U)


def test_merged_tables_of_a_pulse_set_are_remembered_and_safe():
    """`concatenate_without_filter_function` remembers, per set of distinct pulse OBJECTS, the merged operator
    tables (reference pulse_sequence.py:1340-1483 recomputes them per call): sequences drawn from one gate set
    -- randomized benchmarking -- pay them once.  The results must not alias the remembered tables, a pulse
    whose arrays are replaced must be re-merged, different orders of the same set must agree with a fresh
    evaluation."""
    from filter_functions_amd import pulse_sequence as ps
    X, Y, Z = util.paulis[1:]
    rng = np.random.default_rng(3)
    gates = [ff.PulseSequence([[X/2, rng.standard_normal(2), 'X']], [[Z/2, [1, 1], 'Z']], [1.0, 0.5]),
             ff.PulseSequence([[Y/2, rng.standard_normal(3), 'Y']], [[Z/2, [1, 1, 1], 'Z']], [0.3, 0.2, 0.1]),
             ff.PulseSequence([[X/2, rng.standard_normal(1), 'X'], [Y/2, rng.standard_normal(1), 'Y']],
                              [[Z/2, [1], 'Z'], [X/2, [0.5], 'Xn']], [0.7])]

    def fresh(seq):
        ps._MERGED.clear()
        return ps.concatenate_without_filter_function(seq, return_identifier_mappings=True)

    def same(a, b):
        (pa, ca, na), (pb, cb, nb) = a, b
        assert pa == pb and dict(ca.items()) == dict(cb.items()) and dict(na.items()) == dict(nb.items())
        for key in ('c_opers', 'n_opers', 'c_coeffs', 'n_coeffs', 'dt'):
            assert np.array_equal(getattr(pa, key), getattr(pb, key), equal_nan=True), key
        assert list(pa.c_oper_identifiers) == list(pb.c_oper_identifiers) and pa.tau == pb.tau

    for order in ([0, 1, 2, 1, 0, 0, 2], [2, 2, 1, 0], list(rng.integers(0, 3, 40))):
        seq = [gates[k] for k in order]
        want = fresh(seq)
        ps._MERGED.clear()
        first = ps.concatenate_without_filter_function(seq, return_identifier_mappings=True)
        assert len(ps._MERGED) == 1
        again = ps.concatenate_without_filter_function(seq, return_identifier_mappings=True)   # served from memory
        same(first, want)
        same(again, want)
        # no aliasing: scribbling over a result leaves the next one intact
        again[0].c_opers[...] = 0
        again[0].n_coeffs[...] = -1
        same(ps.concatenate_without_filter_function(seq, return_identifier_mappings=True), want)
    # a pulse whose coefficients are REPLACED is merged anew (in-place edits need cleanup, as for every cache)
    seq = [gates[0], gates[1], gates[0]]
    before = ps.concatenate_without_filter_function(seq)
    gates[1].c_coeffs = gates[1].c_coeffs*2.0
    after = ps.concatenate_without_filter_function(seq)
    assert not np.array_equal(before.c_coeffs, after.c_coeffs)
    same((after, {}, {}), (fresh(seq)[0], {}, {}))
    # ... and so is one whose arrays are modified IN PLACE (the reference rebuilds from the live arrays on every
    # call, pulse_sequence.py:1599-1665): with or without a cleanup in between
    want_tail = gates[0].c_coeffs.copy()
    gates[0].c_coeffs[0, :] = [10.0, 20.0]
    edited = ps.concatenate_without_filter_function(seq)
    assert np.array_equal(edited.c_coeffs[0, :2], [10.0, 20.0]) and np.array_equal(edited.c_coeffs[0, -2:], [10.0, 20.0])
    gates[0].dt[0] = 0.25
    gates[0].cleanup('all')
    assert not any(any(r() is gates[0] for r in hit['refs']) for hit in ps._MERGED.values())   # cleanup forgets
    assert ps.concatenate_without_filter_function(seq).dt[0] == 0.25
    gates[0].c_coeffs[...] = want_tail
    gates[0].dt[0] = 1.0
    same((ps.concatenate_without_filter_function(seq), {}, {}), (fresh(seq)[0], {}, {}))
    # an entry dies with its pulses: the remembered copies do not outlive the gate set
    import gc
    tmp = [ff.PulseSequence([[X/2, [0.1, 0.2], 'X']], [[Z/2, [1, 1], 'Z']], [1.0, 0.5]) for _ in range(2)]
    ps.concatenate_without_filter_function(tmp + tmp)
    n_before = len(ps._MERGED)
    del tmp
    gc.collect()
    ps.concatenate_without_filter_function(seq)
    assert len(ps._MERGED) < n_before + 1 and all(all(r() is not None for r in hit['refs']) for hit in ps._MERGED.values())
    ps.clear_merged_tables()
    assert len(ps._MERGED) == 0
    # identifiers that must be disambiguated by the position are never remembered
    clash = ff.PulseSequence([[Z/2, [1.0], 'X']], [[Z/2, [1], 'Z']], [1.0])      # 'X' names another matrix here
    ps._MERGED.clear()
    out = ps.concatenate_without_filter_function([gates[0], clash])
    assert len(ps._MERGED) == 0 and sorted(out.c_oper_identifiers) == ['X_0', 'X_1']

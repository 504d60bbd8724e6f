"""Input builders for the five BASELINE.json configurations.

They only produce INPUT data -- pulses, frequency grids, spectra -- and are written against the
public API shared by the reference (``filter_functions``) and this package
(``filter_functions_amd``): every builder that needs the package takes the module as its first
argument, so that ``oracle/make_golden.py`` can run it with the reference (fixtures), the tests
and ``bench.py`` with the HIP implementation.

cfg 1  README Hadamard                      -> :func:`hadamard_pulse`
cfg 2  random 2-qubit pulse, seed 42        -> :func:`random_pulse_inputs` (d=4, G=256, A=3)
cfg 3  1000-gate RB sequence                -> :func:`rb_cliffords` (naive gates), :func:`rb_cliffords_optimized`
                                               (the example's 100-segment pulses), :func:`rb_draw`
cfg 4  random 3-qubit pulse, seed 43        -> :func:`random_pulse_inputs` (d=8, G=512, A=9)
cfg 5  4-qubit QFT                          -> :func:`qft_pulse`

plus the one workload for which the reference documents wall-clock times of its own
(doc/source/examples/periodic_driving.ipynb): :func:`periodic_driving`.
"""
import numpy as np

CONFIG2 = dict(seed=42, d=4, G=256, A=3, n_cops=3, W=4096)
CONFIG4 = dict(seed=43, d=8, G=512, A=9, n_cops=3, W=65536, n_shards=8)
CONFIG3 = dict(n_gates=1000, W=8192, T=20.0, seed=0)
CONFIG5 = dict(N=4, tau=1.0, W=16384)


# ---- configs 2 and 4: the reference's rand_pulse_sequence recipe (tests/testutil.py:159-190) ----
def random_pulse_inputs(seed, d, G, A, n_cops=3, **_):
    """Traceless Hermitian control and noise operators from N(0,1)+iN(0,1) symmetrised,
    c_coeffs ~ N(0,1), n_coeffs ~ U[0,1), dt = 1 - U[0,1)."""
    rng = np.random.default_rng(seed)

    def herm_traceless(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        M = (M + M.conj().transpose(0, 2, 1))/2
        return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d
    c_opers, n_opers = herm_traceless(n_cops), herm_traceless(A)
    c_coeffs = rng.standard_normal((n_cops, G))
    n_coeffs = rng.random((A, G))
    dt = 1 - rng.random(G)
    return c_opers, c_coeffs, n_opers, n_coeffs, dt


def random_pulse_omega(dt, W):
    """omega = geomspace(1e-2/tau, 1e2/min dt, W); the spectrum of these configs is 1e-3/omega."""
    return np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)


# ---- config 1 ------------------------------------------------------------------------------------
def hadamard_pulse(ff):
    X, Y, Z = ff.util.paulis[1:]
    return ff.PulseSequence([[X/2, [0, np.pi]], [Y/2, [np.pi/2, 0]]], [[Z/2, [1, 1]]], [1, 1])


# ---- config 3: examples/randomized_benchmarking.py:95-151, naive gates ---------------------------
def rb_omega(W=8192, T=20.0, m_max=151):
    return 2*np.pi*np.geomspace(1e-2/(7*m_max*T), 1e2/T, W)


def rb_spectrum(omega, alpha=0.7):
    return 4e-11*(2*np.pi*1e-3/omega)**alpha/2.7241e-4**2


# the 24 single-qubit Cliffords as words in X/2 ('x') and Y/2 ('y'), applied left to right
CLIFFORD_WORDS = ('yyyy', 'xx', 'yy', 'yyxx', 'xy', 'xyyy', 'xxxy', 'xxxyyy', 'yx', 'yxxx', 'yyyx',
                  'yyyxxx', 'x', 'xxx', 'y', 'yyy', 'xyyyxxx', 'xxxyyyx', 'xxy', 'xxyyy', 'yyx',
                  'yyxxx', 'xyx', 'xyyyx')


def rb_cliffords(ff, omega, T=20.0):
    """X/2 and Y/2 atoms (control on X resp. Y, noise on X), control matrices cached at *omega*,
    and the 24 Cliffords built from them with ``@`` (concatenation rule)."""
    X, Y = ff.util.paulis[1], ff.util.paulis[2]
    atoms = {'x': ff.PulseSequence([[X/2, [np.pi/2/T], 'X']], [[X/2, [1], 'X']], [T]),
             'y': ff.PulseSequence([[Y/2, [np.pi/2/T], 'Y']], [[X/2, [1], 'X']], [T])}
    for atom in atoms.values():
        atom.cache_control_matrix(omega)
    cliffords = []
    for word in CLIFFORD_WORDS:
        gate = atoms[word[0]]
        for letter in word[1:]:
            gate = gate @ atoms[letter]
        cliffords.append(gate)
    return atoms, cliffords


def rb_cliffords_optimized(ff, omega, gates):
    """The example's OPTIMISED gate set (examples/randomized_benchmarking.py:112-128): X/2 and Y/2 as 100-segment
    exchange-coupled pulses, control on X with J = exp(eps[0]) and on Z with the constant B[0], noise on X.
    ``gates``: {'X2': (eps (3, 100), t (100,), B (3,)), 'Y2': ...} -- the arrays of the example's
    ``examples/data/X2ID.mat`` / ``Y2ID.mat`` (data; tests/golden/rb_optimized_gates.npz holds them).  Returns the
    atoms (control matrices cached at *omega*, computed from scratch on their 100 segments) and the 24 Cliffords built
    from them with ``@`` (100 to 700 segments each)."""
    X, Z = ff.util.paulis[1], ff.util.paulis[3]
    atoms = {}
    for letter, name in (('x', 'X2'), ('y', 'Y2')):
        eps, t, B = (np.asarray(a, dtype=float) for a in gates[name])
        t = np.ascontiguousarray(t.ravel())
        n_dt = len(t)
        c_coeffs = [np.exp(eps)[0], B.ravel()[0]*np.ones(n_dt)]
        atoms[letter] = ff.PulseSequence(list(zip((X/2, Z/2), c_coeffs, ('X', 'Z'))), [[X/2, np.ones(n_dt), 'X']], t)
        atoms[letter].cache_control_matrix(omega)
    cliffords = []
    for word in CLIFFORD_WORDS:
        gate = atoms[word[0]]
        for letter in word[1:]:
            gate = gate @ atoms[letter]
        cliffords.append(gate)
    return atoms, cliffords


def rb_draw(n_gates=1000, seed=0):
    return np.random.default_rng(seed).integers(0, len(CLIFFORD_WORDS), n_gates)


# ---- config 5: examples/qft.py:42-136 without qutip ----------------------------------------------
def qft_pulse(ff, N=4, tau=1.0, omega=None):
    """The N-qubit quantum Fourier transform as 2N+1 concatenated pulses (N=4: 13 segments, d=16,
    18 control and 18 noise operators with Pauli-string identifiers, default GGM basis).  With
    *omega* every one-segment pulse caches its control matrix first, so that each concatenation
    goes through the concatenation rule instead of leaving the filter function to be evaluated
    from scratch on the assembled pulse (the example's own route).

    Gate set: single-qubit rotations about X/Y (two of them make a Hadamard up to phase),
    simultaneous ZZ phase gates between qubit n and all later ones, and initial/final Z
    rotations; every noise operator is the control operator normalised to unit Hilbert-Schmidt
    norm with sensitivity 1."""
    I2, X, Y, Z = ff.util.paulis
    dim = 2**N

    def string_op(factors):
        """factors: {qubit: (letter, matrix)} -> (operator on the register, identifier)."""
        mats = [factors[q][1] if q in factors else I2 for q in range(N)]
        ident = ''.join(factors[q][0] if q in factors else 'I' for q in range(N))
        return ff.util.tensor(*mats), ident

    def pulse(terms):
        """terms: [(operator, identifier, amplitude)] -> one-segment pulse of length tau."""
        H_c = [[op, [amp], ident] for op, ident, amp in terms]
        H_n = [[op/np.sqrt(dim), [1], ident] for op, ident, _ in terms]
        new = ff.PulseSequence(H_c, H_n, [tau])
        if omega is not None:
            new.cache_control_matrix(omega)
        return new

    def rotation(k, theta, phi):
        return pulse([(*string_op({k: ('X', X)}), theta/2/tau*np.cos(phi)),
                      (*string_op({k: ('Y', Y)}), theta/2/tau*np.sin(phi))])

    def hadamard(k):
        return ff.concatenate([rotation(k, np.pi, 0), rotation(k, np.pi/2, -np.pi/2)])

    def z_layer(exponent):
        return pulse([(*string_op({k: ('Z', Z)}), np.pi/4*(1 - 2.0**exponent(k + 1))/tau)
                      for k in range(N)])

    def phase_layer(n):
        return pulse([(*string_op({n - 1: ('Z', Z), l - 1: ('Z', Z)}), -np.pi/4*2.0**(n - l)/tau)
                      for l in range(n + 1, N + 1)])

    pulses = [z_layer(lambda k: 1 - k)]
    for n in range(N - 1):
        pulses += [hadamard(n), phase_layer(n + 1)]
    pulses += [hadamard(N - 1), z_layer(lambda k: k - N)]
    return ff.concatenate(pulses, calc_pulse_correlation_FF=False, omega=omega)


def qft_matrix(N=4):
    """The unitary the QFT pulse implements after reversing the qubit order."""
    dim = 2**N
    j = np.arange(dim)
    return np.exp(2j*np.pi*np.outer(j, j)/dim)/np.sqrt(dim)


def bit_reversal(N=4):
    """Permutation matrix that reverses the order of the N qubits."""
    dim = 2**N
    P = np.zeros((dim, dim))
    for j in range(dim):
        P[int(format(j, f'0{N}b')[::-1], 2), j] = 1
    return P


# ---- the reference's own timed example: doc/source/examples/periodic_driving.ipynb -----------------
PERIODIC_DRIVING = dict(n_periods=10000, n_per_period=20, W=500,
                        published_s=dict(atomic_filter_function=0.0157, concatenate_periodic=0.0286,
                                         concatenate_standard=0.9008, echo_concatenation=0.0093,
                                         brute_force=38.38))


def periodic_driving(ff, n_periods=10000, n_per_period=20, W=500):
    """Weak resonant Rabi driving of one qubit in the lab frame (20 GHz drive, 1 MHz Rabi frequency):
    returns (X_ATOMIC, WAIT, NOT_FULL, omega): one drive period as a 20-segment pulse, a 1 ms idle
    pulse, the NOT gate written out as n_periods*n_per_period segments (200 000 in the notebook), and
    the 500 frequencies of the notebook.  Control on Z (static) and X (drive), noise on Z and X."""
    omega_d = 20e9*2*np.pi
    omega_0 = omega_d
    phi = np.pi/2
    amplitude = 1e6*2*np.pi
    T = 2*np.pi/omega_d
    X, _, Z = ff.util.paulis[1:]

    def drive(t):
        dt = np.diff(t)
        H_c = [[Z, [omega_0/2]*len(dt), 'Z'], [X, amplitude*np.sin(omega_d*t[1:] + phi), 'X']]
        H_n = [[Z, np.ones_like(dt), 'Z'], [X, np.ones_like(dt), 'X']]
        return ff.PulseSequence(H_c, H_n, dt)
    atomic = drive(np.linspace(0, T, n_per_period + 1))
    full = drive(np.linspace(0, T*n_periods, n_per_period*n_periods + 1))
    wait = ff.PulseSequence([[X, [0], 'X']], [[Z, [1.0], 'Z'], [X, [1.0], 'X']], [1e-3])
    omega = np.geomspace(1e-8*omega_0, 1e2*omega_0, W)
    return atomic, wait, full, omega

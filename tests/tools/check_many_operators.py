import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import ff_oracle as orc
import filter_functions_amd as ff
rng = np.random.default_rng(3)
for d, A, G, W in [(4, 12, 40, 300), (4, 16, 33, 100), (2, 40, 20, 77), (4, 9, 64, 130)]:
    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        return (M + M.conj().transpose(0, 2, 1))/2
    c_opers, n_opers = herm(2), herm(A)
    c_coeffs, n_coeffs = rng.standard_normal((2, G)), rng.random((A, G))
    dt = 1 - rng.random(G)*0.5
    omega = np.geomspace(1e-2, 50, W)
    ids = [f'n{i:02d}' for i in range(A)]
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs, ids)), dt, ff.Basis.pauli(int(np.log2(d))))
    F = pulse.get_filter_function(omega)
    R = pulse.get_control_matrix(omega)
    H = orc.hamiltonian(pulse.c_opers, pulse.c_coeffs)
    D, V, Q = orc.diagonalize(H, dt)
    Rr = orc.control_matrix_from_scratch(D, V, Q, omega, np.asarray(pulse.basis), pulse.n_opers, pulse.n_coeffs, dt)
    Fr = orc.filter_function(Rr)
    print(d, A, G, W, np.abs(R-Rr).max()/np.abs(Rr).max(), np.abs(F-Fr).max()/np.abs(Fr).max())
    assert np.abs(F-Fr).max()/np.abs(Fr).max() < 1e-10
print('ok')

import numpy as np, sys
sys.path.insert(0, '.')
import filter_functions_amd as ff
rng = np.random.default_rng(0)
U = np.linalg.qr(rng.standard_normal((4, 16, 16)) + 1j*rng.standard_normal((4, 16, 16)))[0]
L = ff.liouville_representation(U, ff.Basis.pauli(4))
print(L.shape)

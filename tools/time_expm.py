import sys, time, numpy as np
sys.path.insert(0, '.')
import filter_functions_amd as ff
rng = np.random.default_rng(0)
for scale in (1e-3, 1.0, 40.0):
    K = rng.standard_normal((256, 256))*scale/16
    ff.error_transfer_matrix(cumulant_function=K[None])
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); U = ff.error_transfer_matrix(cumulant_function=K[None]); ts.append(time.perf_counter() - t0)
    print(scale, 'median ms', np.median(ts)*1e3, 'min', min(ts)*1e3)

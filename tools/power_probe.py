"""Board power and shader clock (rocm-smi) while the config-2 pass runs back to back for a few seconds.
    python tools/power_probe.py [--seconds 4]          (FFK_TUNE_* / FFK_LIBRARY select the variant)
"""
import argparse, json, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import filter_functions_amd as ff  # noqa: E402
from filter_functions_amd.device import DevicePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=4.0)
    args = ap.parse_args()
    d, G, A, W = 4, 256, 3, 4096
    rng = np.random.default_rng(42)

    def herm(n):
        M = rng.standard_normal((n, d, d)) + 1j*rng.standard_normal((n, d, d))
        M = (M + M.conj().transpose(0, 2, 1))/2
        return M - np.trace(M, axis1=1, axis2=2)[:, None, None]*np.eye(d)/d
    c_opers, n_opers = herm(3), herm(A)
    c_coeffs, n_coeffs = rng.standard_normal((3, G)), rng.random((A, G))
    dt = 1 - rng.random(G)
    omega = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W)
    pipe = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, ff.Basis.pauli(2), omega, spectrum=1e-3/omega)
    stream = torch.cuda.current_stream().cuda_stream
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout
                samples.append((time.perf_counter(), json.loads(out)))
            except Exception as e:  # noqa: BLE001
                samples.append((time.perf_counter(), {'error': str(e)}))
            time.sleep(0.1)
    th = threading.Thread(target=sampler)
    th.start()
    time.sleep(0.5)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(200):
            pipe.launch(stream=stream)
        torch.cuda.synchronize()
        n += 200
    t1 = time.perf_counter()
    time.sleep(0.3)
    stop.set()
    th.join()
    print(f'{n} passes in {t1 - t0:.2f} s = {(t1 - t0)/n*1e6:.1f} us per pass (one stream, back to back)')
    for ts, js in samples:
        tag = 'RUN ' if t0 <= ts <= t1 else 'idle'
        card = next(iter(js.values())) if js else {}
        if isinstance(card, dict):
            keys = [k for k in card if 'ower' in k or 'sclk' in k]
            print(tag, f'{ts - t0:6.2f}', {k: card[k] for k in keys})
        else:
            print(tag, js)


if __name__ == '__main__':
    main()

"""Random shapes: decay amplitudes through decay_gemm_sym256_kernel against the register-fed kernel
(FFK_DECAY_REGISTER_FED=1, read per call) on the same device data.    python tools/tuning/check_decay_forms.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from filter_functions_amd import _lib  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(11)
p = lambda t: ctypes.c_void_p(t.data_ptr())
worst = 0.0
for trial in range(24):
    A = int(rng.integers(1, 33))
    n_idx = int(rng.integers(1, A + 1))
    W = int(rng.choice([1024, 1500, 2048, 2049, 4095, 8200, 16384, 20011]))
    s_ndim = int(rng.integers(1, 3))
    N = 256
    dev = torch.device('cuda')
    gen = torch.Generator(device=dev).manual_seed(trial)
    R = torch.randn(A, N, W, 2, dtype=torch.float64, device=dev, generator=gen)
    S = torch.zeros((n_idx, W, 2) if s_ndim == 2 else (W, 2), dtype=torch.float64, device=dev)
    S[..., 0] = torch.rand(S.shape[:-1], dtype=torch.float64, device=dev, generator=gen) - 0.3
    Wg = W + int(rng.integers(0, 50))
    w_off = int(rng.integers(0, Wg - W + 1))
    omega = torch.sort(torch.rand(Wg, dtype=torch.float64, device=dev, generator=gen)*40)[0]
    idx = torch.from_numpy(rng.permutation(A)[:n_idx].astype(np.int32)).to(dev)
    need = 0
    for form in ('', '1'):                      # (the plan, and with it the workspace, follows the switch)
        if form:
            os.environ['FFK_DECAY_REGISTER_FED'] = form
        else:
            os.environ.pop('FFK_DECAY_REGISTER_FED', None)
        need = max(need, lib.ffk_decay_amplitudes_workspace_bytes(1, N, W, n_idx, s_ndim))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    outs = []
    for form in ('', '1'):
        if form:
            os.environ['FFK_DECAY_REGISTER_FED'] = form
        else:
            os.environ.pop('FFK_DECAY_REGISTER_FED', None)
        out = torch.full((n_idx, N, N), float('nan'), dtype=torch.float64, device=dev)
        _lib.check(lib.ffk_decay_amplitudes_shard_dev(p(R), 1, A, N, W, p(S), s_ndim, p(omega), Wg, w_off, p(idx), n_idx,
                                                      p(out), p(ws), need, None))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    err = np.abs(outs[0] - outs[1]).max()/np.abs(outs[1]).max()
    worst = max(worst, err)
    fills = n_idx*(W//128) >= 256
    print(f'A={A} n_idx={n_idx} W={W} Wg={Wg} off={w_off} s_ndim={s_ndim} symmetric-block form {"yes" if fills else "no "}: '
          f'max rel diff {err:.2e}, finite {np.isfinite(outs[0]).all()}')
print('worst', worst)
assert worst < 1e-12

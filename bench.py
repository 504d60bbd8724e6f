#!/usr/bin/env python
"""bench.py -- filter-function elements/s on BASELINE.json config 2, 1..8 GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of the hot path over one pulse, HBM-resident in and out: Hamiltonian ->
eigendecomposition + expm + cumulative propagators -> control matrix -> filter function ->
infidelity (ffk_pipeline_dev).  Headline workload (config 2): random 2-qubit pulse, d=4, 256
segments, 3 noise operators, Pauli basis, 4096 omega per GPU, seed 42.  With
N > 1 the frequency axis is sharded: every rank evaluates its own block of 4096 omega of a 4096*N
grid (weak scaling), one RCCL all-gather reassembles F(omega) on every rank and the infidelity is
integrated over the full grid.

Before the W warm-up steps an UNTIMED, disclosed clock pre-warm runs the same step until the
accumulate kernel's HIP-event time is stable (`prewarm` object of the JSON line): after an idle
period the part needs ~300 steps (35 ms) to reach its sustained clocks (profiles/r01_q_*), which a
`--steps 20 --warmup 5` run never gets to on its own.

Rank 0 prints ONE JSON line.  value = elements/s over all ranks, element count
E = n_seg * n_omega_total * n_nops * d^2 per step (BASELINE.json metric).  Besides the contract
keys the line carries `roofline` (+ `frac_step`), `cpu_baseline`, `api_call_ms` (the user-facing
PulseSequence.get_filter_function + ff.infidelity call on host arrays) and `configs`: BASELINE
configs 3, 4 (one rank's shard; with N > 1 the whole 65536-omega grid strong-scaled over the ranks)
and 5, each timed in-process after the headline.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import workloads as wl  # noqa: E402

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD datasheet); see DESIGN.md
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md


# ---- CPU baseline ---------------------------------------------------------------------------------
def cpu_baseline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum, budget_s=14.0):
    """The oracle (NumPy restatement of the reference's algorithm) timed on this box's host cores on
    full passes of the headline workload.  The BLAS thread count is chosen first: one pass per
    candidate, the fastest one is then timed for the rest of the budget."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ff_oracle as orc
    d = c_opers.shape[-1]
    A, G, W = len(n_opers), len(dt), len(omega)

    def one_pass():
        H = orc.hamiltonian(c_opers, c_coeffs)
        D, V, Q = orc.diagonalize(H, dt)
        R = orc.control_matrix_from_scratch(D, V, Q, omega, basis, n_opers, n_coeffs, dt)
        F = orc.filter_function(R)
        infid = orc.infidelity_from_filter_function(F, spectrum, omega, np.arange(A), d)
        return R, F, infid
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    n_cpu = os.cpu_count() or 1
    result = one_pass()                          # warm-up (page faults, BLAS thread pool)
    survey = {}
    best = None
    if threadpool_limits is not None:
        for threads in [t for t in (1, 4, 8, 16, 32, 64) if t <= n_cpu]:
            with threadpool_limits(limits=threads):
                t0 = time.perf_counter()
                one_pass()
                survey[threads] = time.perf_counter() - t0
        best = min(survey, key=survey.get)
    t0 = time.perf_counter()
    n = 0
    ctx = threadpool_limits(limits=best) if best else None
    if ctx is not None:
        ctx.__enter__()
    try:
        while True:
            result = one_pass()
            n += 1
            elapsed = time.perf_counter() - t0
            if elapsed >= budget_s - sum(survey.values()) or n >= 50:
                break
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    value = n*G*W*A*d*d/elapsed
    return dict(value=value, unit='elements/s', cores=int(best or n_cpu), kind='port',
                sample=f'{n} full passes of config 2 (d={d}, {G} segments, {A} noise ops, {W} omega) '
                       f'in {elapsed:.1f} s, NumPy/OpenBLAS oracle with {best or n_cpu} BLAS threads '
                       f'(fastest of one pass each at '
                       f'{ {k: round(v, 2) for k, v in survey.items()} } s; '
                       f'os.cpu_count()={n_cpu})'), result


# ---- HBM traffic of the dominant kernel from the PMC counters ---------------------------------------
def measure_hbm_traffic(omega_per_gpu):
    """HBM traffic of the accumulate kernel, per launch, from the PMC counters: two short child runs
    of this script under ``rocprofv3 --pmc`` (FETCH_SIZE and WRITE_SIZE in separate passes, as the
    MI355X guide prescribes; FETCH_SIZE counts half the bytes of wide streaming reads on gfx950,
    hence the factor 2).  Returns (bytes, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or \
            'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None, 'already running under a profiler'
    script = os.path.abspath(__file__)
    means = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out_dir = tempfile.mkdtemp(prefix='ffk_pmc_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            for key in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
                env.pop(key, None)
            cmd = [prof, '--pmc', counter, '--output-format', 'csv', '-d', out_dir, '--',
                   sys.executable, script, '--steps', '8', '--warmup', '2', '--child',
                   '--omega-per-gpu', str(omega_per_gpu)]
            res = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, timeout=90)
            if res.returncode != 0:
                return None, f'rocprofv3 --pmc {counter} exited with {res.returncode}'
            values = []
            for f in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if 'ctrl_accumulate' in row['Kernel_Name'] and row['Counter_Name'] == counter:
                            values.append(float(row['Counter_Value']))
            if not values:
                return None, f'no {counter} samples for the accumulate kernel'
            means[counter] = sum(values)/len(values)
        except (OSError, subprocess.SubprocessError) as err:
            return None, f'rocprofv3 --pmc {counter} failed: {err}'
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    traffic = (2.0*means['FETCH_SIZE'] + means['WRITE_SIZE'])*1024.0
    return traffic, ('live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this script, '
                     f"per launch: FETCH_SIZE {means['FETCH_SIZE']:.0f} KiB (x2), "
                     f"WRITE_SIZE {means['WRITE_SIZE']:.0f} KiB")


def probe_sharded_step_one_rank():
    """Host cost of the frequency-sharded step on ONE rank (RCCL process group of size 1, test hook
    FFK_FORCE_COLLECTIVE): tools/profile_ring_step.py in a child process with a timeout, once with the
    whole step -- pass, all-gather, integral -- replayed from one hipGraph (FFK_GRAPH_COLLECTIVE=1) and
    once with the exchange enqueued call by call (the default for N > 1).  Returns a dict for the
    JSON line; a failure or timeout of the child is reported, never raised."""
    import re
    import socket
    import subprocess
    script = os.path.join(ROOT, 'tools', 'profile_ring_step.py')
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or \
            'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return {'skipped': 'running under a profiler'}
    out = {}
    for label, graph_collective in (('graph_incl_collective', '1'), ('pass_graph_exchange_call_by_call', '')):
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, FFK_FORCE_COLLECTIVE='1', FFK_GRAPH_COLLECTIVE=graph_collective, RANK='0',
                   WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        try:
            res = subprocess.run([sys.executable, script, 'rccl', '2', '8', 'x'], env=env, capture_output=True,
                                 text=True, timeout=120)
            m = re.search(r'host enqueue ([0-9.]+) us/step, step ([0-9.]+) us', res.stdout)
            out[label] = ({'host_enqueue_ms_per_step': float(m.group(1))*1e-3, 'ms_per_step': float(m.group(2))*1e-3}
                          if m else {'error': (res.stderr or res.stdout)[-300:]})
        except (OSError, subprocess.SubprocessError) as err:
            out[label] = {'error': str(err)[:300]}
    out['note'] = ('one rank, RCCL all-gather of size 1 (test hook): what the host pays per sharded step; with N > 1 '
                   'the exchange stays call by call unless FFK_GRAPH_COLLECTIVE=1 (a multi-rank capture of the '
                   'collective cannot be rehearsed on a one-GPU box)')
    return out


class AccumulateTimer:
    """HIP events around the accumulate kernel, recorded by libffk on the stream the kernel is
    launched on (ffk_set_accumulate_events)."""

    def __init__(self, lib, _lib, n):
        self.lib, self._lib = lib, _lib
        self.armed = False      # while armed, steps must be enqueued call by call (not replayed)
        self.pairs = [[ctypes.c_void_p(), ctypes.c_void_p()] for _ in range(n)]
        for pair in self.pairs:
            for e in pair:
                _lib.check(lib.ffk_event_create(ctypes.byref(e)))

    def arm(self, j, gate_on_previous=False):
        """Events of launch j; gate_on_previous: the launch stream first waits for the stop event of
        launch j - 1 (another stream's accumulate kernel), so that the interval is this kernel's
        execution, not its wait for the other pass's blocks to retire."""
        self._lib.check(self.lib.ffk_set_accumulate_events(self.pairs[j][0], self.pairs[j][1]))
        self.armed = True
        if gate_on_previous and j > 0:
            self._lib.check(self.lib.ffk_set_accumulate_gate(self.pairs[j - 1][1]))

    def disarm(self):
        self._lib.check(self.lib.ffk_set_accumulate_events(None, None))
        self.armed = False

    def read_ms(self, upto=None):
        ms = ctypes.c_float()
        out = []
        for a, b in self.pairs[:upto]:
            self._lib.check(self.lib.ffk_event_elapsed_ms(a, b, ctypes.byref(ms)))
            out.append(ms.value)
        return out

    def close(self):
        for pair in self.pairs:
            for e in pair:
                self.lib.ffk_event_destroy(e)


def prewarm_clocks(step, sync, timer, max_s, adaptive, batch=100):
    """Untimed: run `step` in batches until the accumulate kernel's HIP-event time (last step of a
    batch) changes by less than 1 % between batches twice in a row and at least 0.3 s have
    passed, or `max_s` seconds are over.  Not adaptive (N > 1: every rank must issue the same
    collectives): a fixed 30 batches."""
    t0 = time.perf_counter()
    trace, steps, stable = [], 0, 0
    while True:
        for i in range(batch):
            if i == batch - 1:
                sync()              # the instrumented launch runs alone: a clean kernel time
                timer.arm(0)
            step()
        timer.disarm()
        sync()
        steps += batch
        trace.append(timer.read_ms(1)[0])
        elapsed = time.perf_counter() - t0
        if not adaptive:
            if steps >= 30*batch:
                break
            continue
        if len(trace) > 1 and abs(trace[-1] - trace[-2]) <= 0.01*trace[-1]:
            stable += 1
        else:
            stable = 0
        if (stable >= 2 and elapsed >= 0.3) or elapsed >= max_s:
            break
    return dict(steps=steps, seconds=time.perf_counter() - t0,
                accumulate_ms_first_batch=trace[0], accumulate_ms_last_batch=trace[-1],
                rule=('adaptive: batches of 100 steps until the accumulate kernel time is stable '
                      f'within 1 % twice in a row and 0.3 s have passed or {max_s} s' if adaptive
                      else 'fixed 3000 steps (N > 1: ranks must issue identical collectives)'))


# ---- the other BASELINE configurations, timed in-process after the headline -----------------------
def time_pipeline(pipe, torch, lib, _lib, stream, reps, extra=None, warm_s=0.25):
    """ms per pass of `pipe.launch` (+ `extra()` per pass), and of its accumulate kernel.  Untimed
    passes for `warm_s` seconds first (at least three): like the headline's pre-warm, so that the
    configs are timed at the sustained clock whatever ran (or idled) before them."""
    def run():
        pipe.launch(stream=stream)
        if extra is not None:
            extra()
    t_warm, n_warm = time.perf_counter(), 0
    while n_warm < 3 or time.perf_counter() - t_warm < warm_s:
        run()
        torch.cuda.synchronize()
        n_warm += 1
    timer = AccumulateTimer(lib, _lib, 1)
    t0 = time.perf_counter()
    for i in range(reps):
        if i == reps - 1:
            timer.arm(0)
        run()
    timer.disarm()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0)/reps*1e3
    kernel_ms = timer.read_ms()[0]
    timer.close()
    return ms, kernel_ms


def bench_config4_shard(ff, torch, lib, _lib, DevicePipeline, device, stream, rank=0, world=1):
    """d = 8, 512 segments, 9 noise operators: one rank's block of the 65536-omega grid."""
    from filter_functions_amd.parallel import shard_bounds
    cfg = wl.CONFIG4
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    omega_full = wl.random_pulse_omega(dt, cfg['W'])
    n = max(world, cfg['n_shards']) if world == 1 else world
    w0, w1 = shard_bounds(cfg['W'], n, rank if world > 1 else 0)
    omega = omega_full[w0:w1]
    basis = ff.Basis.pauli(3)
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)
    pipe = DevicePipeline(pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, basis,
                          omega, spectrum=1e-3/omega, device=device)
    ms, kernel_ms = time_pipeline(pipe, torch, lib, _lib, stream, reps=10)
    st = _lib.stats()
    E = cfg['G']*len(omega)*cfg['A']*cfg['d']**2
    return pipe, dict(
        config=4, workload=f"d=8, 512 segments, 9 noise ops, Pauli basis, seed 43: omega block "
                           f"[{w0}, {w1}) of the 65536-omega grid ({n} shards), diagonalize -> "
                           'infidelity, HBM-resident',
        ms=ms, elements_per_s=E/(ms*1e-3), dominant_kernel='ffk::ctrl_accumulate (d = 8)',
        kernel_ms=kernel_ms, executed_flops=st['accumulate_flops'],
        tflops=st['accumulate_flops']/(kernel_ms*1e-3)/1e12,
        frac=st['accumulate_flops']/(kernel_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
        geometry={k: st[k] for k in ('chunks', 'grid_x', 'grid_y', 'grid_z', 'block', 'lds_bytes')})


def bench_config4_full(ff, torch, lib, _lib, DevicePipeline, device, stream):
    """BASELINE config 4 WHOLE on one GPU: d = 8, 512 segments, 9 noise operators, all 65536 omega
    (the strong-scaling baseline of `scaling_model`: a measurement, not 8 x one block)."""
    cfg = wl.CONFIG4
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    omega = wl.random_pulse_omega(dt, cfg['W'])
    basis = ff.Basis.pauli(3)
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)
    pipe = DevicePipeline(pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, basis,
                          omega, spectrum=1e-3/omega, device=device)
    ms, kernel_ms = time_pipeline(pipe, torch, lib, _lib, stream, reps=3, warm_s=0.2)
    st = _lib.stats()
    E = cfg['G']*len(omega)*cfg['A']*cfg['d']**2
    del pipe
    torch.cuda.empty_cache()
    return dict(
        config='4 (whole grid, one GPU)',
        workload='d=8, 512 segments, 9 noise ops, Pauli basis, seed 43: ALL 65536 omega on one GPU, '
                 'diagonalize -> infidelity, HBM-resident',
        ms=ms, elements_per_s=E/(ms*1e-3), dominant_kernel='ffk::ctrl_accumulate (d = 8)',
        kernel_ms=kernel_ms, executed_flops=st['accumulate_flops'],
        tflops=st['accumulate_flops']/(kernel_ms*1e-3)/1e12,
        frac=st['accumulate_flops']/(kernel_ms*1e-3)/1e12/FP64_PEAK_TFLOPS)


def bench_other_dimensions(ff, torch, lib, _lib, DevicePipeline, device, stream):
    """The accumulate kernels no bench entry covered until round 6: ctrl.hip's symmetric
    generate / contract kernel at d = 2 and d = 3 (the reference's most common case: one qubit, one qutrit) and at
    d = 6 (the shape of doc/source/examples/calculating_quantum_processes.ipynb's CNOT: 6 noise operators, 250
    segments, 400 omega), and generic.hip's runtime-d kernel at d = 32 (five qubits).  Each entry: ms per pass
    (diagonalize -> infidelity, HBM-resident; d = 32: the control-matrix call on host arrays, transfers included),
    the accumulate kernel by HIP events, its executed flops (ffk_api.hip::accumulate_flops) and the FP64 fraction.
    Reference loop numeric.py:707-881."""
    entries = []
    shapes = [dict(d=2, G=256, A=3, W=4096, basis='GGM(2)', seed=52, kernel='ffk::ctrl_accumulate_d2_kernel<3> (ctrl_d2.hip)'),
              dict(d=3, G=256, A=3, W=4096, basis='GGM(3)', seed=53, kernel='ffk::ctrl_accumulate_kernel<3> (ctrl.hip)'),
              dict(d=6, G=250, A=6, W=400, basis='GGM(6)', seed=56, kernel='ffk::ctrl_accumulate_kernel<6> (ctrl.hip)'),
              dict(d=7, G=256, A=3, W=4096, basis='GGM(7)', seed=58,
                   kernel='pad -> ffk::ctrl_accumulate_pcr_kernel<3, true> (d = 8) -> unpad (ctrl.hip: launch_accumulate; '
                          'kernel_ms spans the four launches, the flops are the d = 8 kernel\'s)'),
              dict(d=2, G=4096, A=2, W=500, basis='GGM(2)', seed=57,
                   kernel='ffk::ctrl_accumulate_d2_kernel<2> (ctrl_d2.hip; until late round 6 the one-wave kernel of ctrl.hip)')]
    for sh in shapes:
        d, G, A, W = sh['d'], sh['G'], sh['A'], sh['W']
        c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(sh['seed'], d, G, A)
        omega = wl.random_pulse_omega(dt, W)
        basis = ff.Basis.ggm(d)
        pipe = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum=1e-3/omega,
                              device=device)
        ms, kernel_ms = time_pipeline(pipe, torch, lib, _lib, stream, reps=20, warm_s=0.1)
        st = _lib.stats()
        entries.append(dict(
            config=f"d={d}", workload=f"d={d}, {G} segments, {A} noise ops, {sh['basis']}, {W} omega, seed {sh['seed']}: "
                                      'diagonalize -> infidelity, HBM-resident',
            ms=ms, elements_per_s=G*W*A*d*d/(ms*1e-3), dominant_kernel=sh['kernel'], kernel_ms=kernel_ms,
            executed_flops=st['accumulate_flops'], tflops=st['accumulate_flops']/(kernel_ms*1e-3)/1e12,
            frac=st['accumulate_flops']/(kernel_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
            geometry={k: st[k] for k in ('chunks', 'grid_x', 'grid_y', 'grid_z', 'block', 'lds_bytes')}))
        del pipe
    # d = 32: the runtime-d kernels serve the array calls (DevicePipeline is compiled per dimension, d <= 16)
    d, G, A, W = 32, 100, 10, 1000
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(59, d, G, A)
    omega = wl.random_pulse_omega(dt, W)
    basis = ff.Basis.pauli(5)
    H = np.einsum('ijk,il->ljk', c_opers, c_coeffs)
    D, V, Q = ff.numeric.diagonalize(H, dt)
    timer = AccumulateTimer(lib, _lib, 1)
    best, kernel_ms = float('inf'), float('nan')
    for rep in range(3):
        timer.arm(0)
        t0 = time.perf_counter()
        ff.numeric.calculate_control_matrix_from_scratch(D, V, Q, omega, basis, n_opers, n_coeffs, dt)
        best = min(best, time.perf_counter() - t0)
        timer.disarm()
        kernel_ms = timer.read_ms()[0]
    st = _lib.stats()
    timer.close()
    entries.append(dict(
        config='d=32', workload='d=32 (five qubits), 100 segments, 10 noise ops, Pauli(5) (1024 elements), 1000 omega, '
                                'seed 59: calculate_control_matrix_from_scratch on host arrays (R is 164 MB: the '
                                'call is mostly its copy back)',
        ms=best*1e3, elements_per_s=G*W*A*d*d/best, dominant_kernel='ffk::accumulate_generic_kernel (generic.hip)',
        kernel_ms=kernel_ms, executed_flops=st['accumulate_flops'],
        tflops=st['accumulate_flops']/(kernel_ms*1e-3)/1e12,
        frac=st['accumulate_flops']/(kernel_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
        geometry={k: st[k] for k in ('chunks', 'grid_x', 'grid_y', 'grid_z', 'block', 'lds_bytes')}))
    return entries


def bench_config5(ff, torch, lib, _lib, DevicePipeline, device, torch_stream):
    """examples/qft.py: d = 16, 13 segments, 18 noise operators, 16384 omega: control matrix ->
    decay amplitudes -> cumulant function -> error transfer matrix."""
    W = wl.CONFIG5['W']
    omega = np.logspace(-2, 2, W)
    qft = wl.qft_pulse(ff)
    A = len(qft.n_opers)
    S = np.outer(1e-6*(np.arange(A) + 1), 1/omega)
    pipe = DevicePipeline(qft.c_opers, qft.c_coeffs, qft.n_opers, qft.n_coeffs, qft.dt, qft.basis,
                          omega, spectrum=S, device=device)
    stream = torch_stream.cuda_stream
    result = {}

    def etm():
        # everything of one pass on ONE stream: the sum over the operators and the copy to the host
        # must see this pass's cumulant function (torch's own ops follow torch's current stream)
        with torch.cuda.stream(torch_stream):
            gamma = pipe.decay_amplitudes(stream=stream)
            K = pipe.cumulant_function(gamma, stream=stream)
            result['U'] = pipe.error_transfer_matrix(K, stream=stream).cpu().numpy()
    ms, kernel_ms = time_pipeline(pipe, torch, lib, _lib, stream, reps=10, extra=etm)
    st = _lib.stats()
    E = len(qft.dt)*W*A*qft.d**2
    U = result['U']
    # the decay amplitudes alone (weights, the symmetric-block GEMM, the reduction over the frequency chunks)
    with torch.cuda.stream(torch_stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gamma_ms = []
        for _ in range(12):
            e0.record()
            pipe.decay_amplitudes(stream=stream)
            e1.record()
            e1.synchronize()
            gamma_ms.append(e0.elapsed_time(e1))
    gamma_ms = float(np.median(gamma_ms[2:]))
    N = qft.d**2
    gamma_flops = A*10*64*64*2.0*(2*W)          # ten of the sixteen 64 x 64 tiles of each symmetric 256 x 256 block
    return dict(
        config=5, workload='examples/qft.py 4-qubit QFT: d=16, 13 segments, 18 noise ops, GGM basis, '
                           '16384 omega; control matrix + F + infidelity -> decay amplitudes -> '
                           'cumulant function -> exp (error transfer matrix returned to the host)',
        ms=ms, elements_per_s=E/(ms*1e-3), dominant_kernel='ffk::ctrl_accumulate_mfma4 (d = 16)',
        kernel_ms=kernel_ms, executed_flops=st['accumulate_flops'],
        tflops=st['accumulate_flops']/(kernel_ms*1e-3)/1e12,
        frac=st['accumulate_flops']/(kernel_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
        entanglement_infidelity=float(1 - np.trace(U)/qft.d**2),
        decay_amplitudes=dict(ms=gamma_ms, executed_flops=gamma_flops, tflops=gamma_flops/(gamma_ms*1e-3)/1e12,
                              frac_whole_call=gamma_flops/(gamma_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
                              kernel='ffk::decay_gemm_sym256_kernel',
                              note='whole call (weights, GEMM, reduction of 14 frequency chunks) over the flops of '
                                   'the symmetric half; the GEMM kernel alone: profiles/r06_m_*'),
        geometry={k: st[k] for k in ('chunks', 'grid_x', 'grid_y', 'grid_z', 'block', 'lds_bytes')},
        note='one segment chunk: since round 4 the accumulate kernel expands its complete Y in the basis in '
             'its epilogue and writes the control matrix itself (no expansion launch); kernel_ms spans its two '
             'launches (16 + 2 operators) including that epilogue, tflops / frac count the accumulation flops '
             'alone over it')


def bench_liouville(ff, torch, lib, _lib, device, dense_basis=False):
    """superoperator.liouville_representation (the one kernel BASELINE's north_star
    names for the matrix cores): d = 16, batch 512, device resident (`tools/time_liouville.py` is the
    stand-alone version, with a parity check).  Pauli basis: since round 6 the conjugation kernel (matrix cores)
    contracts its tile with the operand's non-zeros itself and no GEMM runs; *dense_basis* (the Pauli basis rotated
    by a random unitary: Hermitian, orthonormal, no zeros) keeps the conjugation + GEMM form measured."""
    import ctypes
    d, B = 16, 512
    N = d*d
    rng = np.random.default_rng(0)
    basis = np.asarray(ff.Basis.pauli(4))
    if dense_basis:
        V = np.linalg.qr(rng.standard_normal((d, d)) + 1j*rng.standard_normal((d, d)))[0]
        basis = V @ basis @ V.conj().T
    U = np.linalg.qr(rng.standard_normal((B, d, d)) + 1j*rng.standard_normal((B, d, d)))[0]
    with torch.cuda.device(device):
        Ud = torch.from_numpy(U).to(device)
        Cd = torch.from_numpy(np.ascontiguousarray(basis)).to(device)
        out = torch.empty((B, N, N), dtype=torch.float64, device=device)
        need = lib.ffk_liouville_workspace_bytes(B, d, N)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        p = lambda t: ctypes.c_void_p(t.data_ptr())

        def run():
            _lib.check(lib.ffk_liouville_dev(p(Ud), B, d, p(Cd), N, 1, p(out), p(ws), need,
                                             ctypes.c_void_p(stream)))
        for _ in range(5):
            run()
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for _ in range(20):
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize(device)
            times.append(e0.elapsed_time(e1))
    ms = float(np.median(times))
    gemm = B*N*float(d*d)*N*2.0                # Hermitian basis: d^2 operand rows
    conj = B*N*(256 + 160)/4*512.0             # matrix instructions of the conjugation: 416 per four elements
    if dense_basis:
        return dict(
            config='K5-dense', workload='superoperator.liouville_representation: d=16 (N=256), a basis without zero '
                                        'entries (Pauli rotated by a random unitary), batch 512, device resident '
                                        '(basis conjugation on v_mfma_f64_4x4x4 + GEMM on v_mfma_f64_16x16x4 through LDS)',
            ms=ms, dominant_kernel='ffk::liouville_gemm_block_kernel<4>',
            executed_gemm_flops=gemm, tflops_whole_call=gemm/(ms*1e-3)/1e12,
            frac_whole_call=gemm/(ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
            note='whole launch sequence (operand build, conjugation, GEMM) over the GEMM flops the library '
                 'executes; per-kernel split and the GEMM alone (0.68 of the FP64 matrix peak): '
                 'profiles/r04_l_liouville_block_gemm.txt')
    return dict(
        config='K5', workload='superoperator.liouville_representation: d=16 (N=256), Pauli basis, batch 512, device '
                              'resident (round 6: basis conjugation on v_mfma_f64_4x4x4, contracted in the same '
                              'kernel with the 8-16 non-zero operand rows of each basis element; no GEMM, no operand '
                              'round trip through HBM)',
        ms=ms, dominant_kernel='ffk::conjugate_basis_mfma_kernel<16, true, true>',
        executed_matrix_flops=conj, tflops_whole_call=conj/(ms*1e-3)/1e12,
        frac_whole_call=conj/(ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
        hbm_bytes=float(B*N*N*8), hbm_gbs_whole_call=B*N*N*8/(ms*1e-3)/1e9,
        tflops_a_gemm_would_need=gemm/(ms*1e-3)/1e12,
        note='frac_whole_call counts the conjugation\'s matrix instructions only (the sparse contraction is LDS reads '
             'and vector FMAs); tflops_a_gemm_would_need = the rate the round-5 GEMM form (0.48 ms, entry K5-dense keeps '
             'it measured) would have to sustain to match this time; profiles/r06_l_liouville_fused.txt')


def bench_config3(ff):
    """examples/randomized_benchmarking.py: 1000 Clifford gates drawn from 24, 8192 omega, filter
    function by the concatenation rule (whole Python call, host arrays in and out)."""
    from filter_functions_amd import numeric, util
    cfg = wl.CONFIG3
    omega = wl.rb_omega(cfg['W'], cfg['T'])
    _, cliffords = wl.rb_cliffords(ff, omega, cfg['T'])
    draw = wl.rb_draw(cfg['n_gates'], cfg['seed'])
    seq = [cliffords[k] for k in draw]
    from filter_functions_amd import pulse_sequence as ps_mod
    times, cold = [], []
    for rep in range(12):
        t0 = time.perf_counter()
        total = ff.concatenate(seq)
        total.get_filter_function(omega)
        times.append(time.perf_counter() - t0)
    for rep in range(5):          # every call with the remembered merged tables of the gate set forgotten first
        ps_mod.clear_merged_tables()
        t0 = time.perf_counter()
        total = ff.concatenate(seq)
        total.get_filter_function(omega)
        cold.append(time.perf_counter() - t0)
    table = np.array([c.get_control_matrix(omega) for c in cliffords])
    phases = np.array([c.get_total_phases(omega) for c in cliffords])
    L = util.adot(np.array([p.total_propagator_liouville for p in seq[:-1]]))
    rule = []
    for _ in range(4):
        t0 = time.perf_counter()
        numeric.calculate_control_matrix_from_atomic_indexed(phases, table, draw.astype(np.int32), L)
        rule.append(time.perf_counter() - t0)
    E = cfg['n_gates']*cfg['W']*1*4
    # the device call alone (ffk_concatenate_sequence_resident: one H2D of index/basis/durations,
    # front launch, rule launch, F and the total propagator back), result left resident
    from filter_functions_amd import pulse_sequence as ps
    from filter_functions_amd._resident import ResidentResult
    _, distinct, _, index = ps._validated_sequence(seq)
    residents, taus = [p._resident for p in distinct], [p.tau for p in distinct]
    dev = []
    for _ in range(8):
        keep = ResidentResult()
        t0 = time.perf_counter()
        numeric.concatenate_sequence_resident(residents, taus, index, distinct[0].basis, which='total',
                                              return_filter_function=True, keep=keep)
        dev.append(time.perf_counter() - t0)
    # executed work of the rule kernel (from_atomic_block_kernel<1,4>): per position and frequency
    # 4 complex products with the running phase (24 flop), a real 4 x 4 matrix on 4 complex numbers
    # (64), the phase update (6) and the accumulation (8); bytes: the 24 control-matrix and phase
    # tables once per 64 frequencies (LDS-staged), R and F out
    G, W, T = cfg['n_gates'], cfg['W'], len(distinct)
    rule_flops = 102.0*G*W
    rule_bytes = 16.0*W*(T*5 + 4 + 1)
    # the rule kernel timed IN THIS RUN: HIP events recorded by the library around its launch, on the stream
    # it runs on (ffk_set_accumulate_events), over eight more device calls
    import ctypes
    from filter_functions_amd import _lib
    lib = _lib.load()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    for e in (e0, e1):
        _lib.check(lib.ffk_event_create(ctypes.byref(e)))
    rule_kernel = []
    ms_ev = ctypes.c_float()
    try:
        for _ in range(8):
            keep = ResidentResult()
            _lib.check(lib.ffk_set_accumulate_events(e0, e1))
            numeric.concatenate_sequence_resident(residents, taus, index, distinct[0].basis, which='total',
                                                  return_filter_function=True, keep=keep)
            _lib.check(lib.ffk_event_elapsed_ms(e0, e1, ctypes.byref(ms_ev)))
            rule_kernel.append(ms_ev.value)
    finally:
        _lib.check(lib.ffk_set_accumulate_events(None, None))
        for e in (e0, e1):
            _lib.check(lib.ffk_event_destroy(e))
    kernel_ms = float(np.mean(rule_kernel))
    kernel_src = ('live: HIP events around the rule launch on its stream, mean of 8 calls of this run '
                  f'(min {min(rule_kernel)*1e3:.1f}, max {max(rule_kernel)*1e3:.1f} us)')
    roof = None
    if kernel_ms:
        t_fp64 = rule_flops/(FP64_PEAK_TFLOPS*1e12)*1e3
        t_hbm = rule_bytes/(HBM_PEAK_GBS*1e9)*1e3
        roof = dict(bound='fp64_valu', executed_flops=rule_flops, algorithmic_bytes=rule_bytes,
                    bound_ms=max(t_fp64, t_hbm), frac=max(t_fp64, t_hbm)/kernel_ms,
                    note='a latency-bound launch: 128 blocks of 16 wavefronts, each a chain of 62 dependent '
                         'positions after staging 120 KiB of tables; see DESIGN.md')
    return dict(
        device_call_ms=min(dev)*1e3, kernel_ms=kernel_ms, kernel_ms_source=kernel_src, roofline=roof,
        config=3, workload='1000-gate randomized-benchmarking sequence (24 Cliffords from X/2, Y/2), '
                           'd=2, 1 noise op, 8192 omega, ff.concatenate + get_filter_function on host '
                           'arrays',
        ms=float(np.median(times))*1e3, elements_per_s=E/float(np.median(times)),
        ms_first=times[0]*1e3, ms_median=float(np.median(times))*1e3, ms_min=min(times)*1e3,
        ms_median_tables_forgotten=float(np.median(cold))*1e3,
        ms_note='12 identical calls: ms_first pays the arena, the block pools, the merge of the gate set\'s operator '
                'tables and the clocks\' ramp; from the second call on the merged tables of the 24 pulse objects are '
                'remembered (pulse_sequence._merged_tables: randomized benchmarking draws many sequences from one gate '
                'set); ms_median_tables_forgotten: five more calls, each after clear_merged_tables(); `ms` and '
                '`elements_per_s` are the MEDIAN of the 12 (rounds 3-5 reported the minimum)',
        dominant_kernel='ffk::from_atomic_block_kernel<1,4> (table rule + reduction + F, LDS-staged tables)',
        rule_call_ms=min(rule)*1e3,
        note='ms = whole Python call incl. host bookkeeping over 1000 pulse objects; the 24 Cliffords '
             'keep their control matrices resident in HBM and are read in place; device_call_ms = '
             'ffk_concatenate_sequence_resident alone (two launches: front = gather + running products + '
             'Liouville representations + total phases, rule = table rule + slab reduction + F); '
             'rule_call_ms = the indexed concatenation rule alone on host arrays (tables H2D + kernel + R D2H)')


def bench_config3_optimized(ff):
    """Config 3 with the example's OPTIMISED gate set (examples/randomized_benchmarking.py:112-151): X/2, Y/2 as
    100-segment pulses (data: tests/golden/rb_optimized_gates.npz), so the atoms come from the d = 2 from-scratch
    kernel, the 24 Cliffords (100-700 segments) from the concatenation rule, the 1000-gate sequence from the rule
    kernel with 332 200 segments of host bookkeeping."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'rb_optimized_gates.npz'))
    cfg = wl.CONFIG3
    omega = wl.rb_omega(cfg['W'], cfg['T'])
    gates = {name: (g[f'{name}_eps'], g[f'{name}_t'], g[f'{name}_B']) for name in ('X2', 'Y2')}
    t0 = time.perf_counter()
    atoms, cliffords = wl.rb_cliffords_optimized(ff, omega, gates)
    t_gate_set = time.perf_counter() - t0
    t_atoms = []
    for _ in range(5):
        t0 = time.perf_counter()
        for name in ('x', 'y'):
            fresh = ff.PulseSequence(list(zip(atoms[name].c_opers, atoms[name].c_coeffs)),
                                     list(zip(atoms[name].n_opers, atoms[name].n_coeffs)), atoms[name].dt)
            fresh.cache_control_matrix(omega)
        t_atoms.append(time.perf_counter() - t0)
    draw = wl.rb_draw(cfg['n_gates'], cfg['seed'])
    seq = [cliffords[k] for k in draw]
    times = []
    for _ in range(8):
        t0 = time.perf_counter()
        total = ff.concatenate(seq)
        total.get_filter_function(omega)
        times.append(time.perf_counter() - t0)
    infid = float(ff.infidelity(total, wl.rb_spectrum(omega), omega)[0])
    return dict(
        config='3 (optimised gates)',
        workload='the same 1000-gate sequence from the example\'s optimised X/2, Y/2 pulses (100 segments each, '
                 'examples/data/X2ID.mat, Y2ID.mat): 24 Cliffords of 100-700 segments, 332 200 segments in all, d=2, '
                 '1 noise op, 8192 omega',
        ms_first=times[0]*1e3, ms_median=float(np.median(times))*1e3, ms_min=min(times)*1e3,
        ms=float(np.median(times))*1e3, n_segments=int(len(total)),
        elements_per_s=cfg['n_gates']*cfg['W']*4/float(np.median(times)),
        gate_set_ms=t_gate_set*1e3, atoms_from_scratch_ms=float(np.median(t_atoms))*1e3, infidelity=infid,
        note='gate_set_ms: both atoms from scratch (ffk::ctrl_accumulate_d2_kernel, 100 segments, 8192 omega) + 41 '
             'concatenations building the 24 Cliffords, once; atoms_from_scratch_ms: the two atoms alone (PulseSequence + '
             'cache_control_matrix); ms: ff.concatenate + get_filter_function of the 1000-gate sequence, median of 8 '
             '(the coefficient tables of 332 200 segments are host bookkeeping)')


def bench_published_example(ff):
    """The one workload the reference publishes wall-clock times for
    (doc/source/examples/periodic_driving.ipynb, hardware unstated): a 20-segment drive period
    repeated 10 000 times, d=2, 2 noise operators, 500 omega -- by concatenate_periodic, by
    ff.concatenate over 10 000 pulse objects, and written out as 200 002 segments from scratch.
    Whole Python calls on host arrays.  Not BASELINE.json's metric: vs_baseline stays null."""
    from itertools import repeat
    cfg = wl.PERIODIC_DRIVING
    atomic, wait, full, omega = wl.periodic_driving(ff)

    def best(fn, reps):
        ts, out = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), out

    def atomic_ff():
        atomic.cleanup('all')
        atomic.cache_filter_function(omega)
    t_atomic, _ = best(atomic_ff, 3)
    t_periodic, not_periodic = best(lambda: ff.concatenate_periodic(atomic, cfg['n_periods']), 3)
    t_standard, not_standard = best(lambda: ff.concatenate(repeat(atomic, cfg['n_periods'])), 2)
    t_echo, echo = best(lambda: ff.concatenate((wait, not_periodic, wait)), 3)

    def brute():
        written_out = ff.concatenate((wait, full, wait), calc_filter_function=False)
        return written_out.get_filter_function(omega)
    t_brute, F_brute = best(brute, 2)
    F_echo = echo.get_filter_function(omega)
    rel = lambda a, b: float(np.abs(a - b).max()/np.abs(b).max())   # noqa: E731
    ours = dict(atomic_filter_function=t_atomic, concatenate_periodic=t_periodic,
                concatenate_standard=t_standard, echo_concatenation=t_echo, brute_force=t_brute)
    pub = cfg['published_s']
    return dict(
        config='published_example',
        workload='doc/source/examples/periodic_driving.ipynb: Rabi driving, 20-segment period x 10000, '
                 'd=2, 2 noise ops, 500 omega (whole Python calls, host arrays in and out)',
        ms={k: v*1e3 for k, v in ours.items()},
        reference_published_ms={k: v*1e3 for k, v in pub.items()},
        speedup_vs_published={k: pub[k]/v for k, v in ours.items()},
        elements_per_s_brute_force=(cfg['n_periods']*cfg['n_per_period'] + 2)*cfg['W']*2*4/t_brute,
        agreement=dict(periodic_vs_standard=rel(not_periodic.get_filter_function(omega),
                                                not_standard.get_filter_function(omega)),
                       concatenated_vs_brute_force=rel(F_echo, F_brute)),
        note='the reference notebook does not name its hardware; these ratios are not vs_baseline')


def bench_api_call(ff, c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum, reps=30):
    """perf_counter around the user-facing call on host arrays: a fresh PulseSequence each time
    (nothing cached), pulse.get_filter_function(omega) then ff.infidelity(pulse, S, omega)."""
    H_c, H_n = list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs))
    t_ff, t_inf = [], []
    for i in range(reps + 3):
        pulse = ff.PulseSequence(H_c, H_n, dt, basis)
        t0 = time.perf_counter()
        pulse.get_filter_function(omega)
        t1 = time.perf_counter()
        infid = ff.infidelity(pulse, spectrum, omega)
        t2 = time.perf_counter()
        if i >= 3:
            t_ff.append(t1 - t0)
            t_inf.append(t2 - t1)
    total = np.array(t_ff) + np.array(t_inf)
    # the other common usage (the reference's README): ff.infidelity straight on a fresh pulse -- path and
    # integral in one library call since round 3
    t_one = []
    for i in range(reps + 3):
        pulse = ff.PulseSequence(H_c, H_n, dt, basis)
        t0 = time.perf_counter()
        infid_one = ff.infidelity(pulse, spectrum, omega)
        t1 = time.perf_counter()
        if i >= 3:
            t_one.append(t1 - t0)
    assert np.allclose(infid_one, infid, rtol=1e-12, atol=0)
    return dict(api_call_ms=float(np.median(total)*1e3), api_call_ms_min=float(total.min()*1e3),
                get_filter_function_ms=float(np.median(t_ff)*1e3),
                infidelity_ms=float(np.median(t_inf)*1e3),
                infidelity_on_fresh_pulse_ms=float(np.median(t_one)*1e3)), infid


GPU_MODULES = ('torch', 'filter_functions_amd')


def self_launch_command(n_gpus, argv, port):
    """Command line of the child that runs the N ranks: one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
            f'--nproc-per-node={n_gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
            os.path.abspath(__file__), *argv]


def pick_json_line(text):
    """Rank 0's result line among everything the ranks wrote to stdout (the last one, should a
    library have printed a JSON-looking line of its own)."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith('{"metric"'):
            return line
    return None


def self_launch(n_gpus, argv, run=None):
    """`bench.py --gpus N` started WITHOUT torch.distributed.run (no WORLD_SIZE in the environment):
    start the N ranks as a fresh child (`python -m torch.distributed.run ... bench.py <same
    arguments>`; a child process, never os.exec*), pass its stderr through, print rank 0's one JSON
    line as this process's single stdout line and return the child's exit code.  This process must
    not have initialised the GPU (it is the parent of processes that will): checked."""
    import socket
    import subprocess
    loaded = [m for m in GPU_MODULES if m in sys.modules]
    if loaded:
        raise RuntimeError(f'the launcher process imported {loaded}: it must stay off the GPU')
    with socket.socket() as s:                   # a free port for the rendezvous
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = self_launch_command(n_gpus, argv, port)
    env = dict(os.environ, FFK_BENCH_LAUNCHER='self', HSA_ENABLE_IPC_MODE_LEGACY='0',
               OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '4'))
    for key in ('RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(key, None)
    print('bench.py: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    res = (run or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = pick_json_line(res.stdout or '')
    if line is not None:
        print(line, flush=True)
    else:                                        # nothing to relay: show what the ranks said
        sys.stderr.write(res.stdout or '')
    if res.returncode == 0 and line is None:
        return 1
    return res.returncode


# ---- scaling model ("the measured single-GPU throughput plus the modelled comm cost") ----
XGMI_LINK_GBS_PER_DIRECTION = 76.5     # 7 links x ~153 GB/s per GPU, bidirectional: half per direction
RCCL_SMALL_MESSAGE_LATENCY_MS = 0.025  # launch + rendezvous of one all-gather (assumed; not measured)


def measure_all_gather(torch, dist, device, world, rank, block_bytes_list, reps=100):
    """N > 1: the RCCL all-gather ALONE, per rank and at several block sizes (the headline's F block, config 4's),
    HIP events on the current stream, next to the two constants `scaling_model` assumes -- so that the first run on
    real xGMI validates or replaces them without a code change.  Returns a dict for rank 0: per
    size the per-rank mean / min in ms and, from the smallest and the largest size, the latency and the per-link rate
    the model's formula t = latency + bytes / rate implies."""
    rows = []
    for nbytes in block_bytes_list:
        n = max(2, int(nbytes)//16*2)
        mine = torch.full((n,), float(rank), dtype=torch.float64, device=device)
        full = torch.empty((world*n,), dtype=torch.float64, device=device)
        for _ in range(10):
            dist.all_gather_into_tensor(full, mine)
        torch.cuda.synchronize(device)
        dist.barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            dist.all_gather_into_tensor(full, mine)
            b.record()
        torch.cuda.synchronize(device)
        ms = [a.elapsed_time(b) for a, b in ev]
        stat = torch.tensor([sum(ms)/len(ms), min(ms)], dtype=torch.float64, device=device)
        every = [torch.zeros(2, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(every, stat)
        rows.append({'block_bytes': n*8, 'mean_ms_per_rank': [float(e[0]) for e in every],
                     'min_ms_per_rank': [float(e[1]) for e in every]})
    out = {'sizes': rows, 'model_assumes': {'link_GBps_per_direction': XGMI_LINK_GBS_PER_DIRECTION,
                                            'rccl_all_gather_latency_ms': RCCL_SMALL_MESSAGE_LATENCY_MS}}
    if len(rows) >= 2:
        (b0, t0), (b1, t1) = [(r['block_bytes'], max(r['mean_ms_per_rank'])) for r in (rows[0], rows[-1])]
        if b1 > b0 and t1 > t0:
            rate = (b1 - b0)/((t1 - t0)*1e-3)/1e9
            out['measured_fit'] = {'link_GBps_per_direction': rate, 'latency_ms': t0 - b0/(rate*1e9)*1e3,
                                   'note': 'slowest rank\'s mean at the smallest and the largest size, t = latency + '
                                           'block_bytes / rate (a rank receives its N - 1 blocks on N - 1 links at once)'}
    return out


def scaling_model(step_ms_one_gpu, f_block_bytes, configs):
    """Predicted step time and efficiency at N = 2, 4, 8 from what ONE GPU measures.  The path shards
    along omega; the only exchange is one all-gather of the rank's block of F (A x A x W/N complex
    doubles) per step over point-to-point xGMI links, one peer per link, so a rank receives N - 1 blocks
    in parallel.  Two bounds per N: exchange overlapped with the next step's compute (the sharded ring
    runs the collective on its own stream) and exchange exposed (added to the step)."""
    def exchange_ms(block_bytes):
        return RCCL_SMALL_MESSAGE_LATENCY_MS + block_bytes/(XGMI_LINK_GBS_PER_DIRECTION*1e9)*1e3

    def rows(compute_ms_of, block_bytes_of, strong):
        out = {}
        for n in (2, 4, 8):
            comp, comm = compute_ms_of(n), exchange_ms(block_bytes_of(n))
            lo, hi = max(comp, comm), comp + comm
            ideal = compute_ms_of(1)/n if strong else compute_ms_of(1)
            out[str(n)] = {'compute_ms': comp, 'all_gather_bytes_per_rank': int(block_bytes_of(n)),
                           'all_gather_ms': comm, 'ms_per_step_overlapped': lo, 'ms_per_step_exposed': hi,
                           'efficiency_overlapped': ideal/lo, 'efficiency_exposed': ideal/hi}
        return out
    model = {
        'assumptions': {
            'link_GBps_per_direction': XGMI_LINK_GBS_PER_DIRECTION,
            'rccl_all_gather_latency_ms': RCCL_SMALL_MESSAGE_LATENCY_MS,
            'note': 'xGMI is point to point: the N - 1 blocks a rank receives travel on N - 1 different '
                    'links at once; no term for the prologue (<= 1 MB, recomputed on every rank) nor for '
                    'the integral (on the full grid, identical on every rank).  MODEL, not a measurement: '
                    'no multi-GPU node has been available to this build (SCALE_r01..r03 skipped).'},
        'headline_weak': rows(lambda n: step_ms_one_gpu, lambda n: f_block_bytes, strong=False),
    }
    for c in configs or []:
        if c.get('config') == '4 (whole grid, one GPU)' and 'ms' in c:
            # measured in this run: the whole 65536-omega grid on one GPU
            full = c['ms']
            model['config4_strong'] = rows(lambda n: full/n, lambda n: 9*9*65536//n*16, strong=True)
            model['config4_strong']['one_gpu_ms_measured_whole_grid'] = full
        if c.get('config') == 5 and 'ms' in c:
            # measured whole on one GPU; control matrix + F + decay-amplitude partial integrals shard
            # along omega (all-gather of F: 18 x 18 x 16384/N; all-reduce of 256 x 256 x 18 partial
            # decay amplitudes = 9.4 MB), the cumulant function and the exponential (0.3 ms) do not
            serial = 0.3
            model['config5_strong'] = rows(lambda n: (c['ms'] - serial)/n + serial,
                                           lambda n: 18*18*16384//n*16 + 18*256*256*8, strong=True)
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--omega-per-gpu', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipeline-depth', type=int, default=8,
                    help='N > 1: buffer sets in flight (the all-gather of step i may complete while '
                         'steps i+1 .. i+depth-1 compute)')
    ap.add_argument('--event-steps', type=int, default=0,
                    help='time the accumulate kernel with HIP events on the last n timed steps '
                         '(default: max(5, steps/8))')
    ap.add_argument('--no-pmc', action='store_true',
                    help='skip the rocprofv3 --pmc child runs that measure the HBM traffic')
    ap.add_argument('--no-configs', action='store_true', help='headline only (no configs 3-5)')
    ap.add_argument('--streams', type=int, default=2,
                    help='N = 1: independent passes in flight (round robin over this many HIP streams, '
                         'one buffer set each); 1 = strictly one pass after the other')
    ap.add_argument('--no-graph', action='store_true',
                    help='enqueue every step call by call instead of replaying captured hipGraphs')
    ap.add_argument('--no-prewarm', action='store_true')
    ap.add_argument('--published-example', action='store_true',
                    help='also time the reference notebook periodic_driving (outside the hot-path scope)')
    ap.add_argument('--prewarm-max-s', type=float, default=1.0)
    ap.add_argument('--child', action='store_true',
                    help='profiler child run: headline loop only, no baseline / PMC / configs')
    args = ap.parse_args()
    if args.child:
        args.no_cpu_baseline = args.no_pmc = args.no_configs = True
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python3 bench.py --gpus N` as the driver types it: this process becomes the launcher and
        # never touches the GPU (no torch, no libffk); the ranks are fresh child processes
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    import filter_functions_amd as ff
    from filter_functions_amd import _lib
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import shard_bounds

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    # FFK_BENCH_REHEARSE=1: every rank on device 0 with gloo as the control plane -- a dry run of the
    # whole N > 1 flow (ring, one-sided gather over IPC, strong-scaled configs) on a one-GPU box;
    # the numbers of such a run mean nothing
    rehearse = bool(os.environ.get('FFK_BENCH_REHEARSE'))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    lib = _lib.load()
    _lib.check(lib.ffk_set_device(local_rank))
    if os.environ.get('FFK_SEGMENT_CHUNKS'):            # tuning knob, 0/unset = automatic
        _lib.check(lib.ffk_set_segment_chunks(int(os.environ['FFK_SEGMENT_CHUNKS'])))
    # under torch.distributed.run a process group exists even for one rank (lets a 1-GPU box
    # exercise the RCCL path with FFK_FORCE_COLLECTIVE=1)
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('FFK_FORCE_COLLECTIVE'))
    if use_dist and rehearse:
        dist.init_process_group('gloo')
    elif use_dist:
        dist.init_process_group('nccl', device_id=device)

    cfg = wl.CONFIG2
    d, G, A = cfg['d'], cfg['G'], cfg['A']
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    basis = ff.Basis.pauli(2)
    W_total = args.omega_per_gpu*world
    omega_full = wl.random_pulse_omega(dt, W_total)
    spectrum_full = 1e-3/omega_full
    w0, w1 = shard_bounds(W_total, world, rank)
    omega = omega_full[w0:w1]
    # identifiers sort the operators exactly as PulseSequence does
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)

    def make_pipe():
        return DevicePipeline(pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, basis,
                              omega, spectrum=spectrum_full[w0:w1], device=device)
    # With sharding, the all-gather + infidelity of step i run on a second stream while the next
    # steps compute (the steps are independent passes): `depth` sets of buffers, round robin.  The
    # accumulate kernel holds every CU's LDS, so a collective kernel that becomes ready while it
    # runs only starts when its blocks retire; with two buffer sets the next-but-one step then waits
    # for it (0.163 ms/step on one rank), with more the collective falls into the window of the
    # small kernels between two accumulate launches.
    depth = max(2, args.pipeline_depth + (args.pipeline_depth & 1))      # even
    depth = max(depth, 2*max(1, args.streams))
    n_streams = max(1, args.streams)
    # one rank with several passes in flight runs the same schedule as the sharded step, without
    # the exchange (gather = 'none'): passes round robin on the compute streams, the integral of
    # each pass on a stream of its own, so that the front of the next-but-one pass does not queue
    # behind it (89.1 against 92.7 us per step for everything on the pass's own stream)
    use_ring = use_dist or n_streams > 1
    pipes = [make_pipe() for _ in range(depth if use_ring else 1)]
    pipe = pipes[0]
    if use_ring:
        from filter_functions_amd.parallel import ShardedStepRing
        # Two explicitly created streams: HIP spreads created streams over the hardware queues,
        # whereas torch's default stream and one side stream shared a queue on this system (every
        # kernel of the trace on one queue, in submission order: nothing overlapped).
        compute_streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.streams))]
        compute_stream = compute_streams[0]
        comm_stream = torch.cuda.Stream(device=device)
        torch.cuda.synchronize(device)
        # FFK_GATHER: 'rccl' (default: the RCCL all-gather), 'auto' (the one-sided all-gather of
        # csrc/peer.hip if its set-up and a verified round trip succeed on every rank, else RCCL), 'push'
        # Steps are independent passes (one pulse each): with two passes in flight on two HIP
        # streams the latency-bound launches of one pass (eigensolver, scan, prologue, expansion,
        # integral: ~32 us on 256 wavefronts or fewer) run beside the accumulate kernel of the
        # other.  Fusing those launches was tried and is slower (profiles/r02_a_fusion_attempts.md).
        ring = ShardedStepRing(pipes, W_total, omega_full, spectrum_full, compute_streams,
                               comm_stream, world, rank,
                               gather=os.environ.get('FFK_GATHER', 'rccl') if use_dist else 'none',
                               use_graph=not args.no_graph)
    else:
        compute_stream = torch.cuda.current_stream(device)
    stream = compute_stream.cuda_stream
    # HIP events around the accumulate kernel on the LAST n_ev steps of the timed region: each
    # timed event record costs several us of barrier-packet handling on this stack, so only a
    # contiguous tail is instrumented (an isolated pair in a gap-free stream reads ~3.5 us long,
    # back-to-back pairs ~1 us), and the tail rather than the head because right after the
    # barrier the queue is still shallow.
    n_ev = args.event_steps or max(3, args.steps//8)   # (a gated launch cannot overlap its predecessor: few of them)
    n_ev = max(1, min(args.steps, n_ev))
    # with passes in flight one more launch is instrumented ahead of the measured ones and its time
    # discarded: it gives the first measured launch a stop event to be gated on (ungated, that launch
    # overlaps its predecessor and reads its wait for free CUs as well: 122 instead of 84 us)
    lead = 1 if (max(1, args.streams) > 1 and args.steps > n_ev) else 0
    # a short timed region (the driver's --steps 20) instruments 3 launches only: further launches
    # of the same schedule are instrumented right AFTER the closing bracket (untimed steps, queue
    # kept full by a few plain steps first) so that the mean is over at least 8 launches
    n_extra = max(0, 8 - n_ev)
    lead_extra = 1 if (n_extra and max(1, args.streams) > 1) else 0
    timer = AccumulateTimer(lib, _lib, n_ev + lead + n_extra + lead_extra)

    def step(i=None):
        if i is not None and i >= args.steps - n_ev - lead:
            timer.arm(i - (args.steps - n_ev - lead), gate_on_previous=max(1, args.streams) > 1)
        # instrumented launches go call by call (the events are recorded around the kernel inside
        # ffk_control_matrix_dev); everything else is replayed from captured graphs
        if use_ring:
            return ring.step(eager=timer.armed)
        if args.no_graph or timer.armed:
            pipe.launch(stream=stream, with_infidelity=True)
        else:
            pipe.graph(with_infidelity=True).launch(stream)
        return pipe.infid

    def sync():
        torch.cuda.synchronize(device)

    def measure():
        """Pre-warm, W warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize."""
        warm = None
        if not args.no_prewarm:
            warm = prewarm_clocks(step, sync, timer, args.prewarm_max_s, adaptive=not use_dist)
        for _ in range(args.warmup):
            step()
        timer.disarm()
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = step(i)
        issue = time.perf_counter() - t0        # host time to enqueue all steps
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)
        total = time.perf_counter() - t0
        timer.disarm()
        if n_extra:
            for _ in range(2*depth):
                step()
            for j in range(n_extra + lead_extra):
                timer.arm(n_ev + lead + j, gate_on_previous=max(1, args.streams) > 1 and j > 0)
                step()
            timer.disarm()
            torch.cuda.synchronize(device)
        return warm, last, issue, total

    prewarm, infid, t_issue, elapsed = measure()
    gather_fallback = None
    if use_dist and ring.peer is not None:
        # did a poll of the one-sided gather time out on ANY rank?  (collective decision; the error
        # word is 0 / 1 / 2)  If so the numbers above are void: measure again through RCCL.
        torch.cuda.synchronize(device)
        code = ring.peer.error.to(torch.int32).clone()
        dist.all_reduce(code, op=dist.ReduceOp.MAX)
        if int(code.item()) != 0:
            gather_fallback = f'one-sided all-gather timed out (code {int(code.item())}); measured again with RCCL'
            ring.peer.close()
            ring = ShardedStepRing(pipes, W_total, omega_full, spectrum_full, compute_streams,
                                   comm_stream, world, rank, gather='rccl', use_graph=not args.no_graph)
            prewarm, infid, t_issue, elapsed = measure()

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    # what lets the driver verify the run: the ranks that took part and the device each one used
    prop = torch.cuda.get_device_properties(local_rank)
    mine = dict(rank=rank, local_rank=local_rank, pid=os.getpid(),
                pci_bus_id='%04x:%02x:%02x' % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id),
                uuid=str(getattr(prop, 'uuid', '')))
    rank_devices = [mine]
    push_error_word = None
    if use_dist:
        rank_devices = [None]*world
        dist.all_gather_object(rank_devices, mine)
        if ring.peer is not None:
            torch.cuda.synchronize(device)
            word = ring.peer.error.to(torch.int32).clone()
            dist.all_reduce(word, op=dist.ReduceOp.MAX)
            push_error_word = int(word.item())
    pipe.check_status()                          # eigensolver flags of the device-resident run
    all_gather_measured = None
    if use_dist:
        # the collective alone at the headline's block size and at config 4's per-rank size: what `scaling_model`
        # assumes (76.5 GB/s per link and direction, 25 us) against what this node does
        all_gather_measured = measure_all_gather(torch, dist, device, world, rank,
                                                 [A*A*args.omega_per_gpu*16, 9*9*(65536//world)*16])
    gather_ab = {'fallback': gather_fallback} if gather_fallback else None
    if gather_fallback:
        os.environ['FFK_GATHER'] = 'rccl'        # the strong-scaled configs below follow suit
    if use_dist and not gather_fallback:
        # the other gather method on the same workload, 200 steps, for the record
        other = 'rccl' if ring.gather == 'push' else 'push'
        try:
            ring_b = ShardedStepRing(pipes, W_total, omega_full, spectrum_full, compute_streams,
                                     comm_stream, world, rank,
                                     gather=other if other == 'rccl' else 'auto',
                                     use_graph=not args.no_graph)
            if ring_b.gather == other:
                for _ in range(50):
                    ring_b.step()
                torch.cuda.synchronize(device)
                dist.barrier()
                tb = time.perf_counter()
                for _ in range(200):
                    ring_b.step()
                torch.cuda.synchronize(device)
                dist.barrier()
                tb = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=device)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                gather_ab = {'headline': ring.gather, other + '_ms_per_step': float(tb.item())/200*1e3}
                if ring_b.peer is not None:
                    gather_ab['push_error_word'] = int(ring_b.peer.error.cpu().item())
        except RuntimeError as err:
            gather_ab = {'headline': ring.gather, other: f'failed: {err}'}
    # latency of one pass on its own (one stream, nothing else in flight), for reference
    latency_ms = None
    if not use_dist:
        one = stream
        torch.cuda.synchronize(device)
        reps = 200
        t1 = time.perf_counter()
        for _ in range(reps):
            if args.no_graph:
                pipe.launch(stream=one, with_infidelity=True)
            else:
                pipe.graph(with_infidelity=True).launch(one)
        torch.cuda.synchronize(device)
        latency_ms = (time.perf_counter() - t1)/reps*1e3

    # dominant kernel: ctrl_accumulate, timed by HIP events on its own stream inside the region
    all_ms = timer.read_ms()
    in_region_ms = all_ms[lead:lead + n_ev]
    extra_ms = all_ms[lead + n_ev + lead_extra:]
    acc_ms = float(np.mean(in_region_ms + extra_ms))
    stats = _lib.stats()
    timer.close()

    # the other BASELINE configurations (every rank takes part in the strong-scaled config 4)
    configs = []
    if not args.no_configs:
        if use_dist:
            configs.append(bench_config4_strong(ff, torch, dist, lib, _lib, DevicePipeline, device,
                                                compute_stream, comm_stream, world, rank,
                                                args.pipeline_depth))
            configs.append(bench_config5_strong(ff, torch, dist, DevicePipeline, device, world, rank))
        elif rank == 0:
            configs.append(bench_config3(ff))
            configs.append(bench_config3_optimized(ff))
            configs.append(bench_config4_shard(ff, torch, lib, _lib, DevicePipeline, device, stream)[1])
            configs.append(bench_config4_full(ff, torch, lib, _lib, DevicePipeline, device, stream))
            configs.append(bench_config5(ff, torch, lib, _lib, DevicePipeline, device, compute_stream))
            configs.append(bench_liouville(ff, torch, lib, _lib, device))
            configs.append(bench_liouville(ff, torch, lib, _lib, device, dense_basis=True))
            configs.extend(bench_other_dimensions(ff, torch, lib, _lib, DevicePipeline, device, stream))
            if args.published_example:     # doc notebook (concatenate_periodic): outside SURVEY section 8
                configs.append(bench_published_example(ff))

    if rank == 0:
        E_step = G*W_total*A*d*d
        value = E_step*args.steps/elapsed
        achieved = stats['accumulate_flops']/(acc_ms*1e-3)/1e12
        step_tflops = stats['accumulate_flops']*args.steps/elapsed/1e12
        # HBM traffic of the same kernel, per launch: measured live by two short child runs of this
        # script under rocprofv3 --pmc (2*FETCH_SIZE + WRITE_SIZE as the MI355X guide prescribes for
        # gfx950); if the profiler is unavailable, the committed measurement of the same command
        # (profiles/k3_hbm_traffic.json) when the launch geometry matches, else null
        traffic, traffic_src = None, None
        if world == 1 and not args.no_pmc:
            traffic, traffic_src = measure_hbm_traffic(args.omega_per_gpu)
            if traffic is None:
                traffic_src = f'PMC run failed ({traffic_src}); '
        if traffic is None:
            try:
                with open(os.path.join(ROOT, 'profiles', 'k3_hbm_traffic.json')) as fh:
                    prof = json.load(fh)
                if prof['geometry'] == [stats[k] for k in ('grid_x', 'grid_y', 'grid_z', 'block')]:
                    traffic = prof['traffic_bytes']
                    traffic_src = (traffic_src or '') + 'committed: ' + prof['source']
            except (OSError, KeyError, ValueError):
                pass
        out = {
            'metric': 'filter-function elements/sec (n_seg*n_omega*n_nops*d^2) at d=4',
            'value': value, 'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed/args.steps*1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': f'BASELINE config 2: random 2-qubit pulse d={d}, {G} segments, '
                                   f'{A} noise ops, Pauli basis, {args.omega_per_gpu} omega per GPU '
                                   f'({W_total} total), seed 42; one step = diagonalize + control '
                                   'matrix + filter function + infidelity, HBM-resident',
                       'sharding': ('omega blocks, ' + ('one-sided all-gather of F over IPC-mapped '
                                    'peer memory (csrc/peer.hip)' if ring.gather == 'push' else
                                    'RCCL all-gather of F')) if use_dist else 'none',
                       'passes_in_flight': max(1, args.streams),
                       'enqueue': ('call by call' if args.no_graph else
                                   'hipGraph replay: one hipGraphLaunch per step (pass + integral); the '
                                   'HIP-event-instrumented launches call by call'),
                       'schedule': ('passes round robin on %d compute streams, the integral of each pass '
                                    'on a stream of its own, %d buffer sets' % (max(1, args.streams), depth))
                       if use_ring else 'one stream',
                       **({'REHEARSAL': 'all ranks on one GPU over gloo (FFK_BENCH_REHEARSE): the '
                                        'numbers of this line mean nothing'} if rehearse else {})},
            'single_stream_ms_per_step': latency_ms, 'gather_ab': gather_ab,
            'ranks_seen': dist.get_world_size() if use_dist else 1,
            'rank_devices': rank_devices,
            'distinct_devices': len({r['pci_bus_id'] for r in rank_devices}),
            'gather': ring.gather if use_ring else 'none', 'push_error_word': push_error_word,
            'launcher': ('self: bench.py started torch.distributed.run as a child process'
                         if os.environ.get('FFK_BENCH_LAUNCHER') == 'self' else
                         ('torch.distributed.run' if 'RANK' in os.environ else 'direct')),
            'prewarm': prewarm,
            'roofline': {
                'kernel': 'ffk::ctrl_accumulate_pq_kernel<3, true>', 'bound': 'mfma',
                'bound_detail': 'FP64 issue: vector FMA and 4x4x4 matrix instructions share one pipe and one peak',
                'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved/FP64_PEAK_TFLOPS, 'frac_step': step_tflops/FP64_PEAK_TFLOPS,
                'traffic': traffic, 'traffic_source': traffic_src,
                'frac_r4_algorithm': (838.0*A + 198.0*((A + 2)//3))*G*args.omega_per_gpu/(acc_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
                'frac_step_r4_algorithm': (838.0*A + 198.0*((A + 2)//3))*G*args.omega_per_gpu/(elapsed/args.steps)/1e12/FP64_PEAK_TFLOPS,
                'frac_r3_algorithm': ((16.0*d**3 + 6.0*d*d)*A + 18.0*(d*(d - 1) + 1) + 62.0)*G*args.omega_per_gpu/(acc_ms*1e-3)/1e12/FP64_PEAK_TFLOPS,
                'avg_launch_ms': acc_ms, 'launches_timed': n_ev + n_extra,
                'launches_in_timed_region': n_ev,
                'avg_launch_ms_in_timed_region': float(np.mean(in_region_ms)),
                'launch_ms_min_max': [float(np.min(in_region_ms + extra_ms)),
                                      float(np.max(in_region_ms + extra_ms))],
                'flops_per_launch': stats['accumulate_flops'],
                'note': 'FP64 issue bound (vector FMA and v_mfma_f64_4x4x4 share the pipe: 78.6 TFLOP/s either '
                        'way on MI355X); flops = FMA-counted flops the kernel EXECUTES '
                        '(ffk_api.hip::accumulate_flops).  Round 5 (ctrl_pq.hip): the second product on the '
                        'matrix cores as THREE real products (Gauss) with psi folded into the A operand once '
                        'per set of four frequencies: 624 per operator + 304 per tile = 2176 per (segment, '
                        'omega) at A = 3, where the round-4 kernel executed 2712 and the round-3 kernel 3656 '
                        'for the same elements.  frac = dominant kernel alone (HIP events, each instrumented '
                        'launch gated on the previous accumulate kernel), frac_step = the same flops over the '
                        'whole step time -- above frac when passes pipeline.  frac_r4_algorithm / '
                        'frac_step_r4_algorithm = the SAME elements priced at the round-4 kernel\'s executed '
                        'flops (2712): a speed comparison with BENCH_r04\'s 0.50, not a utilisation -- an '
                        'algorithm that needs fewer flops per element lowers frac and raises `value`; '
                        'frac_r3_algorithm likewise against BENCH_r03 (3656).  What bounds the kernel: '
                        'tools/fp64_mix_probe.hip (profiles/r05_a_*): the consumers\' instruction mix (13 vector '
                        '+ 3 matrix instructions per operator and four frequencies) alone, operands from LDS, '
                        'sustains 18 900 sets/us on this part = 41.6 us for this launch\'s 786 432 sets before '
                        'the tile generation (23 % of the issue slots) is added; DESIGN.md section 6.1.  '
                        'SURVEY 8(d)\'s (8 d^2 + 8) flop per element is the Liouville-space form, which this '
                        'kernel does not execute, and is not used here',
            },
            'roofline_hbm': {
                'bound': 'hbm', 'achieved': stats['accumulate_bytes']/(acc_ms*1e-3)/1e9,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': stats['accumulate_bytes']/(acc_ms*1e-3)/1e9/HBM_PEAK_GBS,
                'bytes_per_launch': stats['accumulate_bytes'],
                'note': 'not the binding roof: algorithmic bytes of the accumulate kernel '
                        '(partial sums out + operands in)',
            },
            'kernel_geometry': {k: stats[k] for k in ('chunks', 'grid_x', 'grid_y', 'grid_z',
                                                       'block', 'lds_bytes')},
            'seg_omega_nop_per_s': G*W_total*A*args.steps/elapsed,
            'host_enqueue_ms_per_step': t_issue/args.steps*1e3,
            'device': _lib.device_info()[0],
        }
        if configs:
            out['configs'] = configs
        if all_gather_measured is not None:
            out['all_gather_measured'] = all_gather_measured
        if world == 1:      # the model extrapolates from what ONE GPU measures
            out['scaling_model'] = scaling_model(elapsed/args.steps*1e3, A*A*args.omega_per_gpu*16, configs)
        if world == 1 and not args.child:
            api, infid_api = bench_api_call(ff, c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega,
                                            spectrum_full)
            out.update(api)
        if world == 1 and not args.child and not args.no_configs:
            out['sharded_step_one_rank'] = probe_sharded_step_one_rank()
        if world == 1 and not args.no_cpu_baseline:
            base, (R_ref, F_ref, infid_ref) = cpu_baseline(
                pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, np.asarray(basis),
                omega, spectrum_full)
            out['cpu_baseline'] = base
            F_gpu = pipe.filter_function.cpu().numpy()
            R_gpu = pipe.control_matrix.cpu().numpy()
            out['parity'] = {
                'control_matrix_max_rel_err': float(np.abs(R_gpu - R_ref).max()/np.abs(R_ref).max()),
                'filter_function_max_rel_err': float(np.abs(F_gpu - F_ref).max()/np.abs(F_ref).max()),
                'infidelity_max_rel_err': float(np.abs(infid.cpu().numpy() - infid_ref).max()
                                                / np.abs(infid_ref).max()),
                'api_infidelity_max_rel_err': float(np.abs(infid_api - infid_ref).max()
                                                    / np.abs(infid_ref).max()),
                'against': 'oracle (NumPy restatement of the reference), same inputs',
            }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def bench_config4_strong(ff, torch, dist, lib, _lib, DevicePipeline, device, compute_stream,
                         comm_stream, world, rank, depth):
    """Config 4 as BASELINE states it: the 65536-omega grid split over the ranks (strong scaling),
    all-gather of F (9 x 9 x 65536/N c128 per rank), infidelity over the full grid."""
    from filter_functions_amd.parallel import ShardedStepRing, shard_bounds
    cfg = wl.CONFIG4
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    W = cfg['W']
    omega_full = wl.random_pulse_omega(dt, W)
    S_full = 1e-3/omega_full
    w0, w1 = shard_bounds(W, world, rank)
    basis = ff.Basis.pauli(3)
    pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, basis)
    depth = 2
    pipes = [DevicePipeline(pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, basis,
                            omega_full[w0:w1], spectrum=S_full[w0:w1], device=device)
             for _ in range(depth)]
    ring = ShardedStepRing(pipes, W, omega_full, S_full, compute_stream, comm_stream, world, rank,
                           gather=os.environ.get('FFK_GATHER', 'rccl'))
    reps = 10
    for _ in range(2):
        ring.step()
    torch.cuda.synchronize(device)
    dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        ring.step()
    torch.cuda.synchronize(device)
    dist.barrier()
    torch.cuda.synchronize(device)
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item())/reps*1e3
    E = cfg['G']*W*cfg['A']*cfg['d']**2
    push_error = None
    if ring.peer is not None:
        torch.cuda.synchronize(device)
        push_error = int(ring.peer.error.cpu().item())       # 0 = no poll timed out
    return dict(config=4, scaling='strong', n_gpus=world, gather=ring.gather, push_error_word=push_error,
                workload=f'd=8, 512 segments, 9 noise ops, 65536 omega split over {world} ranks '
                         f'({w1 - w0} per rank), all-gather of F, infidelity over the full grid',
                ms=ms, elements_per_s=E/(ms*1e-3))


def bench_config5_strong(ff, torch, dist, DevicePipeline, device, world, rank):
    """Config 5 as BASELINE states it: the QFT's 16384 omega split over the ranks; every rank
    integrates its block of the decay amplitudes with the global trapezoid weights, the (18, 256,
    256) partial results are all-gathered and summed in rank order, cumulant function and matrix
    exponential run redundantly on every rank (parallel.sharded_error_transfer_matrix)."""
    from filter_functions_amd.parallel import shard_bounds, sharded_error_transfer_matrix
    W = wl.CONFIG5['W']
    omega = np.logspace(-2, 2, W)
    qft = wl.qft_pulse(ff)
    A = len(qft.n_opers)
    S = np.outer(1e-6*(np.arange(A) + 1), 1/omega)
    w0, w1 = shard_bounds(W, world, rank)
    pipe = DevicePipeline(qft.c_opers, qft.c_coeffs, qft.n_opers, qft.n_coeffs, qft.dt, qft.basis,
                          omega[w0:w1], spectrum=S[:, w0:w1], device=device)
    omega_dev = torch.from_numpy(omega).to(device)

    def one():
        pipe.launch(with_infidelity=False)
        return sharded_error_transfer_matrix(pipe, omega_dev, w0)
    for _ in range(2):
        one()
    torch.cuda.synchronize(device)
    dist.barrier()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        gamma, K, U = one()
    torch.cuda.synchronize(device)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item())/reps*1e3
    E = len(qft.dt)*W*A*qft.d**2
    return dict(config=5, scaling='strong', n_gpus=world,
                workload=f'examples/qft.py 4-qubit QFT, d=16, 13 segments, 18 noise ops, 16384 omega '
                         f'split over {world} ranks ({w1 - w0} per rank): control matrix -> decay '
                         'amplitudes (all-gather of the partial integrals, rank-ordered sum) -> '
                         'cumulant function -> exp on every rank',
                ms=ms, elements_per_s=E/(ms*1e-3),
                entanglement_infidelity=float(1 - np.trace(U)/qft.d**2))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""bench.py -- filter-function elements/s on BASELINE.json config 2, 1..8 GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of the hot path over one pulse, HBM-resident in and out: Hamiltonian ->
eigendecomposition + expm + cumulative propagators -> control matrix -> filter function ->
infidelity (ffk_pipeline_dev).  Workload (config 2): random 2-qubit pulse, d=4, 256 segments,
3 noise operators, Pauli basis, 4096 omega per GPU, seed 42.  With N > 1 the
frequency axis is sharded: every rank evaluates its own block of 4096 omega of a 4096*N grid
(weak scaling), one RCCL all-gather reassembles F(omega) on every rank and the infidelity is
integrated over the full grid.

Rank 0 prints ONE JSON line.  value = elements/s over all ranks, element count
E = n_seg * n_omega_total * n_nops * d^2 per step (BASELINE.json metric).
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

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD datasheet); see DESIGN.md
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md


def config2(seed=42, d=4, G=256, A=3, n_cops=3):
    """rand_pulse_sequence recipe of the reference's tests/testutil.py:159-190."""
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


def cpu_baseline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum, budget_s=12.0):
    """The oracle (NumPy restatement of the reference's algorithm) timed on this box's host cores:
    full passes of the same workload until ~budget_s seconds have been spent."""
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
    one_pass()                                   # warm-up (BLAS threads, page faults)
    t0 = time.perf_counter()
    n = 0
    while True:
        result = one_pass()
        n += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= budget_s or n >= 50:
            break
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    value = n*G*W*A*d*d/elapsed
    return dict(value=value, unit='elements/s', cores=int(threads), kind='port',
                sample=f'{n} full passes of config 2 (d={d}, {G} segments, {A} noise ops, '
                       f'{W} omega) in {elapsed:.1f} s, NumPy/OpenBLAS oracle, '
                       f'os.cpu_count()={os.cpu_count()}'), result


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
                   sys.executable, script, '--steps', '8', '--warmup', '2', '--no-cpu-baseline',
                   '--no-pmc', '--omega-per-gpu', str(omega_per_gpu)]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # Defaults long enough to measure sustained throughput: after an idle period the accumulate
    # kernel runs 95 us and settles at 81.6 us only ~300 steps (35 ms) later, as the clocks ramp
    # (profiles/r01_q_clock_ramp.txt); 2500 steps are 0.3 s of GPU time.
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=500)
    ap.add_argument('--omega-per-gpu', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipeline-depth', type=int, default=8,
                    help='N > 1: buffer sets in flight (the all-gather of step i may complete while '
                         'steps i+1 .. i+depth-1 compute)')
    ap.add_argument('--event-stride', type=int, default=8,
                    help='time the accumulate kernel with HIP events on the last steps/n steps')
    ap.add_argument('--no-pmc', action='store_true',
                    help='skip the rocprofv3 --pmc child runs that measure the HBM traffic')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import filter_functions_amd as ff
    from filter_functions_amd import _lib
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import gather_omega_shards, shard_bounds

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    lib = _lib.load()
    _lib.check(lib.ffk_set_device(local_rank))
    if os.environ.get('FFK_SEGMENT_CHUNKS'):            # tuning knob, 0/unset = automatic
        _lib.check(lib.ffk_set_segment_chunks(int(os.environ['FFK_SEGMENT_CHUNKS'])))
    # under torch.distributed.run a process group exists even for one rank (lets a 1-GPU box
    # exercise the RCCL path with FFK_FORCE_COLLECTIVE=1)
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('FFK_FORCE_COLLECTIVE'))
    if use_dist:
        dist.init_process_group('nccl', device_id=device)

    d, G, A = 4, 256, 3
    c_opers, c_coeffs, n_opers, n_coeffs, dt = config2(d=d, G=G, A=A)
    basis = ff.Basis.pauli(2)
    W_total = args.omega_per_gpu*world
    omega_full = np.geomspace(1e-2/dt.sum(), 1e2/dt.min(), W_total)
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
    pipes = [make_pipe() for _ in range(depth)] if use_dist else [make_pipe()]
    pipe = pipes[0]
    if use_dist:
        omega_full_dev = torch.from_numpy(omega_full).to(device)
        S_full_dev = torch.from_numpy(spectrum_full.astype(complex)).to(device)
        idx_dev = torch.arange(A, dtype=torch.int32, device=device)
        comm_stream = torch.cuda.Stream(device=device)
        free_events = [None]*depth
        # all-gather buffers (world, A, A, W_shard) and results, allocated once
        gathered = [torch.empty((world, A, A, w1 - w0), dtype=torch.complex128, device=device)
                    for _ in range(depth)]
        infid_out = [torch.empty(A, dtype=torch.float64, device=device) for _ in range(depth)]
        equal_shards = all(shard_bounds(W_total, world, r)[1] - shard_bounds(W_total, world, r)[0]
                           == w1 - w0 for r in range(world))

    if use_dist and not os.environ.get('FFK_BENCH_DEFAULT_STREAM'):
        # Two explicitly created streams: HIP spreads created streams over the hardware queues,
        # whereas torch's default stream and one side stream shared a queue on this system (every
        # kernel of the trace on one queue, in submission order: nothing overlapped).
        compute_stream = torch.cuda.Stream(device=device)
        comm_stream = torch.cuda.Stream(device=device)
        torch.cuda.synchronize(device)
    else:
        compute_stream = torch.cuda.current_stream(device)
    stream = compute_stream.cuda_stream
    # HIP events around the accumulate kernel on the last steps/`stride` steps of the timed region
    # (at least 25): each timed event record costs several us of barrier-packet handling on this
    # stack (kernel trace: the only two gaps of a step were the ones around the instrumented
    # launch), so instrumenting every step taxed the measured throughput by 3-4 %.  A contiguous
    # block, not every n-th step: an isolated event pair in an otherwise gap-free stream reads
    # ~3.5 us longer than the kernel (91.2 against rocprofv3's 87.6 us), back-to-back pairs ~1 us;
    # and the last steps, not the first: right after the barrier the queue is still shallow and the
    # clocks are ramping (first 25 steps: 96-97 us).
    stride = max(1, args.event_stride)
    n_ev = min(args.steps, max(25, (args.steps + stride - 1)//stride))
    ev = [[ctypes.c_void_p(), ctypes.c_void_p()] for _ in range(n_ev)]
    for pair in ev:
        for e in pair:
            _lib.check(lib.ffk_event_create(ctypes.byref(e)))
    counter = [0]

    def step(i=None):
        if i is not None:
            j = i - (args.steps - n_ev)
            if j >= 0:
                _lib.check(lib.ffk_set_accumulate_events(ev[j][0], ev[j][1]))
        if not use_dist:
            pipe.launch(stream=stream, with_infidelity=True)
            return pipe.infid
        k = counter[0] % depth
        counter[0] += 1
        p = pipes[k]
        # Buffer set k was last read by the gather of step c - depth.  The comm stream is in order,
        # so it is enough that the compute stream waits, every depth/2 steps, for the comm work of
        # depth/2 steps ago: for every step j of the following half-window, gather(j - depth) is
        # older than that.  (A wait per step costs ~4 us of barrier-packet handling each.)
        c = counter[0] - 1
        half = depth//2
        if c >= half and c % half == 0:
            compute_stream.wait_event(free_events[(c - half) % depth])
        p.launch(stream=stream, with_infidelity=False)
        ready = torch.cuda.Event()
        ready.record(compute_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            if equal_shards:
                # one collective into a preallocated buffer; the integral reads the shards in place
                dist.all_gather_into_tensor(torch.view_as_real(gathered[k]),
                                            torch.view_as_real(p.filter_function))
                out = p.infidelity_from_shards(gathered[k], omega_full_dev, S_full_dev, idx_dev,
                                               infid_out[k], stream=comm_stream.cuda_stream)
            else:
                F_full = gather_omega_shards(p.filter_function, W_total)
                out = p.infidelity_from(F_full, omega_full_dev, S_full_dev, idx_dev,
                                        stream=comm_stream.cuda_stream)
            done = torch.cuda.Event()
            done.record(comm_stream)
            free_events[k] = done
        return out

    for _ in range(args.warmup):
        step()
    _lib.check(lib.ffk_set_accumulate_events(None, None))
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        infid = step(i)
    t_issue = time.perf_counter() - t0          # host time to enqueue all steps
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    _lib.check(lib.ffk_set_accumulate_events(None, None))

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())

    # dominant kernel: ctrl_accumulate, timed by HIP events on its own stream inside the region
    ms = ctypes.c_float()
    acc_ms = []
    for a, b in ev:
        _lib.check(lib.ffk_event_elapsed_ms(a, b, ctypes.byref(ms)))
        acc_ms.append(ms.value)
    acc_ms = float(np.mean(acc_ms))
    stats = _lib.stats()
    for pair in ev:
        for e in pair:
            lib.ffk_event_destroy(e)

    if rank == 0:
        E_step = G*W_total*A*d*d
        value = E_step*args.steps/elapsed
        achieved = stats['accumulate_flops']/(acc_ms*1e-3)/1e12
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
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                       'k3_hbm_traffic.json')) as fh:
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
                       'sharding': 'omega blocks, RCCL all-gather of F' if use_dist else 'none'},
            'roofline': {
                'kernel': 'ffk::ctrl_accumulate_pc_kernel<4,3>', 'bound': 'mfma',
                'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved/FP64_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_src,
                'avg_launch_ms': acc_ms, 'launches_timed': n_ev,
                'flops_per_launch': stats['accumulate_flops'],
                'note': 'FP64 compute bound (vector = matrix peak 78.6 TFLOP/s on MI355X); '
                        'flops = FMA-counted flops of the Hilbert-space algorithm actually run; '
                        'a pure v_fma_f64 stream on pseudo-random operands sustains 55 TFLOP/s on '
                        'this part (tools/fp64_data_probe.hip, profiles/r01_k_*)',
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
                'against': 'oracle (NumPy restatement of the reference), same inputs',
            }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

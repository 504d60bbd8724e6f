"""The one-sided all-gather of the frequency-sharded step (csrc/peer.hip, parallel.PeerGather) with
two ranks.  A gpurun box has one GPU: both ranks use device 0 (IPC mappings between processes work
on one device as across devices; RCCL refuses two ranks on one device, so the control plane is
gloo).  The 8-GPU run itself is the driver's."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp  # noqa: E402

from conftest import ROOT  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, depth, n_steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist

    import filter_functions_amd as ff
    import workloads as wl
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import ShardedStepRing, shard_bounds
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    cfg = dict(wl.CONFIG2, G=24)
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    W = 512
    omega = wl.random_pulse_omega(dt, W)
    S = 1e-3/omega
    basis = ff.Basis.pauli(2)
    w0, w1 = shard_bounds(W, world, rank)
    pipes = [DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega[w0:w1],
                            device=device) for _ in range(depth)]
    compute, comm = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    ring = ShardedStepRing(pipes, W, omega, S, compute, comm, world, rank, gather='push')
    assert ring.gather == 'push'
    outs = [ring.step() for _ in range(n_steps)]
    torch.cuda.synchronize(device)
    ring.peer.check()
    results = np.array([o.cpu().numpy() for o in outs[-depth:]])
    F_gathered = ring.peer.gathered[(n_steps - 1 + ring.count_offset) % depth].cpu().numpy()
    whole = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum=S,
                           device=device)
    whole.launch()
    torch.cuda.synchronize(device)
    np.savez(os.path.join(out_dir, f'peer{rank}.npz'), results=results, F_gathered=F_gathered,
             F_whole=whole.filter_function.cpu().numpy(), infid_whole=whole.infid.cpu().numpy())
    dist.barrier()
    ring.peer.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('depth,n_steps', [(2, 7), (4, 13)])
def test_one_sided_gather_two_ranks_on_one_device(tmp_path, depth, n_steps):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), depth, n_steps, str(tmp_path)), nprocs=world,
             join=True)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f'peer{rank}.npz'))
        F = got['F_gathered']                                    # (world, A, A, W/world)
        F = F.transpose(1, 2, 0, 3).reshape(3, 3, -1)
        # the blocks are evaluated with a chunk count that depends on the block width: equal up to
        # the re-association of the segment sum
        assert np.abs(F - got['F_whole']).max() <= 1e-13*np.abs(got['F_whole']).max()
        for infid in got['results']:                             # every step computes the same pulse
            assert np.abs(infid - got['infid_whole']).max() <= 1e-13*np.abs(got['infid_whole']).max()


def _failing_worker(rank, world, port, out_dir):
    """Rank 1 stalls (host side) for longer than the poll timeout: rank 0's wait must time out
    (code 2), rank 0 must then POISON its signals instead of announcing blocks it could not
    exchange, rank 1 must fail on the poison (code 3) when it resumes, and the collective check
    must raise on both ranks (a skipped push used to be signalled as delivered)."""
    import time
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist

    import filter_functions_amd as ff
    import workloads as wl
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import ShardedStepRing, shard_bounds
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    cfg = dict(wl.CONFIG2, G=8)
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    W = 256
    omega = wl.random_pulse_omega(dt, W)
    basis = ff.Basis.pauli(2)
    w0, w1 = shard_bounds(W, world, rank)
    depth = 2
    pipes = [DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega[w0:w1],
                            device=device) for _ in range(depth)]
    compute, comm = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    ring = ShardedStepRing(pipes, W, omega, 1e-3/omega, compute, comm, world, rank, gather='push')
    for _ in range(4):
        ring.step()
    ring.check()                                  # collective, healthy so far
    ring.peer.set_timeout_ms(150.0)
    dist.barrier()
    if rank == 0:
        for _ in range(3):                        # rank 1 is not stepping: wait(c) times out
            ring.step()
        code_before = ring.peer.error_code()
    else:
        time.sleep(1.5)
        code_before = ring.peer.error_code()      # nothing has failed HERE yet
    dist.barrier()
    for _ in range(2):                            # both step again: rank 1 meets the poison
        ring.step()
    code_after = ring.peer.error_code()
    raised = False
    try:
        ring.check()
    except RuntimeError:
        raised = True
    np.savez(os.path.join(out_dir, f'fail{rank}.npz'), before=code_before, after=code_after,
             raised=raised)
    dist.barrier()
    ring.peer.close()
    dist.destroy_process_group()


def test_one_sided_gather_failure_reaches_every_rank(tmp_path):
    world = 2
    mp.spawn(_failing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), 'fail0.npz'))
    r1 = np.load(os.path.join(str(tmp_path), 'fail1.npz'))
    assert int(r0['before']) == 2                 # rank 0: its peer's signal never came
    assert int(r1['before']) == 0
    assert int(r0['after']) == 2                  # sticky: the first failure stays
    assert int(r1['after']) == 3                  # rank 1: told by the poison, not by a timeout
    assert bool(r0['raised']) and bool(r1['raised'])


def _captured_collective_worker(rank, world, port, out_dir):
    """One rank, nccl (= RCCL) backend: the whole sharded step -- pass, all-gather, integral -- replayed
    from one hipGraph per buffer set (FFK_GRAPH_COLLECTIVE=1) against the same steps enqueued call by
    call.  (A multi-rank capture cannot be rehearsed on a one-GPU box: RCCL refuses two ranks on one
    device.)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), FFK_FORCE_COLLECTIVE='1',
                      FFK_GRAPH_COLLECTIVE='1')
    import torch.distributed as dist

    import filter_functions_amd as ff
    import workloads as wl
    from filter_functions_amd.device import DevicePipeline
    from filter_functions_amd.parallel import ShardedStepRing
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    cfg = dict(wl.CONFIG2, G=16)
    c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
    W = 320
    omega = wl.random_pulse_omega(dt, W)
    basis = ff.Basis.pauli(2)
    pipes = [DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, device=device)
             for _ in range(4)]
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    comm = torch.cuda.Stream(device=device)
    ring = ShardedStepRing(pipes, W, omega, 1e-3/omega, streams, comm, world, rank, gather='rccl',
                           use_graph=True)
    assert ring.graph_collective
    outs = []
    for i in range(30):
        out = ring.step(eager=(i % 7 == 0))
        torch.cuda.synchronize(device)
        outs.append(out.cpu().numpy().copy())
    whole = DevicePipeline(c_opers, c_coeffs, n_opers, n_coeffs, dt, basis, omega, spectrum=1e-3/omega,
                           device=device)
    whole.launch()
    torch.cuda.synchronize(device)
    np.savez(os.path.join(out_dir, 'captured.npz'), outs=np.array(outs), ref=whole.infid.cpu().numpy(),
             nodes=ring._step_graphs[('coll', 1)].nodes)
    dist.destroy_process_group()


def test_sharded_step_with_the_collective_captured_in_a_graph(tmp_path):
    mp.spawn(_captured_collective_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(str(tmp_path), 'captured.npz'))
    assert int(got['nodes']) >= 7                          # pass (5) + collective + integral
    for out in got['outs']:
        assert np.array_equal(out, got['outs'][0])         # replayed == call by call (step 0), bit for bit
    assert np.abs(got['outs'][0] - got['ref']).max() <= 1e-13*np.abs(got['ref']).max()

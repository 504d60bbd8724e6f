"""Host cost of one sharded step (ShardedStepRing.step) on one rank, by cProfile:
    FFK_FORCE_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \\
        --master-addr 127.0.0.1 --master-port 29555 tools/profile_ring_step.py [push|rccl]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import filter_functions_amd as ff  # noqa: E402
import workloads as wl  # noqa: E402
from filter_functions_amd.device import DevicePipeline  # noqa: E402
from filter_functions_amd.parallel import ShardedStepRing  # noqa: E402

gather = sys.argv[1] if len(sys.argv) > 1 else 'push'
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2*n_streams
device = torch.device('cuda', 0)
torch.cuda.set_device(0)
if gather != 'none':
    dist.init_process_group('nccl', device_id=device)
cfg = wl.CONFIG2
c_opers, c_coeffs, n_opers, n_coeffs, dt = wl.random_pulse_inputs(**cfg)
omega = wl.random_pulse_omega(dt, 4096)
pulse = ff.PulseSequence(list(zip(c_opers, c_coeffs)), list(zip(n_opers, n_coeffs)), dt, ff.Basis.pauli(2))
pipes = [DevicePipeline(pulse.c_opers, pulse.c_coeffs, pulse.n_opers, pulse.n_coeffs, dt, pulse.basis,
                        omega, spectrum=1e-3/omega, device=device) for _ in range(depth)]
streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
comm = torch.cuda.Stream(device=device)
ring = ShardedStepRing(pipes, 4096, omega, 1e-3/omega, streams, comm, 1, 0, gather=gather,
                       use_graph=not os.environ.get("FFK_NO_GRAPH"))
for _ in range(500):
    ring.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    ring.step()
issue = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f'gather={ring.gather} streams={n_streams} depth={depth}: host enqueue {issue/2000*1e6:.1f} us/step, step {total/2000*1e6:.1f} us')
if len(sys.argv) > 4:
    if gather != 'none':
        dist.destroy_process_group()
    sys.exit(0)
prof = cProfile.Profile()
prof.enable()
for _ in range(2000):
    ring.step()
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats('tottime').print_stats(16)
if gather != 'none':
    dist.destroy_process_group()

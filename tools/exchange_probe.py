#!/usr/bin/env python
"""What the Det step's exchange costs, by route (one-rank nccl group on one GPU is enough for the host / queue costs):
   torch.distributed on the main stream | torch.distributed on a side stream | RCCL called directly on the main stream
each eager and captured into a hipGraph.   python -m torch.distributed.run --nproc-per-node 1 ... tools/exchange_probe.py"""
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))


def timed(fn, steps=200, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    return round(sorted(ts)[len(ts) // 2], 4)


def main():
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dev = torch.device('cuda', torch.cuda.current_device())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)
    out = {}
    n = 2401
    send = torch.randn(n, device=dev); recv = torch.empty(world * n, device=dev)
    x = torch.randn(600, 128, device=dev)

    def work():     # stand-in for the kernels around the exchange (a few small launches)
        return (x * 1.0001).sum()

    def torch_main():
        work(); dist.all_gather_into_tensor(recv, send); work()
    side = torch.cuda.Stream(); ev = torch.cuda.Event()

    def torch_side():
        work()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(recv, send)
            ev.record(side)
        work()
        torch.cuda.current_stream().wait_event(ev)
    out['work_only_ms'] = timed(lambda: (work(), work()))
    for _ in range(5):
        torch_main(); torch_side()
    out['torch_main_ms'] = timed(torch_main)
    out['torch_side_ms'] = timed(torch_side)

    # RCCL directly, on torch's current stream
    from epropnp import sharding
    try:
        comm = sharding.RcclComm()
        def direct():
            work(); comm.all_gather(recv, send); work()
        for _ in range(5):
            direct()
        torch.cuda.synchronize()
        assert torch.equal(recv[rank * n:(rank + 1) * n], send)
        out['rccl_direct_main_ms'] = timed(direct)
    except Exception as e:
        out['rccl_direct_error'] = repr(e)[:300]
        comm = None
    print(json.dumps(out), flush=True)

    def capture(fn, name):
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            g.replay(); torch.cuda.synchronize()
            out[name] = timed(g.replay)
        except Exception as e:
            out[name] = 'error: ' + repr(e)[:200]
        print(json.dumps({name: out[name]}), flush=True)
    which = sys.argv[1:] or ['direct', 'torch_main', 'torch_side']
    capture(lambda: (work(), work()), 'graph_work_only_ms')
    if comm is not None and 'direct' in which:
        capture(direct, 'graph_rccl_direct_ms')
    if 'torch_main' in which:
        capture(torch_main, 'graph_torch_main_ms')
    if 'torch_side' in which:
        capture(torch_side, 'graph_torch_side_ms')
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

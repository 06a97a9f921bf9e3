"""A captured hipMemsetAsync NODE replays garbage once eager launches have run between two replays (ROCm 7.0 user space,
PyTorch 2.10): PyTorch only, no part of this repository involved.  DBG_VARIANT = what happens between the two checks
(barrier | allreduce | sync | eager_sum | barrier_ids).  Kernel nodes (the captured sum, a fill_ kernel) and memcpy nodes replay
correctly; the memset node leaves what look like its own launch arguments in the buffer.

    for v in sync eager_sum barrier; do DBG_VARIANT=$v python tools/ubench/graph_memset_node.py; done
"""
import os, sys, ctypes, torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29534')
V=os.environ.get('DBG_VARIANT','barrier')
dev=torch.device('cuda:0')
dist.init_process_group('nccl', rank=0, world_size=1)
hip=ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes=[ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
x=torch.rand(153600, device=dev)
buf=torch.zeros(4096, device=dev, dtype=torch.int32)
keep={}
def body():
    s=x.sum()*0.5
    hip.hipMemsetAsync(buf.data_ptr(), 0xff, buf.numel()*4, torch.cuda.current_stream().cuda_stream)
    t=buf.clone()          # what the memset left (the clone itself is a memcpy node)
    f=torch.empty_like(buf).fill_(-1)   # a fill KERNEL node for comparison
    c=x[:4096].clone()     # a memcpy node
    keep.update(f=f, c=c)
    buf.zero_()            # dirty it again (fill kernel) so that a skipped memset shows
    return s, t
for _ in range(3): body()
torch.cuda.synchronize()
side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s,t=body()
ref=float(x.sum()*0.5)
def check(tag):
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    print(V, tag, 'sum ok' if abs(float(s)-ref) < 1e-2*abs(ref) else f'SUM WRONG {float(s)} vs {ref}', 'memset ok' if bool((t == -1).all()) else f'MEMSET WRONG {t[:4].tolist()}', 'fill kernel ok' if bool((keep['f'] == -1).all()) else 'FILL KERNEL WRONG', 'memcpy ok' if torch.equal(keep['c'], x[:4096]) else 'MEMCPY WRONG', flush=True)
check('before')
if V == 'barrier': dist.barrier()
elif V == 'allreduce': dist.all_reduce(torch.ones(1, device=dev))
elif V == 'sync': torch.cuda.synchronize()
elif V == 'eager_sum': float(torch.rand(500000, device=dev).sum())
elif V == 'barrier_ids': dist.barrier(device_ids=[0])
torch.cuda.synchronize()
check('after')

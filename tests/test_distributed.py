"""Multi-process object sharding on the gloo backend, world_size 2, CPU: every rank runs the hot
path on its contiguous shard of the objects (kernel logic through the test-only emulation build), then one all-gather
of the per-object outputs; the result must equal the single-process run.  On the GPU box the same code path runs over
RCCL (backend 'nccl'), which bench.py --gpus N exercises."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import epropnp_oracle as orc
from helpers import make_layer_objects, pack_noise


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run_shard(prob, noise_packed, dof, S, K, lo, hi):
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    sub = {k: (v[lo:hi] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == prob['x3d'].shape[0] else v)
           for k, v in prob.items()}
    p, cam, cf = make_layer_objects(sub, 'cpu', relative_delta=0.5)
    cf.set_param(p['x2d'], p['w2d'])
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    layer = cls(mc_samples=S, num_iter=K, solver=LMSolver(dof=dof, num_iter=3))
    out = layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'],
                                    force_init_solve=False, noise=noise_packed[lo:hi].contiguous())
    return out[0], out[4]


def _worker(rank, world, port, dof, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import conftest
        from epropnp import _hip, sharding
        from epropnp.losses import MonteCarloPoseLoss
        import install as emu
        emu.install(conftest._emu_lib())
        B, N, S, K = 5, 48, 32, 2          # 5 objects over 2 ranks: uneven tail
        prob = orc.make_problem(B, N, dof, seed=40)
        noise = pack_noise(orc.make_noise(B, S, K, dof, seed=41), dof)
        lo, hi = sharding.shard_range(B)
        pose_l, logw_l = _run_shard(prob, noise, dof, S, K, lo, hi)
        pose = sharding.gather_objects(pose_l, B, obj_dim=0)
        logw = sharding.gather_objects(logw_l, B, obj_dim=1)
        pose_full, logw_full = _run_shard(prob, noise, dof, S, K, 0, B)
        ok = pose.shape == (B, pose_full.shape[1]) and logw.shape == (S, B)
        ok = ok and (pose - pose_full).abs().max().item() < 1e-6 and (logw - logw_full).abs().max().item() < 1e-5
        loss = MonteCarloPoseLoss(init_norm_factor=1.0, momentum=0.5)
        loss(logw_l, torch.zeros(hi - lo), torch.tensor(float(rank + 1)))       # world mean of (1, 2) = 1.5
        ok = ok and abs(loss.norm_factor.item() - 1.25) < 1e-6
        # the same exchange as ONE collective: pose outputs and the loss's norm_factor scalar in one payload
        ex = sharding.ObjectExchange(B).start(pose_l, torch.tensor(float(rank + 1)))
        loss1 = MonteCarloPoseLoss(init_norm_factor=1.0, momentum=0.5)
        loss1(logw_l, torch.zeros(hi - lo), ex)
        ok = ok and abs(loss1.norm_factor.item() - 1.25) < 1e-6 and torch.equal(ex.objects(), pose)
        ex.start(logw_l.t().contiguous())                               # reuse with another shape, no scalars
        ok = ok and torch.equal(ex.objects().t(), logw)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('dof', [6, 4])
def test_object_sharding_all_gather_gloo(dof):
    import conftest
    conftest._emu_lib()     # build once in the parent
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), dof, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_range_covers_everything():
    from epropnp import sharding
    for n in (0, 1, 7, 600, 4096):
        for w in (1, 2, 3, 8):
            edges = [sharding.shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            assert max(e[1] - e[0] for e in edges) - min(e[1] - e[0] for e in edges) <= 1


def _bench_worker(rank, world, port, ret, config='C4', objects='9', stub=None):
    """bench.py's own main() as one of two gloo ranks: the emulation build is installed HERE, from the outside; bench.py gets
    a CPU device and the gloo backend through its test hook and skips only what needs the GPU (hipGraph, HIP events, roofline)."""
    import contextlib
    import io
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if stub:
        os.environ['BENCH_RCCL_LIB'] = stub
    import conftest
    import install as emu
    emu.install(conftest._emu_lib())
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main(['--gpus', str(world), '--config', config, '--objects', objects, '--points', '24', '--samples', '16', '--amis-iters', '2',
                         '--lm-iters', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-hipgraph'], device='cpu', backend='gloo')
    ret[rank] = (rc, buf.getvalue(), dist.is_initialized())


def test_bench_strong_scaling_step_two_gloo_ranks():
    """bench.py --config C4 (ONE batch split over the ranks, ONE collective per step, detection loss with the world-mean
    norm_factor) with TWO real ranks: 9 objects -> 5 + 4 (uneven tail), the exchange inside the timed region, the per-rank
    evidence, `gathered == local` bit for bit, the replayed-step check, and a clean teardown (process group destroyed)."""
    import json
    import conftest
    conftest._emu_lib()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bench_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        ret = dict(ret)
    assert ret[0][0] == 0 and ret[1][0] == 0 and not ret[0][2] and not ret[1][2]        # both exited clean, groups destroyed
    assert ret[1][1] == ''                                                              # only rank 0 prints
    line = json.loads([ln for ln in ret[0][1].splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] > 0
    assert line['config'] == {'name': 'C4', 'objects_per_gpu': 5, 'objects_total': 9}
    rk = line['ranks']
    assert rk['process_group'] == 'gloo' and rk['rccl_world_size'] == 2 and rk['all_reduce_of_ones'] == 2.0
    assert len(rk['ms_per_step_per_rank']) == 2 and line['ms_per_step'] >= rk['ms_per_step_max'] - 1e-3      # max over ranks
    c = line['collective']
    assert c['bytes_per_rank'] == 9 * 4 * 4 + 4 and c['gathered_equals_local_bitwise'] is True and 'gloo' in c['route']
    chk = c['replayed_step_check']
    assert abs(chk['norm_factor_input_last_step'] - chk['evaluated_eagerly']) <= 1e-5 * abs(chk['evaluated_eagerly'])


@pytest.mark.parametrize('config,objects', [('C4', '21'), ('C2', '2'), ('C5', '2')])
def test_bench_line_with_eight_ranks(tmp_path, config, objects):
    """`bench.py --gpus 8` as the driver launches it at round end, here with EIGHT gloo ranks on the emulation build: the `ranks` block of
    the JSON line for N = 8 (group size, all-reduce of ones, eight per-rank step times, max over ranks), weak configs C2 / C5 with
    disjoint shards and no collective, and the strong Det step C4 -- 21 objects split 3,3,3,3,3,2,2,2 -- with its ONE collective
    on the DIRECT RCCL route (ncclAllGather of the shared-memory stub through the same ctypes binding as on an 8-GPU node)."""
    import json
    import conftest
    conftest._emu_lib()
    stub = _build_rccl_stub(tmp_path) if config == 'C4' else None
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bench_worker, args=(8, _free_port(), ret, config, objects, stub), nprocs=8, join=True)
        ret = dict(ret)
    assert all(ret[r][0] == 0 and not ret[r][2] for r in range(8)) and all(ret[r][1] == '' for r in range(1, 8))
    line = json.loads([ln for ln in ret[0][1].splitlines() if ln.startswith('{')][-1])
    rk = line['ranks']
    assert line['n_gpus'] == 8 and rk['rccl_world_size'] == 8 and rk['all_reduce_of_ones'] == 8.0 and len(rk['ms_per_step_per_rank']) == 8
    assert sorted(rk['devices']) == list(range(8)) and line['ms_per_step'] >= rk['ms_per_step_max'] - 1e-3
    if config == 'C4':
        assert line['scaling'] == 'strong' and line['config'] == {'name': 'C4', 'objects_per_gpu': 3, 'objects_total': 21}
        c = line['collective']
        assert c['route'].startswith('rccl ncclAllGather') and c['gathered_equals_local_bitwise'] is True and c['bytes_per_rank'] == 21 * 4 * 4 + 4
        chk = c['replayed_step_check']
        assert abs(chk['norm_factor_input_last_step'] - chk['evaluated_eagerly']) <= 1e-5 * abs(chk['evaluated_eagerly'])
    else:
        assert line['scaling'] == 'weak' and 'collective' not in line and line['config']['objects_total'] == 16
        assert abs(line['value'] - 16 / (line['ms_per_step'] * 1e-3)) <= 0.01 * line['value']       # whole-job aggregate over the 8 ranks


def _nccl_worker(rank, world, port, ret):
    """Object sharding over RCCL on real GPUs: each rank solves its contiguous shard of ONE batch, pose outputs are
    all-gathered, the result equals the single-GPU run (objects are independent; the gather only moves data)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        from epropnp import sharding
        from epropnp.epropnp import EProPnP4DoF
        from epropnp.levenberg_marquardt import LMSolver
        B, N, S, K = 37, 64, 64, 4                       # uneven tail
        prob = orc.make_problem(B, N, 4, seed=40, bounds='tensor')
        noise = pack_noise(orc.make_noise(B, S, K, 4, seed=41), 4)

        def run(lo, hi):
            sub = {k: (v[lo:hi] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == B else v) for k, v in prob.items()}
            p, cam, cf = make_layer_objects(sub, dev, relative_delta=0.5)
            cf.set_param(p['x2d'], p['w2d'])
            layer = EProPnP4DoF(mc_samples=S, num_iter=K, normalize=True, solver=LMSolver(dof=4, num_iter=5))
            out = layer.monte_carlo_forward(p['x3d'], p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'],
                                            force_init_solve=False, noise=noise[lo:hi].contiguous().to(dev))
            return out[0], out[4]
        lo, hi = sharding.shard_range(B)
        pose_l, logw_l = run(lo, hi)
        pose = sharding.gather_objects(pose_l, B, obj_dim=0, force_collective=True)
        logw = sharding.gather_objects(logw_l, B, obj_dim=1, force_collective=True)
        pose_full, logw_full = run(0, B)
        # launch shapes depend on the batch size, so shard vs full batch agree to rounding, not bit for bit
        ret[rank] = bool(pose.shape == (B, 4) and logw.shape == (S, B) and (pose - pose_full).abs().max().item() < 1e-5
                         and (torch.logsumexp(logw, 0) - torch.logsumexp(logw_full, 0)).abs().max().item() < 1e-3)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_object_sharding_all_gather_nccl():
    """RCCL `all_gather_into_tensor` of the pose outputs: 2 ranks when the box has two GPUs, otherwise the same code path
    (process group on the nccl backend, forced collective) with a single rank."""
    import install as emu
    emu.uninstall()
    world = 2 if torch.cuda.device_count() >= 2 else 1
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_nccl_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['direct', 'c10d'])
def test_bench_det_route_switch_and_clean_teardown(route):
    """`bench.py --config C4 --route direct|c10d` under the self-launcher: the line names the route that ran, and the
    teardown record (BENCH_REPORT_TEARDOWN=1, stderr) shows every direct RCCL communicator destroyed before the process
    group -- an undestroyed communicator is an exit-time hang risk that only shows with several ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 2 if torch.cuda.device_count() >= 2 else 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', BENCH_SELF_LAUNCH='1', BENCH_REPORT_TEARDOWN='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1', '--config', 'C4',
           '--route', route, '--no-cpu-baseline', '--no-hipgraph']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:       # one retry for a rendezvous that did not come up (seen once on a box that took 75 s to receive the tree)
        sys.stderr.write('bench.py self-launch failed once, retrying:\n' + r.stderr[-1500:] + '\n')
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    want = 'rccl ncclAllGather' if route == 'direct' else 'torch.distributed.all_gather_into_tensor'
    assert line['collective']['route'].startswith(want), line['collective']['route']
    td = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith('{"teardown"')][-1])['teardown']
    assert td == {'rccl_comms_closed': True, 'process_group_destroyed': True}


@pytest.mark.gpu
@pytest.mark.parametrize('config', ['C1', 'C3', 'C3-train'])
def test_bench_reference_caller_configs(config):
    """BASELINE.json configs[0] / configs[2] are reachable from the driver's command: `python bench.py --config C1 | C3 |
    C3-train` (hipGraph replay by default, the eagerly launched step of the same run beside it)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', config, '--steps', '10', '--warmup', '2'],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['config']['name'] == config and line['value'] > 0 and line['n_gpus'] == 1 and line['launch_note'] is None
    assert line['launch'].startswith('hipGraph') and line['eager']['ms_per_step'] > 0
    assert line['kernel_ms']['amis_forward'] > 0 and line['kernel_ms']['amis_backward'] > 0
    objects = {'C1': 1, 'C3': 32, 'C3-train': 32}[config]
    assert line['config']['objects_per_gpu'] == objects and abs(line['value'] - objects / (line['ms_per_step'] * 1e-3)) < 0.01 * line['value']
    if config != 'C3':
        assert line['kernel_ms']['rslm_solve'] > 0


@pytest.mark.gpu
@pytest.mark.parametrize('form', ['torchrun', 'plain'])
@pytest.mark.parametrize('config', ['C2', 'C4', 'C5'])
def test_bench_multi_gpu_harness(config, form):
    """The multi-GPU bench command in both forms -- `python -m torch.distributed.run ... bench.py --gpus N` and the plain
    `python bench.py --gpus N`, which spawns its N ranks itself (the launch model of
    EPro-PnP-Det/tools/train.py:123 replaced) -- on the nccl backend: 2 ranks when the box has two GPUs, one rank (through
    the same launcher route, BENCH_SELF_LAUNCH=1) otherwise.  C4 = 600 objects split over the ranks + ONE
    all_gather_into_tensor per step inside the timed region, C2 / C5 = disjoint shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 2 if torch.cuda.device_count() >= 2 else 1
    extra = {'C5': ['--objects', '512'], 'C2': ['--objects', '256'], 'C4': []}[config]     # slices keep the test short
    tail = [os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1', '--config', config,
            '--no-cpu-baseline', '--no-hipgraph'] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    if form == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
               '127.0.0.1', '--master-port', str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
        env['BENCH_SELF_LAUNCH'] = '1'
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == n and line['config']['name'] == config and line['value'] > 0
    rk = line['ranks']
    assert rk['launcher'] == 'torch.distributed.run' and rk['process_group'] == 'nccl'
    assert rk['rccl_world_size'] == n and rk['all_reduce_of_ones'] == float(n) and sorted(rk['devices']) == list(range(n))
    assert len(rk['ms_per_step_per_rank']) == n and rk['ms_per_step_min'] <= rk['ms_per_step_max']
    assert line['roofline']['achieved'] > 0 and 'IC-cold' in line['roofline']['cache_state']
    if config == 'C4':
        assert line['scaling'] == 'strong' and line['config']['objects_total'] == 600
        assert line['collective']['backend'].startswith('nccl') and line['collective']['bytes_per_rank'] == 600 * 4 * 4 + 4
        assert line['collective']['gathered_equals_local_bitwise'] is True
    else:
        assert line['scaling'] == 'weak' and 'collective' not in line


# ---- the direct-RCCL route of ObjectExchange with MORE THAN ONE rank ------------------------------------------
# A one-GPU box can only ever run RcclComm with one rank, and gloo takes the c10d branch: the branch an 8-GPU node takes was never
# reached by a multi-rank test.  tests/stubs/rccl_stub.c exports RCCL's five entry points over a shared-memory segment, and the same
# ctypes path binds it here: 2 and 8 ranks, an uneven split, the collective agreement when ONE rank's ncclCommInitRank fails, teardown.
def _build_rccl_stub(tmp):
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(str(tmp), 'librccl_stub.so')
    subprocess.check_call(['gcc', '-O1', '-shared', '-fPIC', os.path.join(root, 'stubs', 'rccl_stub.c'), '-o', out, '-lrt'])
    return out


def _direct_worker(rank, world, port, stub, num_obj, fail_rank, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RCCL_STUB_TIMEOUT_MS'] = '1500'
    if fail_rank is not None:
        os.environ['RCCL_STUB_FAIL_RANK'] = str(fail_rank)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import warnings
        from epropnp import sharding
        full = torch.arange(num_obj * 4, dtype=torch.float32).reshape(num_obj, 4) * 0.5 + 1.0
        lo, hi = sharding.shard_range(num_obj)
        ex = sharding.ObjectExchange(num_obj, direct=True, rccl_lib=stub)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            ex.start(full[lo:hi].clone(), torch.tensor(float(rank + 1)))
        ok = torch.equal(ex.objects(), full)                                        # every rank holds the whole batch, in order
        ok = ok and abs(float(ex.world_mean()) - (world + 1) / 2.0) < 1e-6        # mean of 1 .. world
        ok = ok and ex.scalar_slots().shape == (world,) and torch.equal(ex.scalar_slots(), torch.arange(1, world + 1, dtype=torch.float32))
        if fail_rank is None:
            ok = ok and ex.route.startswith('rccl') and not caught and len(sharding.RcclComm._live) == 1
            ex.start(full[lo:hi, :2].contiguous())                                  # reuse, another row shape, no scalars
            ok = ok and torch.equal(ex.objects(), full[:, :2])
        else:       # ONE rank's init failed: EVERY rank fell back to c10d together (and said so), nobody hangs
            ok = ok and ex.route.startswith('torch.distributed') and len(caught) == 1 and 'direct RCCL communicator unavailable' in str(caught[0].message)
            ok = ok and ex._comm is None and len(sharding.RcclComm._live) == 0
        ex.close()
        ok = ok and ex._comm is None and len(sharding.RcclComm._live) == 0           # ncclCommDestroy on every rank
        ret[rank] = bool(ok)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world,num_obj,fail_rank', [(2, 5, None), (8, 600, None), (8, 601, None), (2, 7, 1), (8, 600, 3)])
def test_direct_rccl_route_with_several_ranks(tmp_path, world, num_obj, fail_rank):
    stub = _build_rccl_stub(tmp_path)
    ret = mp.Manager().dict()
    mp.spawn(_direct_worker, args=(world, _free_port(), stub, num_obj, fail_rank, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world, dict(ret)


@pytest.mark.gpu
@pytest.mark.parametrize('form', ['torchrun', 'plain'])
def test_bench_refuses_more_ranks_than_devices(form):
    """`bench.py --gpus N` on a node that shows fewer than N devices leaves with ONE line and status 2 in both launch forms --
    before the rendezvous, so that no rank is left waiting for a peer that died (the first SCALE run on an 8-GPU node should
    yield a curve or this line, not a timeout)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count()
    n = have + 1
    tail = [os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1', '--config', 'C2', '--objects', '64',
            '--no-cpu-baseline', '--no-hipgraph']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    if form == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert 'visible HIP device' in r.stderr or 'HIP device(s) are visible' in r.stderr, r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


@pytest.mark.gpu
@pytest.mark.parametrize('visible', [None, '0'])
def test_real_rccl_world_of_one_under_device_masks(visible):
    """What the stub of tests/stubs/rccl_stub.c cannot stand in for: the REAL librccl.so behind sharding.RcclComm (ncclGetUniqueId,
    ncclCommInitRank, ncclAllGather on the caller's stream, ncclCommDestroy), with and without a HIP_VISIBLE_DEVICES mask, through
    bench.py's C4 step under the launcher -- the line must name the direct route, carry the rank's PCI address, and tear down clean."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', BENCH_SELF_LAUNCH='1', BENCH_REPORT_TEARDOWN='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    if visible is not None:
        env['HIP_VISIBLE_DEVICES'] = visible
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--config', 'C4',
           '--route', 'direct', '--no-cpu-baseline', '--no-hipgraph']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['collective']['route'].startswith('rccl ncclAllGather'), line['collective']['route']
    rk = line['ranks']
    assert rk['rccl_world_size'] == 1 and len(rk['pci_addresses']) == 1 and len(set(rk['pci_addresses'])) == 1
    td = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith('{"teardown"')][-1])['teardown']
    assert td == {'rccl_comms_closed': True, 'process_group_destroyed': True}

#!/usr/bin/env python
"""bench.py -- EPro-PnP hot path throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C3-train|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config ...]

Both forms work for N > 1: started without RANK / WORLD_SIZE in the environment, `--gpus N` re-launches itself as N ranks
(one process per GPU) under torch.distributed.run on 127.0.0.1 and passes the ranks' output through; the line carries
the evidence that RCCL saw N ranks (`ranks`: process-group size, an all-reduce of ones, every rank's device and
ms_per_step).

One "step" = one pass of the hot path over one batch of synthetic objects:
    cost(pose_init) -> [RSLM initialiser] -> fused LM solve (1+L sweeps) -> fused AMIS sampler (S pose evaluations) ->
    Monte-Carlo pose loss -> backward to d/dx3d, d/dx2d, d/dw2d (one recompute kernel + autograd through set_param / loss).

--config (BASELINE.json `configs`), one process per GPU, objects sharded over ranks:
  C2 (default, the headline metric): 4096 objects/GPU x N=512 points, S=512, K=4, L=3, 6-DoF.  Every rank owns its own
      objects, NO data-path collective: weak scaling.
  C5 (stress): 65 536 objects over 8 GPUs = 8192 objects/GPU x N=2048 x S=1024, disjoint shards, no collective: weak.
  C4 (EPro-PnP-Det nuScenes shape): ONE batch of 600 objects x N=128, 4-DoF, RSLM(16,64,3) + LM 5 + AMIS S=128/K=4,
      normalize=True, split contiguously over the ranks (75/GPU at 8: `sharding.shard_objects`), fwd+bwd on the shard,
      with ONE RCCL `all_gather_into_tensor` per step (`sharding.ObjectExchange`: the pose outputs and the detection
      loss's norm_factor scalar in the same payload) issued in stream order right after the forward -- INSIDE the timed
      region: strong scaling.  The line reports the collective's share of the step.
  C1 (demo/fit_identity.ipynb cells 5-10, the plumbing case): 1 object x N=64, identity camera, RSLM(8,128,5) + LM 10 +
      AMIS 512/4, with_pose_opt_plus and the notebook's derivative regularisation; launch-bound.
  C3 (LineMOD shape as BASELINE.json words it): 32 crops x 64 x 64 = 4096 DENSE correspondences, per-object tensor bounds,
      z_min 0.01, relative_delta 0.1, LM 5 + AMIS 512/4, fwd+bwd (lib/train.py:143-180 without the sub-sampling).
  C3-train (the call lib/train.py:177-179 makes): 32 crops x 512 sub-sampled correspondences, RSLM(16,4,3) + LM 5,
      force_init_solve, with_pose_opt_plus + the loop's translation / rotation regularisers.
      C1 / C3 / C3-train are launch-bound: like C4 they are replayed from a hipGraph by default (`--launch auto`), and the
      line carries the eagerly launched step time of the same run beside it (`eager`).
fp32, inputs resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `roofline` is the Jacobian sweep: the fused LM kernel credited one 28 B/point read per
logical sweep with its physical traffic beside it (`roofline.fused_lm`), and `roofline` itself =
normal_equations_kernel, where one logical sweep IS one physical read of the correspondences, timed after the step loop
IC-cold (launches rotate over distinct copies of the inputs, > 2 x the 256 MiB Infinity Cache).
`roofline_valu` reports the VALU-bound AMIS kernels against the fp32 vector peak.  `cpu_baseline` is the oracle (a
PyTorch-CPU restatement with the reference's op structure, pinned to the reference and timed beside it in
profiles/r02_cpu_reference_vs_oracle.txt) on a bounded sample of the same workload on the host cores.

Diagnostics behind environment variables, after the timed region, to stderr (never part of the line): BENCH_HOST_SEGMENTS=1 --
host time of an eagerly launched step by statement; BENCH_TORCH_PROFILE=1 -- torch.profiler's CPU-side op table of the same step
(tools/gpu_host_segments.sh, profiles/r06_eager_host_time.txt).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3    # MI355X_MICROARCH.md: peak FP32 vector


def synth_problem(B, N, device, seed, dof=6, cam='pinhole800'):
    """SURVEY.md 8(d) generator, on `device`: x3d ~ N(0,0.5^2); gt t ~ N(0,I), t_z += 5; q ~ normalised N(0,I4);
    K = [[800,0,320],[0,800,240],[0,0,1]]; x2d = project(gt) + N(0,1 px); w2d = softmax_N(U(0,1)) * 2;
    pose_init = gt perturbed (t += 0.1 N, q = normalize(q + 0.05 N)).  cam='identity': K = I (the demo notebook's
    normalised image plane), image noise 1/800."""
    from epropnp.camera import project_b
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    x3d = rn(B, N, 3) * 0.5
    t = rn(B, 3)
    t[:, 2] += 5 if dof == 6 else 10
    if dof == 6:
        q = torch.nn.functional.normalize(rn(B, 4), dim=-1)
        pose_gt = torch.cat((t, q), -1)
    else:
        pose_gt = torch.cat((t, rn(B, 1)), -1)
    if cam == 'identity':
        K, px = torch.eye(3, device=device).expand(B, 3, 3), 1.0 / 800.0
    else:
        K, px = torch.tensor([[800., 0, 320], [0, 800., 240], [0, 0, 1]], device=device).expand(B, 3, 3), 1.0
    x2d = project_b(x3d, pose_gt, K, 0.1)[0] + rn(B, N, 2) * px
    w2d = torch.softmax(torch.rand(B, N, 2, generator=g, device=device), dim=1) * 2.0
    if dof == 6:
        qi = torch.nn.functional.normalize(pose_gt[:, 3:] + 0.05 * rn(B, 4), dim=-1)
        pose_init = torch.cat((pose_gt[:, :3] + 0.1 * rn(B, 3), qi), -1)
    else:
        pose_init = torch.cat((pose_gt[:, :3] + 0.1 * rn(B, 3), pose_gt[:, 3:] + 0.05 * rn(B, 1)), -1)
    return dict(x3d=x3d.contiguous(), x2d=x2d.contiguous(), w2d=w2d.contiguous(), cam_mats=K, pose_init=pose_init,
                pose_gt=pose_gt)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(N, S, K, L, total_objects=1024, chunk=64, one_thread_objects=16):
    """The oracle (oracle/epropnp_oracle.py) timed on the host cores by BASELINE.md section 3's procedure: `total_objects`
    (>= 1024 at C2) objects of the same workload -- monte_carlo_forward + MC loss + backward -- in chunks of `chunk` <= 256
    objects (the reference keeps ~12 MB per object alive for its backward), one warm-up chunk, MEDIAN of 3 passes over all the
    chunks.  Test / baseline infrastructure -- never the product path.
    torch-CPU oversubscribes badly on many-core hosts (256 threads on these small ops is ~1000x slower than 16), so the headline
    `value` is the best of a few thread counts (chosen on one chunk, then the full procedure at that count), and the dict
    carries the 1-thread figure and the os.cpu_count()-thread figure beside it: the latter on a tiny sample in a bounded
    subprocess (it reports an upper bound when the run does not finish in time).  `reference_vs_port` quotes the
    build-container timing of the UNMODIFIED reference next to this oracle (oracle/time_reference_vs_oracle.py)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import epropnp_oracle as orc
    host_cores = os.cpu_count() or 1
    chunk = min(chunk, 256, total_objects)
    prob = synth_problem(total_objects, N, torch.device('cpu'), seed=0)
    noise = orc.make_noise(total_objects, S, K, 6, seed=1)

    def piece(lo, hi):
        return ({k: v[lo:hi].contiguous() for k, v in prob.items()}, {k: v[:, :, lo:hi].contiguous() for k, v in noise.items()})

    def one_pass(objects, size):
        t0 = time.perf_counter()
        for lo in range(0, objects, size):
            pp, nn = piece(lo, min(objects, lo + size))
            orc.run_mc(pp, nn, 6, S, K, L)
        return time.perf_counter() - t0
    # thread count: one chunk each (1 warm-up + 2 timed), the best goes through the full procedure
    tried = []
    for threads in sorted({min(host_cores, t) for t in (8, 16, 32)}):
        torch.set_num_threads(threads)
        ts = [one_pass(chunk, chunk) for _ in range(3)]
        tried.append((threads, round(chunk / min(ts[1:]), 1)))
    best_threads = max(tried, key=lambda t: t[1])[0]
    torch.set_num_threads(best_threads)
    one_pass(chunk, chunk)                                       # warm-up at the chosen thread count
    passes = [one_pass(total_objects, chunk) for _ in range(3)]
    value = total_objects / _median(passes)
    # one thread: BASELINE.md's 1-thread figure, same procedure on `one_thread_objects` objects
    torch.set_num_threads(1)
    k1 = min(one_thread_objects, total_objects)
    one_pass(min(4, k1), min(4, k1))
    t1 = [one_pass(k1, k1) for _ in range(3)]
    one_thread = {'value': round(k1 / _median(t1), 2), 'unit': 'instances/s', 'cores': 1,
                  'sample': f'{k1} objects in one chunk, median of 3 after a warm-up'}
    torch.set_num_threads(best_threads)
    # every core (torch.set_num_threads(os.cpu_count())): in a subprocess with a wall-clock bound
    all_cores = {'value': round(value, 2), 'unit': 'instances/s', 'cores': host_cores, 'sample': 'the headline run (best thread count = all cores)'}
    if host_cores != best_threads:
        import subprocess
        n_all, bound_s = 4, float(os.environ.get('BENCH_CPU_ALL_CORES_BOUND_S', '25'))
        code = (f'import sys, time, json, torch; sys.path[:0] = [{os.path.join(ROOT, "oracle")!r}, {ROOT!r}]\n'
                f'import bench, epropnp_oracle as orc\n'
                f'torch.set_num_threads({host_cores})\n'
                f'p = bench.synth_problem({n_all}, {N}, torch.device("cpu"), seed=0); n = orc.make_noise({n_all}, {S}, {K}, 6, seed=1)\n'
                f't0 = time.perf_counter(); orc.run_mc(p, n, 6, {S}, {K}, {L}); print(json.dumps(time.perf_counter() - t0))\n')
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=bound_s)
            t_all = float(r.stdout.strip().splitlines()[-1])
            all_cores = {'value': round(n_all / t_all, 3), 'unit': 'instances/s', 'cores': host_cores,
                         'sample': f'{n_all} objects, ONE run without warm-up in a fresh process (thread oversubscription: ~1e5 small ops each waking {host_cores} threads)'}
        except Exception as e:      # timed out (or failed): an upper bound is what was measured
            all_cores = {'value': None, 'upper_bound': round(n_all / max(time.perf_counter() - t0, 1e-9), 3), 'unit': 'instances/s', 'cores': host_cores,
                         'sample': f'{n_all} objects did not finish within {bound_s:.0f} s ({type(e).__name__}); BENCH_CPU_ALL_CORES_BOUND_S raises the bound'}
    ref = None
    try:
        ref = json.load(open(os.path.join(ROOT, 'profiles', 'r05_cpu_reference_vs_oracle.json')))
        ref['source'] = 'profiles/r05_cpu_reference_vs_oracle.json'
    except (OSError, ValueError):
        pass
    port = dict(value=round(value, 2), unit='instances/s', cores=best_threads, kind='port',
                pass_seconds=[round(t, 3) for t in passes])
    out = dict(port, host_cores=host_cores,
               procedure=f'BASELINE.md section 3: {total_objects} objects in chunks of {chunk}, 1 warm-up chunk, median of 3 passes',
               tried_threads_inst_per_s_one_chunk=tried, one_thread=one_thread, all_cores=all_cores, reference_vs_port=ref,
               sample=f'{total_objects} objects x N={N}, S={S}, K={K}, L={L} (fwd+bwd; oracle = PyTorch-CPU restatement with the '
                      f'reference op structure) in chunks of {chunk} objects on {best_threads} threads; larger CPU chunks are slower per object (256: ~100/s)')
    # north_star: "the reference's own PyTorch-CPU solver timed on the host cores".  Where a reference checkout is reachable
    # ($EPROPNP_REFERENCE, default /root/reference -- the build container; it does not travel to the GPU box) the UNMODIFIED modules
    # are timed by the same procedure in a process of their own (oracle/cpu_baseline_runner.py) and become the headline: kind
    # "reference", with the port's figure beside it.
    timed_ref = reference_cpu_baseline(N, S, K, L, total_objects, chunk, best_threads)
    if timed_ref is not None:
        out.update(value=timed_ref['value'], kind='reference', pass_seconds=timed_ref['pass_seconds'], port=port,
                   sample=f'{total_objects} objects x N={N}, S={S}, K={K}, L={L} (fwd+bwd) through the UNMODIFIED reference modules '
                          f'({timed_ref["where"]}; epropnp/epropnp.py:87-196) in chunks of {chunk} objects on {best_threads} threads')
    return out


def reference_cpu_baseline(N, S, K, L, total_objects, chunk, threads, bound_s=None):
    """The unmodified reference timed by oracle/cpu_baseline_runner.py in a subprocess, or None when no checkout is reachable (the
    GPU box) or the run does not finish within the bound (BENCH_CPU_REFERENCE_BOUND_S, default 240 s)."""
    import subprocess
    root = os.environ.get('EPROPNP_REFERENCE', '/root/reference')
    if not os.path.isdir(os.path.join(root, 'epropnp')):
        return None
    bound_s = bound_s or float(os.environ.get('BENCH_CPU_REFERENCE_BOUND_S', '240'))
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'cpu_baseline_runner.py'), 'reference', str(N), str(S), str(K), str(L),
                            str(total_objects), str(chunk), str(threads)], capture_output=True, text=True, timeout=bound_s)
        res = json.loads(r.stdout.strip().splitlines()[-1])
        if r.returncode != 0 or 'value' not in res:
            return None
        res['where'] = root
        return res
    except Exception:
        return None


def hipgraph_replay():
    """The C2 step captured once into a hipGraph and replayed (tools/graph_step.py), in its own process and outside the
    timed region: `value` above is the eager step, whose per-kernel HIP events cannot sit inside a graph."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'graph_step.py'), 'C2'], capture_output=True,
                           text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
        d = json.loads(line)
        return {'ms_per_step': d['graph_ms'], 'value': round(d['objects'] / (d['graph_ms'] * 1e-3), 1),
                'unit': 'instances/s', 'eager_ms_same_process': d['eager_ms'],
                'fresh_samples_per_replay': d['fresh_samples_per_replay']}
    except Exception as e:      # never let the extra measurement break the contract line
        return {'error': str(e)[:200]}


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: become the launcher.  Re-executes this very command line as N ranks
    under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port), passes stdout / stderr
    through and returns its exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus:
        sys.stderr.write(f'bench.py: --gpus {n_gpus} but only {have} HIP device(s) are visible\n')
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


SPINUP_STEPS = 40    # untimed steps before the W warm-up steps (device transient, see main()); reported in the JSON line

CONFIGS = {
    # name: objects (per GPU for weak configs / total for the strong one), points, samples, AMIS iters, LM iters, dof
    'C2': dict(objects=4096, points=512, samples=512, amis_iters=4, lm_iters=3, dof=6, scaling='weak'),
    'C5': dict(objects=8192, points=2048, samples=1024, amis_iters=4, lm_iters=3, dof=6, scaling='weak'),
    'C4': dict(objects=600, points=128, samples=128, amis_iters=4, lm_iters=5, dof=4, scaling='strong'),
    # BASELINE.json configs[0] / configs[2]: the reference's own callers (demo notebook, EPro-PnP-6DoF lib/train.py)
    'C1': dict(objects=1, points=64, samples=512, amis_iters=4, lm_iters=10, dof=6, scaling='weak'),
    'C3': dict(objects=32, points=4096, samples=512, amis_iters=4, lm_iters=5, dof=6, scaling='weak'),
    'C3-train': dict(objects=32, points=512, samples=512, amis_iters=4, lm_iters=5, dof=6, scaling='weak'),
}
# what the callers of those configurations set besides the sizes (demo/fit_identity.ipynb cell 5; lib/train.py:47-57,168-179)
CALLER = {
    'C1': dict(cam='identity', rslm=(8, 128, 5), relative_delta=0.5, z_min=0.1, crop_bounds=False, plus=True),
    'C3': dict(cam='pinhole800', rslm=None, relative_delta=0.1, z_min=0.01, crop_bounds=True, plus=False),
    'C3-train': dict(cam='pinhole800', rslm=(16, 4, 3), relative_delta=0.1, z_min=0.01, crop_bounds=True, plus=True),
}
LAUNCH_BOUND = ('C4', 'C1', 'C3', 'C3-train')       # `--launch auto` replays these from a hipGraph


def measured_traffic(kernel, shape_key):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r03_pmc_traffic.json, written by
    tools/pmc_traffic.py from separate --pmc runs; gfx950 corrections applied there) -- None when no profile of this
    exact shape is on file, so a stale number can never be attached to a changed workload."""
    # newest committed profile of this shape.  (r04_pmc_traffic.json is not consulted: its C2 `normal_equations_kernel` entry averages
    # 121 C2-size with 27 C5-size dispatches -- tools/pmc_traffic.py grouped by kernel name only then; it groups by launch size now.)
    for name in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
        try:
            rec = json.load(open(os.path.join(ROOT, 'profiles', name))).get(shape_key, {}).get(kernel)
        except (OSError, ValueError):
            continue
        if rec:
            return rec['hbm_bytes_per_launch'], 'profiles/' + name
    return None, None


IC_BYTES = 256 * 2 ** 20       # MI355X Infinity Cache (MI355X_MICROARCH.md); FETCH_SIZE counts its hits as fetches
MAX_SWEEP_SETS = 256           # tiny workloads (C1: 2 KB per set) cannot be rotated out of the cache: the line says IC-warm
LARGE_SWEEP = (8192, 2048)     # `roofline.large`: the C5-shard Jacobian sweep (470 MB per launch: no cache can serve it)


class ClockSampler:
    """Engine clock, memory clock and package power of THE device under test, read from the amdgpu hwmon nodes
    (/sys/class/drm/card*/device/hwmon/hwmon*/{freq1_input, freq2_input, power1_input}: Hz, Hz, microwatt) by a background thread
    while a timed window runs -- so that an HBM figure that moves from box to box can be read next to the clocks it was taken at
    (VERDICT r05 item 7).  The card is matched by PCI address; where sysfs is not readable the record says so."""

    def __init__(self, device_index=0, period_s=0.0005):
        import glob
        self.period_s, self.samples, self.dir, self.note = period_s, [], None, None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
            for d in sorted(glob.glob('/sys/class/drm/card*/device')):
                if os.path.basename(os.path.realpath(d)) == want:
                    hw = sorted(glob.glob(os.path.join(d, 'hwmon', 'hwmon*')))
                    if hw and os.path.exists(os.path.join(hw[0], 'freq1_input')):
                        self.dir, self.pci = hw[0], want
            if self.dir is None:
                self.note = f'no amdgpu hwmon node for PCI device {want}'
        except Exception as e:      # no sysfs, no such attribute: the record says so
            self.note = f'{type(e).__name__}: {e}'

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def __enter__(self):
        import threading
        self._stop = threading.Event()
        if self.dir is not None:
            def run():
                while not self._stop.is_set():
                    self.samples.append((self._read('freq1_input'), self._read('freq2_input'), self._read('power1_input')))
                    self._stop.wait(self.period_s)
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.dir is not None:
            self._thread.join()

    def summary(self):
        if self.dir is None:
            return {'available': False, 'note': self.note}

        def stat(i, scale):
            v = [s[i] / scale for s in self.samples if s[i] is not None]
            return None if not v else {'min': round(min(v), 1), 'mean': round(sum(v) / len(v), 1), 'max': round(max(v), 1)}
        return {'available': True, 'source': f'{self.dir} (PCI {self.pci})', 'samples': len(self.samples),
                'sclk_mhz': stat(0, 1e6), 'mclk_mhz': stat(1, 1e6), 'power_w': stat(2, 1e6)}


def single_sweep(F, make_problem, pose, bytes_per_set, windows=32):
    """normal_equations_kernel alone (one logical sweep = one physical read of the correspondences), IC-COLD: the launches
    rotate over `sets` distinct copies of the problem buffers, > 2x the 256 MiB Infinity Cache in total, so that a set has
    been evicted long before it comes round again and every read is served by HBM (back-to-back launches on ONE 59 MB set
    would be served by the Infinity Cache).  HIP events on the launch stream around windows of one full rotation
    -> (mean, median) ms per launch, number of sets."""
    sets = min(MAX_SWEEP_SETS, max(2, -(-2 * IC_BYTES // int(bytes_per_set)) + 1))
    probs = [make_problem() for _ in range(sets)]
    for _ in range(3):          # untimed rotations: first touch of the fresh copies (page tables), clocks back up after the
        for hp in probs:        # host-side pause that follows the step loop
            F.normal_equations(hp, pose)
    evs = []
    with ClockSampler() as clk:
        for _ in range(windows):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for hp in probs:
                F.normal_equations(hp, pose)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) / sets for a, b in evs)
    LAST_SWEEP_CLOCKS[0] = clk.summary()
    return sum(ts) / len(ts), ts[len(ts) // 2], sets


LAST_SWEEP_CLOCKS = [None]      # clocks / power sampled during the most recent single_sweep (bench line: roofline.clocks)


def main(argv=None, device=None, backend='nccl'):
    """`device` / `backend`: test hook (tests/test_distributed.py drives this very function with two gloo ranks on the CPU
    emulation of the kernels, which the TEST installs from the outside): a non-HIP device skips everything that needs the
    GPU -- hipGraph capture, HIP events, the roofline legs -- and prints a reduced line.  The driver's command never sets it."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='C2', help='BASELINE.json configuration (default: C2)')
    ap.add_argument('--objects', type=int, default=None, help='objects per GPU (C2, C5) / in the whole batch (C4)')
    ap.add_argument('--points', type=int, default=None)
    ap.add_argument('--samples', type=int, default=None)
    ap.add_argument('--amis-iters', type=int, default=None)
    ap.add_argument('--lm-iters', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='time the cpu_baseline leg of the chosen config and print it; needs no GPU (kind "reference" where a checkout is reachable)')
    ap.add_argument('--no-hipgraph', action='store_true', help='skip the informational hipGraph replay measurement')
    ap.add_argument('--cpu-sample', type=int, default=1024, help='objects of the workload the cpu_baseline leg times (in chunks of 64)')
    ap.add_argument('--no-large-sweep', action='store_true',
                    help="skip `roofline.large` (the C5-size Jacobian sweep): the PMC passes of the C2 workload use it so that every "
                         "normal_equations_kernel dispatch of the process has the C2 launch size")
    ap.add_argument('--launch', choices=['auto', 'eager', 'graph'], default='auto',
                    help="'graph': the rank's whole step (RCCL exchange included) captured once into a hipGraph; the timed "
                         "region replays it (fresh samples per replay).  'auto' (default): graph for the launch-bound Det "
                         "step (C4: 0.2 ms of kernels behind ~25 launches), eager for the GPU-bound C2 / C5")
    ap.add_argument('--route', choices=['direct', 'c10d'], default='direct',
                    help="C4's collective: 'direct' = RCCL's ncclAllGather called on the step's own stream (sharding.RcclComm), "
                         "'c10d' = torch.distributed.all_gather_into_tensor -- to bisect a failing direct route from the command line")
    args = ap.parse_args(argv)
    cfg = dict(CONFIGS[args.config])
    for k in ('objects', 'points', 'samples', 'amis_iters', 'lm_iters'):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    default_shape = all(cfg[k] == CONFIGS[args.config][k] for k in cfg)
    if args.cpu_baseline_only:       # the host-core baseline alone (no device touched): kind "reference" where a checkout is reachable
        Nc, Sc, Kc, Lc = cfg['points'], cfg['samples'], cfg['amis_iters'], cfg['lm_iters']
        res = (cpu_baseline(Nc, Sc, Kc, Lc, args.cpu_sample) if args.config != 'C5'
               else cpu_baseline(Nc, Sc, Kc, Lc, total_objects=8, chunk=8, one_thread_objects=2))
        print(json.dumps({'cpu_baseline': res, 'config': args.config}))
        return None

    # plain `python bench.py --gpus N`: spawn the N ranks ourselves (BENCH_SELF_LAUNCH=1 takes this route for N = 1 too)
    if (args.gpus > 1 or os.environ.get('BENCH_SELF_LAUNCH') == '1') and 'RANK' not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    on_gpu = device is None
    if on_gpu:
        assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU fallback)'
        have, local_world = torch.cuda.device_count(), int(os.environ.get('LOCAL_WORLD_SIZE', '1'))
        if have <= local_rank or have < local_world:
            # a launcher asked for more ranks than this node shows devices: say so in one line and leave BEFORE the rendezvous
            # (a rank that dies inside init_process_group leaves its peers waiting for the timeout)
            sys.stderr.write(f'bench.py: rank {rank} (local rank {local_rank} of {local_world}) finds {have} visible HIP device(s): '
                             f'--gpus {args.gpus} needs one device per rank (check HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)\n')
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    else:
        dev = torch.device(device)
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    dist = None
    if world > 1 or 'RANK' in os.environ:          # under torchrun also with one rank: the RCCL path is then exercised
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if on_gpu:
            dist.init_process_group(backend, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1 and args.gpus == 1, f'WORLD_SIZE={world} but --gpus {args.gpus}'

    from epropnp import functional as F
    from epropnp import sharding
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    from epropnp.losses import MonteCarloPoseLoss, monte_carlo_pose_loss

    N, S, K, L, dof = cfg['points'], cfg['samples'], cfg['amis_iters'], cfg['lm_iters'], cfg['dof']
    caller = CALLER.get(args.config)
    strong = cfg['scaling'] == 'strong'
    if strong:      # ONE batch (the same on every rank), split contiguously over the ranks
        total = cfg['objects']
        full = synth_problem(total, N, dev, seed=1000, dof=dof)
        lo, hi = sharding.shard_range(total, rank, world)
        prob = {k: v[lo:hi].contiguous() for k, v in full.items()}
        B = hi - lo
    else:           # every rank owns its own shard of objects
        B = cfg['objects']
        total = B * world
        prob = synth_problem(B, N, dev, seed=1000 + rank, dof=dof, cam=caller['cam'] if caller else 'pinhole800')
    x3d, x2d, w2d = (prob[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cost_fun = AdaptiveHuberPnPCost(relative_delta=caller['relative_delta'] if caller else 0.5)
    pose_target, with_plus = prob['pose_init'], False
    if caller:
        lb = ub = None
        if caller['crop_bounds']:       # lib/train.py:168-173: the crop box -/+ 30 output pixels, per object
            lo_, hi_ = x2d.detach().amin(1), x2d.detach().amax(1)
            unit = (hi_ - lo_).amax(-1, keepdim=True) / 64.0
            lb, ub = (lo_ - 30 * unit).contiguous(), (hi_ + 30 * unit).contiguous()
        camera = PerspectiveCamera(cam_mats=prob['cam_mats'], z_min=caller['z_min'], lb=lb, ub=ub)
        init = None
        if caller['rslm']:
            init = RSLMSolver(dof=6, num_points=caller['rslm'][0], num_proposals=caller['rslm'][1], num_iter=caller['rslm'][2])
            pose_target = prob['pose_gt']                   # pose_init = the ground truth, force_init_solve=True
        layer = EProPnP6DoF(mc_samples=S, num_iter=K, solver=LMSolver(dof=6, num_iter=L, init_solver=init), seed=1 + rank)
        # lib/train.py:182-183: the 6-DoF repo's loss MODULE with `scale.detach().mean()` as its norm_factor input (the notebook,
        # C1, defines its own loss class in cell 8 -- plain PyTorch statements -- and keeps the functional form below)
        loss_mod = MonteCarloPoseLoss(momentum=0.01).to(dev) if args.config in ('C3', 'C3-train') else None
        scale_net = torch.full((B, 2), 2.0, device=dev)               # stand-in for the network's `scale` output
        force_init, with_plus = init is not None, caller['plus']
    elif args.config == 'C4':
        camera = PerspectiveCamera(z_min=0.1, allowed_border=200)
        camera.set_param(prob['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
        init = RSLMSolver(dof=4, num_points=16, num_proposals=64, num_iter=3)
        layer = EProPnP4DoF(mc_samples=S, num_iter=K, normalize=True, seed=1 + rank,
                            solver=LMSolver(dof=4, num_iter=L, init_solver=init))
        loss_mod = MonteCarloPoseLoss(momentum=0.01).to(dev)           # training mode: world-mean of norm_factor
        obj_weight = torch.ones(B, device=dev)                         # `sample_weights` of the Det head
        scale_det = torch.full((B, 2), 2.0, device=dev)                # its `scale` (n_obj, 2): the weight scale the head predicts
        force_init = True
    else:
        camera = PerspectiveCamera(cam_mats=prob['cam_mats'], z_min=0.1)
        layer = EProPnP6DoF(mc_samples=S, num_iter=K, solver=LMSolver(dof=6, num_iter=L), seed=1 + rank)
        loss_mod, force_init = None, False

    from epropnp import _hip
    # per-kernel times: HIP events on the launch stream, recorded INSIDE the library around each kernel stage
    # (epropnp_profile_*): the forward is one host call (epropnp_monte_carlo_forward), its stages are not visible from here
    coll_events = []
    gathered, layer_last_pose = {}, {}
    # C4: the step's ONE collective (pose outputs + the loss's norm_factor scalar in one payload, sharding.ObjectExchange)
    # (BENCH_RCCL_LIB: test hook -- the RCCL library the direct route binds; tests/test_distributed.py runs this step with 8 ranks on
    # the shared-memory stub of tests/stubs/rccl_stub.c, where no 8-GPU node is at hand)
    exchange = sharding.ObjectExchange(total, force_collective=dist is not None, direct=args.route == 'direct',
                                       rccl_lib=os.environ.get('BENCH_RCCL_LIB') or None) if strong else None
    nf_scale = 1.0 / max(2 * B, 1)

    def norm_factor_input():
        # the Det head's norm_factor input, deform_pnp_head.py:870, as the head writes it (three ATen launches).  The timed step
        # hands `scale` and `sample_weights` to the exchange instead, whose pack launch evaluates the same expression
        # (sharding.ObjectExchange.start); this eager form is what `replayed_step_check` compares the replayed value with.
        return (scale_det * obj_weight[:, None]).sum() * nf_scale

    # BENCH_HOST_SEGMENTS=1: host time of an eagerly launched step by segment (time.perf_counter between the statements below,
    # no synchronisation), printed to stderr after the timed region -- tools/gpu_host_profile.sh
    seg = {} if os.environ.get('BENCH_HOST_SEGMENTS') == '1' else None
    seg_t = [0.0]

    def mark(name):
        if seg is not None:
            now = time.perf_counter()
            seg[name] = seg.get(name, 0.0) + (now - seg_t[0])
            seg_t[0] = now

    def step(timed=False):
        mark('between_steps')
        for t in (x3d, x2d, w2d):
            t.grad = None
        mark('grads_to_none')
        cost_fun.set_param(x2d.detach(), w2d)
        mark('set_param')
        pose_opt, _, plus, _, logw, cost_init = layer.monte_carlo_forward(
            x3d, x2d, w2d, camera, cost_fun, pose_init=pose_target, force_init_solve=force_init,
            **({'with_pose_opt_plus': True} if with_plus else {}))
        mark('monte_carlo_forward')
        if loss_mod is None or caller:
            if loss_mod is None:
                loss = monte_carlo_pose_loss(logw, cost_init).mean()   # Monte-Carlo pose (KL) loss, NaN -> 0
            else:
                loss = loss_mod(logw, cost_init, scale_net.detach().mean())
            if with_plus:       # derivative regularisation of the callers (lib/train.py:184-193, notebook cell 9)
                dist_t = (plus[:, :3] - pose_target[:, :3]).norm(dim=-1)
                loss_t = torch.where(dist_t < 0.05, 0.5 * dist_t.square() / 0.05, dist_t - 0.025).mean()
                dot = (plus[:, 3:] * pose_target[:, 3:]).sum(-1)
                loss = loss + 0.1 * loss_t + 0.1 * ((1 - dot.square()) * 2).mean()
            mark('loss')
            loss.backward()
            mark('backward')
            return loss
        # Det step.  pose_opt is final here: its all-gather (with the rank's norm_factor input in the same payload) is issued
        # now, in stream order (sharding.ObjectExchange: why not a side stream); the loss takes the world mean out of the
        # exchange, the gathered poses are picked up after backward().
        timed = timed and on_gpu
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # (the head's norm_factor input, deform_pnp_head.py:870, is summed inside the exchange's pack launch: sharding.ObjectExchange.start)
        exchange.start(pose_opt, sum_of=scale_det, sum_row_weight=obj_weight, sum_scale=nf_scale)
        mark('exchange_start')
        layer_last_pose['pose_opt'] = pose_opt.detach()
        if timed:
            e1.record()                 # GPU time of the pack kernel + the RCCL kernel in stream order
            coll_events.append((e0, e1))
        # detection loss: per-object weights, avg_factor = the whole batch, world-mean EMA of norm_factor
        loss = loss_mod(logw, cost_init, exchange, weight=obj_weight, avg_factor=float(total))
        mark('loss')
        loss.backward()
        mark('backward')
        gathered['pose_opt'] = exchange.objects()
        mark('exchange_objects')
        return loss

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    # Device spin-up, before the W warm-up steps and outside everything that is timed.  In a fresh process the first
    # ~17 steps of this workload run slow on the MI355X and converge geometrically to the steady state (2.08, 2.02, 1.97,
    # ... 1.72 ms per step, `tools/step_transient.py`, `profiles/r02_step_transient.txt`); 0.3 s of GEMM load beforehand
    # does not change that, so it is not the idle -> busy clock ramp but the device settling on this instruction mix.
    # With W = 5 that transient sat inside the timed region (+3 % on a 20-step run).  SPINUP_STEPS untimed steps of the
    # workload itself take it out whatever W the caller picks; the JSON line reports them as `device_spinup_steps`.
    for _ in range(SPINUP_STEPS if on_gpu else 0):
        step()
    for _ in range(args.warmup):
        step()
    launch = args.launch if args.launch != 'auto' else ('graph' if args.config in LAUNCH_BOUND else 'eager')
    if not on_gpu:
        launch = 'eager'
    launch_note = None
    # Per-kernel stage times: HIP events around every kernel stage inside the library, over `prof_steps` eager steps just BEFORE
    # the timed region.  Inside it they are not free: ~16 event records per step cost the GPU-bound C2 step 3 % (round 4, same
    # box: 2.522 M instances/s with the events in the timed region, 2.605 M with them in front of it -- they used to disappear
    # in a 1.76 ms step; profiles/r04_bench_event_overhead.txt), they slow the host-bound eager Det step by 40 %, and they cannot
    # sit in a hipGraph at all.  BENCH_PROF_IN_REGION=1 puts them back into the timed region of an eager GPU-bound run.
    prof_in_region = (launch == 'eager' and args.config not in LAUNCH_BOUND and os.environ.get('BENCH_PROF_IN_REGION', '0') == '1')
    prof_steps = args.steps if prof_in_region else ((10 if args.config in LAUNCH_BOUND else 20) if on_gpu else 1)
    step_clocks = None
    if not prof_in_region:
        fence()
        _hip.profile(enable=True, reset=True)
        if on_gpu and rank == 0:      # engine clock / power WHILE the step's kernels run (this window, not the timed one: the sampler is a host thread)
            with ClockSampler(dev.index or 0, period_s=0.002) as clk:
                for _ in range(prof_steps):
                    step(timed=True)
                fence()
            step_clocks = clk.summary()
        else:
            for _ in range(prof_steps):
                step(timed=True)
            fence()
        _hip.profile(enable=False)
    graph, held = None, {}
    if launch == 'graph':
        # The whole rank step -- set_param, forward, the exchange, loss, backward -- captured once and replayed.  The Philox
        # call counter lives in device memory and is advanced in-stream, so every replay draws fresh samples.
        try:
            layer.enable_graph_safe_rng(dev)
            # The leaves' AccumulateGrad nodes were created by the eager steps above on the default stream and are kept alive
            # by the autograd graph behind cost_fun.delta (set_param builds the new graph before the old one is released): a
            # backward under capture would have to synchronise with the default stream, which a capture cannot.  Drop that
            # graph so that the warm-up below creates fresh nodes on a side stream.
            cost_fun.delta = None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            fence()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                held['loss'] = step()
        except Exception as e:          # a capture that is refused falls back to eager launches, and the line says so
            graph, launch = None, 'eager'
            launch_note = 'hipGraph capture failed, eager launches timed instead: ' + repr(e)[:200]
            sync()
    run = graph.replay if graph is not None else (lambda: held.__setitem__('loss', step(timed=prof_in_region)))
    for _ in range(3 if graph is not None else 0):
        run()
    fence()
    if prof_in_region:
        _hip.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    _hip.profile(enable=False)
    loss = held['loss']
    loss_val = float(loss.detach())
    replay_check = None
    if strong:      # what the LAST timed step fed the loss, against the same expression evaluated now, eagerly
        got, want = float(exchange.world_mean()), float(norm_factor_input())
        replay_check = {'norm_factor_input_last_step': got, 'evaluated_eagerly': want, 'norm_factor': float(loss_mod.norm_factor)}
        assert abs(got - want) <= 1e-5 * abs(want), ('the replayed step computed another norm_factor input', replay_check)
    my_ms = elapsed / args.steps * 1e3
    ranks = {'launcher': 'torch.distributed.run' if 'RANK' in os.environ else 'single process', 'process_group': None}
    if dist is not None:
        # evidence that the collective backend really spans `world` processes, one GPU each: the group's size, an all-reduce
        # of ones, and every rank's device index / step time gathered over RCCL
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        # (the PHYSICAL device of a rank: its PCI address -- under per-rank HIP_VISIBLE_DEVICES every rank calls its device "0")
        pci = rank
        if on_gpu:
            pr = torch.cuda.get_device_properties(dev)
            pci = (int(pr.pci_domain_id) << 16) | (int(pr.pci_bus_id) << 8) | int(pr.pci_device_id)
        mine = torch.tensor([float(rank), float(torch.cuda.current_device() if on_gpu else rank), my_ms, float(pci)], device=dev, dtype=torch.float64)
        allr = torch.empty(world * 4, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 4).cpu()
        pcis = [int(v) for v in allr[:, 3]]
        assert len(set(pcis)) == world, f'{world} ranks on {len(set(pcis))} distinct devices: {pcis} (two ranks share a GPU)'
        ranks.update(process_group=dist.get_backend(), rccl_world_size=dist.get_world_size(), all_reduce_of_ones=float(ones),
                     devices=[int(v) for v in allr[:, 1]],
                     pci_addresses=[f'{v >> 16:04x}:{(v >> 8) & 0xff:02x}:{v & 0xff:02x}.0' for v in pcis] if on_gpu else None,
                     device_name=torch.cuda.get_device_name(dev) if on_gpu else 'cpu (test hook)',
                     ms_per_step_per_rank=[round(float(v), 4) for v in allr[:, 2]],
                     ms_per_step_min=round(float(allr[:, 2].min()), 4), ms_per_step_max=round(float(allr[:, 2].max()), 4))
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
        assert ranks['rccl_world_size'] == world == int(ranks['all_reduce_of_ones']), ranks
    if strong:
        assert gathered['pose_opt'].shape[0] == total
        # the exchange only moves data: this rank's rows of the gathered tensor are its own pose outputs, bit for bit
        lo_, hi_ = sharding.shard_range(total, rank, world)
        own = layer_last_pose['pose_opt']
        assert torch.equal(gathered['pose_opt'][lo_:hi_], own), 'gathered poses differ from the local ones'
        gathered_bytes = int(gathered['pose_opt'].numel() * 4 + 4)      # (the A/B below re-runs the step WITHOUT the exchange)

    # The same steps with the per-stage HIP events INSIDE the timed window -- rounds 1-3 timed the step that way -- so that the
    # headline can be compared like for like with BENCH_r01..r03 (~16 event records per step: ~3 % of a 1.5 ms step).
    STAGES = ('evaluate_cost', 'rslm_solve', 'lm_solve', 'amis_forward', 'amis_backward', 'adaptive_delta', 'mc_loss_forward',
              'mc_loss_backward', 'center_points', 'shift_poses')
    stage_ms = {n: _hip.profile_read(n) for n in STAGES} if on_gpu else {}       # (read before the next leg re-uses the recorder)
    events_in_region = None
    if on_gpu and launch == 'eager' and not prof_in_region and args.config not in LAUNCH_BOUND and world == 1:
        n_ev = min(args.steps, 100)
        fence()
        _hip.profile(enable=True, reset=True)
        t0 = time.perf_counter()
        for _ in range(n_ev):
            step(timed=True)
        fence()
        t_ev = time.perf_counter() - t0
        _hip.profile(enable=False)
        events_in_region = {'ms_per_step': round(t_ev / n_ev * 1e3, 4), 'value': round(total * n_ev / t_ev, 1), 'unit': 'instances/s',
                            'steps': n_ev, 'note': 'per-stage HIP events recorded inside this window (the protocol of BENCH_r01..r03)'}

    eager_ms = None
    if graph is not None:     # what a caller who just swaps the package gets: the same K steps launched eagerly, same run
        for _ in range(max(args.warmup, 3)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        te = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
        eager_ms = float(te) / args.steps * 1e3

    if seg is not None and on_gpu:
        n_seg = 500
        for _ in range(20):
            step()
        fence()
        seg.clear()
        seg_t[0] = t0 = time.perf_counter()
        for _ in range(n_seg):
            step()
        t_host = time.perf_counter() - t0
        fence()
        t_all = time.perf_counter() - t0
        print(json.dumps({'host_segments_us_per_step': {k: round(v / n_seg * 1e6, 2) for k, v in seg.items()},
                          'host_us_per_step': round(t_host / n_seg * 1e6, 2), 'with_drain_us_per_step': round(t_all / n_seg * 1e6, 2),
                          'steps': n_seg, 'config': args.config}), file=sys.stderr)
        seg = None

    if os.environ.get('BENCH_TORCH_PROFILE') == '1' and on_gpu:      # CPU-side op times of the eager step (torch.profiler), to stderr
        from torch.profiler import ProfilerActivity, profile
        for _ in range(20):
            step()
        fence()
        with profile(activities=[ProfilerActivity.CPU]) as tp:
            for _ in range(200):
                step()
            fence()
        print(tp.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=70), file=sys.stderr)

    ms_without = None
    if strong:          # the collective's cost on the critical path, measured: the same steps without the exchange
        exchange.disabled = True
        run = step
        if graph is not None:
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2):
                step()
            run = graph2.replay
        for _ in range(max(args.warmup, 3)):
            run()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        fence()
        tw = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        ms_without = float(tw) / args.steps * 1e3
        exchange.disabled = False

    if rank == 0 and not on_gpu:      # test hook: the harness logic ran (sharding, exchange, loss, timing protocol); no GPU legs
        ms = elapsed / args.steps * 1e3
        out = {'metric': f'PnP instances/sec (fwd+bwd, N={N} pts, {S} samples)', 'value': round(total * args.steps / elapsed, 1),
               'unit': 'instances/s', 'n_gpus': world, 'ranks': ranks, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(ms, 4), 'scaling': cfg['scaling'], 'device': str(dev), 'loss': round(loss_val, 5),
               'config': {'name': args.config, 'objects_per_gpu': B, 'objects_total': total}}
        if strong:
            out['collective'] = {'route': exchange.route, 'ms_per_step_without_exchange': round(ms_without, 4),
                                 'bytes_per_rank': gathered_bytes,
                                 'gathered_equals_local_bitwise': True, 'replayed_step_check': replay_check}
        print(json.dumps(out), flush=True)
    if rank == 0 and on_gpu:
        ms = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed
        t_lm, t_fw, t_bw, t_ci = (stage_ms[n][0] for n in ('lm_solve', 'amis_forward', 'amis_backward', 'evaluate_cost'))
        assert stage_ms['lm_solve'][1] == prof_steps and stage_ms['amis_backward'][1] == prof_steps, stage_ms
        sweeps = 1 + L
        p_len, d = (7, 6) if dof == 6 else (4, 4)
        lm_bytes = sweeps * 28.0 * B * N + B * 4.0 * (p_len + 9 + 1) + B * 4.0 * (p_len + d * d + 1)
        lm_gbs = lm_bytes / (t_lm * 1e-3) / 1e9
        fw_tf = 40.0 * S * N * B / (t_fw * 1e-3) / 1e12
        bw_tf = 80.0 * (S + 1) * N * B / (t_bw * 1e-3) / 1e12
        shape_key = f'{args.config}:B{B}:N{N}:S{S}:K{K}:L{L}' if default_shape else None
        lm_traffic, lm_src = measured_traffic('lm_solve_kernel', shape_key) if shape_key else (None, None)
        # one physical sweep: normal_equations_kernel, IC-cold (rotation over distinct copies of the correspondences)
        ne_bytes = B * (28.0 * N + 4.0 * (p_len + 9 + 1 + 4) + 4.0 * (d * (d + 1) // 2 + d + 1))
        mk = lambda: F.PnPProblem(x3d.detach().clone(), x2d.detach().clone(), w2d.detach().clone(), camera, cost_fun, dof)
        ne_mean_ms, ne_median_ms, ne_sets = single_sweep(F, mk, prob['pose_init'], ne_bytes)
        ne_gbs = ne_bytes / (ne_mean_ms * 1e-3) / 1e9
        ne_clocks = LAST_SWEEP_CLOCKS[0]
        ne_traffic, ne_src = measured_traffic('normal_equations_kernel', shape_key) if shape_key else (None, None)
        names = {'C2': 'C2 batched synthetic', 'C5': 'C5 stress (one shard per GPU)', 'C4': 'C4 EPro-PnP-Det nuScenes shape',
                 'C1': 'C1 demo/fit_identity.ipynb plumbing case', 'C3': 'C3 EPro-PnP-6DoF LineMOD shape, dense 64x64 crops',
                 'C3-train': 'C3 EPro-PnP-6DoF LineMOD training call (512 sub-sampled correspondences)'}
        ic_cold = ne_sets * ne_bytes >= 2 * IC_BYTES
        # roofline.large (default C2 line only): the same kernel on the C5-shard sweep, 470 MB per launch -- IC-cold by
        # construction and long enough (~0.1 ms) that launch ramp and tail do not set the figure
        large = None
        if args.config == 'C2' and default_shape and world == 1 and not args.no_large_sweep:
            LB, LN = LARGE_SWEEP
            lp = synth_problem(LB, LN, dev, seed=77, dof=6)
            l_bytes = LB * (28.0 * LN + 4.0 * (7 + 9 + 1 + 4) + 4.0 * (21 + 6 + 1))
            lcam = PerspectiveCamera(cam_mats=lp['cam_mats'], z_min=0.1)
            lcf = AdaptiveHuberPnPCost(relative_delta=0.5)
            lcf.set_param(lp['x2d'], lp['w2d'])
            lmk = lambda: F.PnPProblem(lp['x3d'].clone(), lp['x2d'].clone(), lp['w2d'].clone(), lcam, lcf, 6)
            l_mean, l_med, l_sets = single_sweep(F, lmk, lp['pose_init'], l_bytes, windows=12)
            l_traffic, l_src = measured_traffic('normal_equations_kernel', f'C5:B{LB}:N{LN}:S1024:K4:L3')
            large = {'workload': f'{LB} objects x N={LN} points (the C5 shard), one Jacobian sweep', 'achieved': round(l_bytes / (l_mean * 1e-3) / 1e9, 1),
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(l_bytes / (l_mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     'algorithmic_bytes_per_launch': l_bytes, 'launch_ms': round(l_mean, 5), 'launch_ms_median': round(l_med, 5),
                     'cache_state': f'IC-cold: {l_sets} copies x {l_bytes / 2 ** 20:.0f} MiB', 'traffic': l_traffic,
                     'traffic_source': l_src, 'clocks': LAST_SWEEP_CLOCKS[0]}
            del lp, lmk
        par = (f'one batch of {total} objects split x{world} ({B} on rank 0), ONE all_gather_into_tensor (pose outputs + '
               f'norm_factor) inside the step') if strong else f'objects sharded x{world}, no data-path collective'
        out = {
            'metric': f'PnP instances/sec (fwd+bwd, N={N} pts, {S} samples)',
            'value': round(value, 1), 'unit': 'instances/s', 'n_gpus': world, 'ranks': ranks, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': cfg['scaling'],
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'device_spinup_steps': SPINUP_STEPS,
            # how the fp32 arithmetic is carried out where it is not plain fp32 instructions ("Round 4")
            'dtype_note': ('fp32 throughout; the pose x point projection of the two AMIS kernels is evaluated on v_mfma_f32_16x16x32_bf16 with each '
                           'fp32 operand split into three bf16 pieces that sum to it exactly (8 of the 9 cross products carried, fp32 accumulation: '
                           '1.4e-7 relative against 1.2e-7 for the fp32 MFMA); EPROPNP_FWD_PROJ=f32 EPROPNP_BWD_PROJ=f32 select the fp32 MFMA: '
                           + ('fp32 MFMA selected' if os.environ.get('EPROPNP_FWD_PROJ', '')[:1] == 'f' and os.environ.get('EPROPNP_BWD_PROJ', '')[:1] == 'f'
                              else 'split projection (default)')),
            'launch': 'eager' if launch == 'eager' else 'hipGraph replay of the whole rank step, RCCL exchange included (captured once; fresh samples per replay)',
            'launch_note': launch_note,
            'kernel_ms_source': 'HIP events inside the library over the timed region' if prof_in_region else
                                (f'HIP events inside the library over {prof_steps} eager steps BEFORE the timed region -- a separate window, and every event pair '
                                 f'adds ~2.5 us to its stage: the sum of kernel_ms may exceed ms_per_step by ~1 %'),
            'config': {'workload': f'{names[args.config]}: {B} objects/GPU x N={N} points, S={S} MC samples, '
                                   f'K={K} AMIS iters, L={L} LM iters, EProPnP{dof}DoF fwd+bwd'
                                   + (', RSLM(16,64,3) init, normalize=True, Det loss' if args.config == 'C4' else '')
                                   + ((f", RSLM{caller['rslm']} init" if caller['rslm'] else '') + (', per-object crop bounds' if caller['crop_bounds'] else '')
                                      + f", z_min {caller['z_min']}, relative_delta {caller['relative_delta']}"
                                      + (', with_pose_opt_plus + derivative regularisation' if caller['plus'] else '') if caller else ''),
                       'name': args.config, 'objects_per_gpu': B, 'objects_total': total, 'num_points': N, 'mc_samples': S,
                       'amis_iters': K, 'lm_iters': L, 'dof': dof, 'parallelism': par},
            # the Jacobian sweep against the HBM peak.  Headline = ONE physical sweep with the Infinity Cache out of the
            # picture; `fused_lm` = the LM kernel of the step, which reads the points once for its 1+L logical sweeps: its
            # physical fraction first, then SURVEY 8d's logical-sweep credit, named as such.
            'roofline': {'kernel': 'normal_equations_kernel (one Jacobian sweep: one logical = one physical read of the '
                                   'correspondences)', 'bound': 'hbm',
                         'achieved': round(ne_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(ne_gbs / HBM_PEAK_GBS, 4),
                         'cache_state': (f'IC-cold: launches rotate over {ne_sets} distinct copies of the inputs '
                                         f'({ne_sets * ne_bytes / 2 ** 20:.0f} MiB > 2 x 256 MiB Infinity Cache)') if ic_cold else
                                        (f'IC-WARM: {ne_sets} copies x {ne_bytes / 2 ** 20:.2f} MiB fit the 256 MiB Infinity Cache; '
                                         f'this launch-bound shape says nothing about HBM'),
                         'large': large,
                         # HBM bytes per launch from rocprofv3 PMC (separate passes; gfx950: 2 x FETCH_SIZE + WRITE_SIZE,
                         # MI355X_MICROARCH.md), read from the committed profile of this exact shape -- or null
                         'traffic': ne_traffic, 'traffic_source': ne_src,
                         'algorithmic_bytes_per_launch': ne_bytes,
                         'launch_ms': round(ne_mean_ms, 5), 'launch_ms_median': round(ne_median_ms, 5),
                         # engine / memory clock and package power sampled from the amdgpu hwmon nodes WHILE the windows above ran
                         'clocks': ne_clocks,
                         'fused_lm': {'kernel': 'lm_solve_kernel (1+L Jacobian sweeps in one launch, points read once)',
                                      'launch_ms': round(t_lm, 4),
                                      'physical_frac': None if lm_traffic is None else round(lm_traffic / (t_lm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      'traffic': lm_traffic, 'traffic_source': lm_src,
                                      'credit': 'logical sweeps', 'logical_sweeps': sweeps,
                                      'credited_bytes_per_launch': lm_bytes,
                                      'credited_achieved': round(lm_gbs, 1), 'credited_frac': round(lm_gbs / HBM_PEAK_GBS, 4)}},
            'roofline_valu': {
                'what': ('fp32-EQUIVALENT work per second against the fp32 vector peak: SURVEY 8(d) nominal 40 / 80 flop per point-pose; 24 of them '
                         'are the projection, which executes as bf16x3-split products on the matrix pipe (v_mfma_f32_16x16x32_bf16), the rest on the VALU -- '
                         'so `frac` is not a utilisation of the fp32 units (DESIGN.md section 4)'),
                'amis_forward_mfma_kernel': {'bound': 'fp32-equivalent (projection on bf16 MFMA, the rest on the VALU)', 'achieved': round(fw_tf, 2), 'peak': FP32_VECTOR_PEAK_TF,
                                        'unit': 'TFLOP/s', 'frac': round(fw_tf / FP32_VECTOR_PEAK_TF, 4),
                                        'flops_per_point_pose': 40, 'launch_ms': round(t_fw, 4)},
                'amis_backward_mfma_kernel': {'bound': 'fp32-equivalent (projection on bf16 MFMA, the rest on the VALU)', 'achieved': round(bw_tf, 2), 'peak': FP32_VECTOR_PEAK_TF,
                                         'unit': 'TFLOP/s', 'frac': round(bw_tf / FP32_VECTOR_PEAK_TF, 4),
                                         'flops_per_point_pose': 80, 'launch_ms': round(t_bw, 4)}},
            'kernel_ms': {n: round(v[0] * (v[1] / prof_steps), 4) for n, v in stage_ms.items() if v[1]},   # per step
            # engine / memory clock and package power sampled while the kernel_ms window ran (the sustained state the kernel times
            # were taken in: the AMIS kernels run close to the board's power cap, and boxes differ in where that puts the clock)
            'clocks_under_step_load': step_clocks,
            'loss': round(loss_val, 5),
        }
        if events_in_region is not None:
            out['with_stage_events_in_region'] = events_in_region
        if eager_ms is not None:
            out['eager'] = {'ms_per_step': round(eager_ms, 4), 'value': round(total / (eager_ms * 1e-3), 1), 'unit': 'instances/s',
                            'note': 'the same steps launched eagerly in this run (host-bound: ~20 launches per step)'}
        if strong:
            c_ms = sum(a.elapsed_time(b) for a, b in coll_events) / max(len(coll_events), 1)
            out['collective'] = {'op': 'ONE all_gather_into_tensor per step: pose_opt chunk + the norm_factor scalar of the '
                                       'detection loss in the same payload, issued in stream order right after the forward '
                                       '(sharding.ObjectExchange)',
                                 'backend': 'nccl (RCCL)' if dist is not None else 'none (single process)', 'route': exchange.route,
                                 'gpu_ms_in_stream_order': round(c_ms, 4),
                                 # A/B in this run: the same K steps with the exchange switched off (every rank, below)
                                 'ms_per_step_without_exchange': round(ms_without, 4),
                                 'share_of_step': round(max(0.0, ms - ms_without) / ms, 4),
                                 'bytes_per_rank': gathered_bytes,
                                 'gathered_equals_local_bitwise': True,
                                 'replayed_step_check': replay_check}
        if world == 1 and not args.no_cpu_baseline and args.config in ('C2', 'C5'):
            out['cpu_baseline'] = (cpu_baseline(N, S, K, L, args.cpu_sample) if args.config == 'C2'
                                   else cpu_baseline(N, S, K, L, total_objects=8, chunk=8, one_thread_objects=2))
            out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 1)
        if world == 1 and not args.no_hipgraph and args.config == 'C2' and default_shape:
            out['hipgraph_replay'] = hipgraph_replay()      # informational: the same step replayed from a hipGraph
        # RCCL writes its version banner through C stdio: into a pipe that is block-buffered and would land BEHIND this line at
        # exit -- push it out first, so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    comms_closed = None
    if exchange is not None:
        exchange.close()                        # ncclCommDestroy of the direct route's communicator, on every rank ...
    sharding.RcclComm.close_all()
    comms_closed = len(sharding.RcclComm._live) == 0
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()            # ... before the torch group goes
    if os.environ.get('BENCH_REPORT_TEARDOWN') == '1' and rank == 0:
        sys.stderr.write(json.dumps({'teardown': {'rccl_comms_closed': comms_closed,
                                                  'process_group_destroyed': dist is None or not dist.is_initialized()}}) + '\n')
    return 0


if __name__ == '__main__':
    sys.exit(main())

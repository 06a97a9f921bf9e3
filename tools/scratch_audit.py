#!/usr/bin/env python3
"""Scratch (register spill) accesses of every gfx950 kernel of the library, by loop depth.

    python tools/scratch_audit.py [--check] [source.hip ...]

Cross-compiles each csrc/*.hip to assembly with the flags build.py uses and, for every kernel that owns a private segment, counts
its scratch_load / scratch_store instructions by the nesting depth of the loop they sit in (LLVM's block comments).  Spills around a
loop cost traffic (round 4: 33 MB per launch of the C2 backward); spills INSIDE the pose-tile loops that feed matrix-core operands
were intermittently wrong on the MI355X (profiles/r05_bwd_scratch.txt).  `--check` fails when a kernel reloads or stores scratch at
loop depth >= 2 (the inner sweep loops) unless it is on the ALLOWED list of instantiations no launcher selects by default."""
import concurrent.futures as cf
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'epro-pnp_amd', 'csrc')
spec = importlib.util.spec_from_file_location('epropnp_build', os.path.join(ROOT, 'epro-pnp_amd', 'build.py'))
build = importlib.util.module_from_spec(spec)
spec.loader.exec_module(build)

# matrix-core instantiations with scratch inside an inner loop that no launcher selects by default: 4-DoF forward with 12 / 16
# resident tiles (EPROPNP_TUNE=fwd_mfma=..; the launcher streams the points through LDS instead), the bounded fp32-projection forward
# with 8 resident tiles (EPROPNP_FWD_PROJ=f32 only), the 6-DoF split-projection forward with 16 resident tiles (more than 48 point
# tiles go through the registers in chunks of 8 per wave; EPROPNP_TUNE=fwd_no_chunks / fwd_mfma=4,16 only).  The all-VALU kernels of amis_kernels.hip / lm_kernel.hip are not subject to the
# rule (no matrix-core operands; their 16-wave classes live on 128 VGPRs by design).
ALLOWED = ('amis_forward_mfma_kernel<4, true, 12, false, false, true, false>', 'amis_forward_mfma_kernel<4, false, 12, false, false, true, false>',
           'amis_forward_mfma_kernel<4, true, 16, false, false, true, false>', 'amis_forward_mfma_kernel<4, false, 16, false, false, true, false>',
           'amis_forward_mfma_kernel<4, true, 16, false, false, false, false>', 'amis_forward_mfma_kernel<6, true, 8, false, false, false, false>',
           'amis_forward_mfma_kernel<6, true, 16, false, false, true, false>', 'amis_forward_mfma_kernel<6, false, 16, false, false, true, false>')
MFMA_FILES = ('amis_forward_mfma.hip', 'amis_backward_mfma.hip')


def audit(src):
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-value',
           '-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, src)] + build.FILE_FLAGS.get(src, [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    kern, depth, res, meta = None, 0, {}, {}
    for line in open(out):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            kern, depth = m.group(1), 0
            continue
        if re.match(r'^\.LBB\d+_\d+:', line):
            mm = re.search(r'Depth=(\d+)', line)
            depth = int(mm.group(1)) if mm else 0
            continue
        if line.lstrip().startswith(';'):
            mm = re.search(r'Depth=(\d+)', line)
            if mm:
                depth = max(depth, int(mm.group(1)))
            continue
        if 'scratch_load' in line or 'scratch_store' in line:
            key = ('load' if 'scratch_load' in line else 'store', depth)
            res.setdefault(kern, {})[key] = res.setdefault(kern, {}).get(key, 0) + 1
        mm = re.match(r'\s+\.(name|private_segment_fixed_size|vgpr_count):\s+(\S+)', line)
        if mm:
            if mm.group(1) == 'name':
                cur = mm.group(2)
                meta[cur] = {}
            elif 'cur' in locals():
                meta[cur][mm.group(1)] = int(mm.group(2))
    return src, res, meta


def short(name):
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    return re.sub(r'\(.*', '', dem).replace('void pnp::', '')


def main():
    check = '--check' in sys.argv
    srcs = [a for a in sys.argv[1:] if a.endswith('.hip')] or [s for s in build.SOURCES if s not in ('c_api.hip', 'mc_forward.hip')]
    bad = []
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for src, res, meta in ex.map(audit, srcs):
            n_scratch = sum(1 for m in meta.values() if m.get('private_segment_fixed_size', 0) > 0)
            print(f'# {src}: {len(meta)} kernels, {n_scratch} with a private segment')
            for k, m in meta.items():
                if m.get('private_segment_fixed_size', 0) > 0:
                    ops = res.get(k, {})
                    inner = sum(v for (op, d), v in ops.items() if d >= 2)
                    name = short(k)
                    flag = ''
                    if inner:
                        flag = '  <-- scratch inside an inner loop' + (' (allowed: tuning override only)' if name in ALLOWED else '')
                        if name not in ALLOWED and src in MFMA_FILES:
                            bad.append(name)
                    print(f'  {name:70s} {m["private_segment_fixed_size"]:4d} B/lane  vgpr {m.get("vgpr_count", 0):3d}  '
                          f'{ {f"{op}@depth{d}": v for (op, d), v in sorted(ops.items())} }{flag}')
    if check and bad:
        print('FAILED: scratch accesses inside inner loops:', *bad, sep='\n  ')
        sys.exit(1)


if __name__ == '__main__':
    main()

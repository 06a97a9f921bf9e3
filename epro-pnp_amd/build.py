"""Build libepropnp_hip.so (gfx950 code object + C ABI) in-tree with hipcc.

    python epro-pnp_amd/build.py            # -> epro-pnp_amd/lib/libepropnp_hip.so
    python epro-pnp_amd/build.py --emu      # -> tests/emu/_build/libepropnp_emu.so (CPU logic emulation, tests only)

hipcc cross-compiles for gfx950 without a GPU.  No torch involved: the library is a plain C ABI.
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['eval_kernels.hip', 'lm_kernel.hip', 'amis_kernels.hip', 'amis_forward_mfma.hip', 'amis_backward_mfma.hip', 'gn_step_kernel.hip', 'rslm_kernel.hip', 'mc_forward.hip', 'c_api.hip']
# Every translation unit is compiled WITHOUT the SLP vectoriser (HIP build only).  Two reasons, the second one decisive:
#  * where the vectoriser packs independent scalar FMAs into v_pk_* it pays for it in v_mov shuffles and gains nothing (packed fp32
#    issues at half the rate of the scalar form, profiles/r01_ubench_valu_rates.txt): lm 79 -> 70 us at C2, rslm 117 -> 107 us at C4,
#    normal_equations 21.6 -> 16.6 us, forward 0.84 -> 0.82 ms (its Huber sweep is packed explicitly, on 2-vectors);
#  * the packed instructions it forms include a shape that returns wrong results on the MI355X while a bf16 MFMA executes on the same
#    SIMD (profiles/r05_pk_opsel_erratum.txt: the run-to-run different gradients of round 5).  Until round 5 the all-VALU sampler /
#    backward (amis_kernels.hip) and the Gauss-Newton step (gn_step_kernel.hip) kept the vectoriser and with it 106 instances of
#    that shape which no operand swap cures -- harmless only as long as no other stream or process runs a bf16 GEMM on the device.
#    Round 6: no compiler-formed packed arithmetic anywhere; the assembly gate below fails the build for ANY function of ANY unit that
#    still holds the shape.
_NO_SLP = ['-fno-slp-vectorize']
FILE_FLAGS = {src: _NO_SLP for src in SOURCES}
# The DEVICE code of every translation unit goes through tools/pk_opsel_fix.py on its way from the compiler to the assembler: a
# rewrite (swap of the two commuting sources: same arithmetic, clean form) with --strict, then --roundtrip (every packed fp32 line of
# the listing must parse and print back token for token: the guard against an assembler syntax the tool does not know) and --audit
# (no function may keep the shape).  Any of the three failing fails the build.
ERRATUM_FILES = tuple(SOURCES)
# LLVM major versions whose assembly syntax / bundle layout this path has been verified against (bit-identical device .text to a
# plain `hipcc -c` on a unit without swaps: tests/test_build_path.py).  Another toolchain: run that test, then add it here or set
# EPROPNP_ALLOW_LLVM=<major>.
VERIFIED_LLVM_MAJORS = (22,)
HEADERS = ['pnp_math.h', 'wave_ops.h', 'pnp_sweep.h', 'pnp_host.h', 'dispatch.h', 'amis_common.h', 'lm_core.h', 'tuning.h']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    if callable(cmd):
        return cmd()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(' '.join(cmd) + '\n' + r.stdout + r.stderr)
        raise RuntimeError('build failed: ' + cmd[-1])
    return r.stdout + r.stderr


_TOOLCHAINS = {}


def toolchain(hipcc, arch):
    """What `hipcc -###` says about the device pipeline of this installation: the LLVM bin directory, the LLVM major version, the
    device triple and -target-cpu, the lld arguments and the offload bundle's target list -- read off the driver instead of being
    hard-coded, so that a different ROCm layout or an `arch` with feature suffixes (gfx950:xnack-) builds the same way hipcc does."""
    import re
    import shlex
    import tempfile
    key = (hipcc, arch)
    if key in _TOOLCHAINS:
        return _TOOLCHAINS[key]
    with tempfile.TemporaryDirectory() as td:
        probe = os.path.join(td, 'probe.hip')
        open(probe, 'w').write('__global__ void k() {}\n')
        r = subprocess.run([hipcc, f'--offload-arch={arch}', '-O3', '-fno-gpu-rdc', '-c', probe, '-o', os.path.join(td, 'probe.o'), '-###'],
                           capture_output=True, text=True)
    text = r.stdout + r.stderr
    if r.returncode != 0:
        raise RuntimeError('build failed: `hipcc -###` (toolchain discovery):\n' + text)
    m = re.search(r'^InstalledDir:\s*(\S+)', text, flags=re.M)
    v = re.search(r'clang version (\d+)', text)
    cmds = [shlex.split(line) for line in text.splitlines() if line.startswith(' "')]
    dev = next((c for c in cmds if '-fcuda-is-device' in c), None)
    lld = next((c for c in cmds if os.path.basename(c[0]).startswith('lld')), None)
    bun = next((c for c in cmds if 'clang-offload-bundler' in os.path.basename(c[0])), None)
    if not (m and v and dev and lld and bun):
        raise RuntimeError('build failed: cannot read the device pipeline off `hipcc -###` (InstalledDir / device cc1 / lld / '
                           'clang-offload-bundler not found); this build path knows the ROCm 7.x driver layout:\n' + text[:2000])
    tc = {'bin': os.environ.get('EPROPNP_LLVM_BIN', m.group(1)), 'llvm_major': int(v.group(1)),
          'triple': dev[dev.index('-triple') + 1], 'cpu': dev[dev.index('-target-cpu') + 1],
          'features': [dev[i + 1] for i, a in enumerate(dev) if a == '-target-feature'],
          'lld_m': lld[lld.index('-m') + 1],
          'bundle_targets': next(a for a in bun if a.startswith('-targets=')),
          'bundle_align': next((a for a in bun if a.startswith('-bundle-align=')), '-bundle-align=4096')}
    allowed = set(VERIFIED_LLVM_MAJORS) | {int(x) for x in os.environ.get('EPROPNP_ALLOW_LLVM', '').split(',') if x.strip().isdigit()}
    if tc['llvm_major'] not in allowed:
        raise RuntimeError(f"build failed: {hipcc} is LLVM {tc['llvm_major']}; the assembly-rewrite step of this build "
                           f'(tools/pk_opsel_fix.py between `hipcc -S` and the assembler) is verified for LLVM {sorted(VERIFIED_LLVM_MAJORS)} only. '
                           f"Run tests/test_build_path.py with EPROPNP_ALLOW_LLVM={tc['llvm_major']}; if it passes, build with that variable set.")
    _TOOLCHAINS[key] = tc
    return tc


def _compile_with_erratum_fix(cc, src, obj, arch):
    """hipcc -c, with the device assembly checked / rewritten in between: device asm -> tools/pk_opsel_fix.py -> assembler -> lld ->
    offload bundle -> host-only compile that embeds it (the steps `hipcc -###` shows, with one text pass in the middle; every tool
    path, triple, cpu and bundle id is taken from that same `hipcc -###`, toolchain())."""
    tc = toolchain(cc[0], arch)
    fix = os.path.join(ROOT, 'tools', 'pk_opsel_fix.py')
    base = obj[:-2]
    asm, fixed, dev_o, dev_out, fatbin = base + '.dev.s', base + '.dev.fixed.s', base + '.dev.o', base + '.dev.out', base + '.hipfb'
    out = _run(cc + ['-S', '--cuda-device-only', src, '-o', asm])
    out += _run([sys.executable, fix, '--strict', asm, fixed])     # exit 1 if a shape is left that no swap cures
    out += _run([sys.executable, fix, '--roundtrip', fixed])       # exit 1 on a packed fp32 line the tool cannot read back
    out += _run([sys.executable, fix, '--audit', fixed])           # exit 1 if ANY function keeps the unsafe form
    asm_cmd = [os.path.join(tc['bin'], 'clang'), '-x', 'assembler', '-target', tc['triple'], f"-mcpu={tc['cpu']}"]
    for f in tc['features']:                                       # an arch with feature suffixes: gfx950:xnack- -> -mno-xnack
        if f[1:] in ('xnack', 'sramecc'):
            asm_cmd.append(('-m' if f[0] == '+' else '-mno-') + f[1:])
    out += _run(asm_cmd + ['-c', fixed, '-o', dev_o])
    out += _run([os.path.join(tc['bin'], 'lld'), '-flavor', 'gnu', '-m', tc['lld_m'], '--no-undefined', '-shared', '-o', dev_out, dev_o])
    out += _run([os.path.join(tc['bin'], 'clang-offload-bundler'), '-type=o', tc['bundle_align'], tc['bundle_targets'],
                 '-input=/dev/null', f'-input={dev_out}', f'-output={fatbin}'])
    out += _run(cc + ['--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', fatbin, '-c', src, '-o', obj])
    for tmp in (asm, dev_o, dev_out, fatbin):          # the rewritten assembly stays next to the object: tests audit it
        os.remove(tmp)
    return out


def build(emu=False, force=False, verbose=False, defines=(), tag=None, flags=(), file_flags=(), arch=None):
    deps_common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, 'include', 'epropnp_hip.h')]
    if emu:
        out_dir = os.path.join(ROOT, 'tests', 'emu', '_build')
        lib = os.path.join(out_dir, 'libepropnp_emu.so')
        shim = os.path.join(ROOT, 'tests', 'emu', 'hip_emu.h')
        deps_common.append(shim)
        # the kernel headers hold no test branches: <hip/hip_runtime.h> resolves to the shim, which supplies host functions under
        # the names of the AMDGPU builtins the headers use (-ffp-contract=off: what `#pragma clang fp contract(off)` asks for)
        cc = ['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-x', 'c++', '-include', shim,
              '-I', os.path.join(ROOT, 'tests', 'emu', 'include'), '-ffp-contract=off', '-Wno-unknown-pragmas', '-Wno-attributes']
    else:
        out_dir = os.path.join(HERE, 'lib') if not tag else os.path.join(HERE, 'lib', 'variants', tag)
        lib = os.path.join(out_dir, 'libepropnp_hip.so')
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        arch = arch or os.environ.get('EPROPNP_OFFLOAD_ARCH', 'gfx950')     # the kernels are written for gfx950 (MI355X)
        cc = [hipcc, f'--offload-arch={arch}', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-value']
        cc += ['-D' + d for d in defines] + list(flags)
    os.makedirs(out_dir, exist_ok=True)
    per_file = {k: list(v) for k, v in FILE_FLAGS.items()}
    for spec in file_flags:          # tuning variants: "--file-flag lm_kernel.hip=-fslp-vectorize"
        name, flag = spec.split('=', 1)
        per_file.setdefault(name, []).append(flag)
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(out_dir, src.replace('.hip', '.emu.o' if emu else '.o'))
        objs.append(obj)
        if force or _stale(obj, [sp] + deps_common + ([os.path.join(ROOT, 'tools', 'pk_opsel_fix.py')] if src in ERRATUM_FILES else [])):
            if not emu and src in ERRATUM_FILES:
                jobs.append(lambda c=cc + per_file.get(src, []), sp=sp, obj=obj: _compile_with_erratum_fix(c, sp, obj, arch))
            else:
                jobs.append(cc + ([] if emu else per_file.get(src, [])) + ['-c', sp, '-o', obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    if jobs or force or _stale(lib, objs):
        link = (['g++'] if emu else cc[:2]) + ['-shared', '-fPIC', '-o', lib] + objs
        _run(link)
    return lib


def build_torch_binding(force=False, verbose=False):
    """csrc/torch_binding.cpp -> lib/_epropnp_torch.so: the C++ autograd nodes over the C ABI (host code only: g++ against
    the torch headers, linked to libepropnp_hip.so through an $ORIGIN rpath so the pair travels together)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    out_dir = os.path.join(HERE, 'lib')
    lib = os.path.join(out_dir, '_epropnp_torch.so')
    src = os.path.join(CSRC, 'torch_binding.cpp')
    hip_lib = os.path.join(out_dir, 'libepropnp_hip.so')
    if not force and not _stale(lib, [src, os.path.join(ROOT, 'include', 'epropnp_hip.h'), hip_lib]):
        return lib
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wno-attributes', src, '-o', lib,
           '-DTORCH_EXTENSION_NAME=_epropnp_torch', '-DTORCH_API_INCLUDE_EXTENSION_H',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}', '-I' + sysconfig.get_paths()['include']]
    cmd += ['-isystem' + p for p in ce.include_paths()]
    cmd += ['-L' + out_dir, '-lepropnp_hip', '-Wl,-rpath,$ORIGIN']
    for d in ce.library_paths():
        cmd += ['-L' + d, '-Wl,-rpath,' + d]
    cmd += ['-lc10', '-ltorch_cpu', '-ltorch', '-ltorch_python']
    out = _run(cmd)
    if verbose and out.strip():
        print(out)
    return lib


def build_erratum_probe(force=False):
    """tools/ubench/pk_erratum_probe.hip -> tools/ubench/pk_erratum_probe (a stand-alone HIP program: the packed-fp32 / bf16-MFMA
    erratum's microbenchmark, which tests/test_erratum_gpu.py runs on every GPU box; it is rebuilt there if this binary is stale)."""
    src = os.path.join(ROOT, 'tools', 'ubench', 'pk_erratum_probe.hip')
    out = src[:-4]
    if force or _stale(out, [src]):
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        _run([hipcc, f"--offload-arch={os.environ.get('EPROPNP_OFFLOAD_ARCH', 'gfx950')}", '-O2', '-w', src, '-o', out])
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--emu', action='store_true')
    ap.add_argument('--force', action='store_true')
    ap.add_argument('-v', '--verbose', action='store_true')
    ap.add_argument('-D', dest='defines', action='append', default=[], help='extra -D for a tuning variant')
    ap.add_argument('--tag', default=None, help='build into lib/variants/<tag>/ (tuning variants)')
    ap.add_argument('--flag', dest='flags', action='append', default=[], help='extra compiler flag for a tuning variant')
    ap.add_argument('--file-flag', dest='file_flags', action='append', default=[], help='<source>=<flag> for a tuning variant')
    ap.add_argument('--offload-arch', default=None, help='GPU architecture (default gfx950 = MI355X, or $EPROPNP_OFFLOAD_ARCH)')
    a = ap.parse_args()
    print(build(a.emu, a.force, a.verbose, a.defines, a.tag, a.flags, a.file_flags, a.offload_arch))
    if not a.emu and not a.tag:
        print(build_torch_binding(a.force, a.verbose))
        print(build_erratum_probe(a.force))

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
# per-source flags (HIP build only).  Where the SLP vectoriser packs independent scalar FMAs into v_pk_* it pays for it in
# v_mov shuffles and gains nothing (packed fp32 runs at the scalar flop rate): lm 79 -> 70 us at C2, rslm 117 -> 107 us at
# C4, forward 0.84 -> 0.82 ms (its Huber sweep is packed explicitly, on 2-vectors).
# The MFMA backward: until round 5 it kept the vectoriser (its pair loop gains ~4 % from packed multiply-adds).  The packed
# instructions the vectoriser forms there include shapes that return wrong results on the MI355X while a bf16 MFMA executes on
# the same SIMD (profiles/r05_pk_opsel_erratum.txt: the run-to-run different gradients of round 5) -- one shape is identified and
# rewritten below (ERRATUM_FILES), a build with only that rewrite still failed now and then, so the backward is compiled without
# compiler-formed packed arithmetic altogether.
_NO_SLP = ['-fno-slp-vectorize']
FILE_FLAGS = {'lm_kernel.hip': _NO_SLP, 'rslm_kernel.hip': _NO_SLP, 'amis_forward_mfma.hip': _NO_SLP,
              'eval_kernels.hip': _NO_SLP,      # normal_equations 21.6 -> 16.6 us, evaluate_cost 17 -> 13.9 us at C2
              'amis_backward_mfma.hip': _NO_SLP}
# The DEVICE code of every translation unit with kernels goes through tools/pk_opsel_fix.py on its way from the compiler to the
# assembler (the gfx950 erratum of profiles/r05_pk_opsel_erratum.txt: a packed fp32 instruction whose low lane takes src0.lo and
# src1.hi returns wrong results while a v_mfma_f32_16x16x32_bf16 executes on the SIMD; swapping the two commuting sources is the
# same arithmetic in a form that is clean).  The build FAILS if a kernel that issues such an MFMA itself keeps the shape (the two
# *_mfma.hip units); in the others -- which can only meet an MFMA of another stream's kernel -- the few shapes no swap cures stay.
ERRATUM_FILES = ('amis_forward_mfma.hip', 'amis_backward_mfma.hip', 'amis_kernels.hip', 'gn_step_kernel.hip', 'eval_kernels.hip',
                 'lm_kernel.hip', 'rslm_kernel.hip')
LLVM_BIN = os.environ.get('EPROPNP_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
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


def _compile_with_erratum_fix(cc, src, obj, arch):
    """hipcc -c, with the device assembly rewritten in between: device asm -> tools/pk_opsel_fix.py -> assembler -> lld ->
    offload bundle -> host-only compile that embeds it (the steps `hipcc -###` shows, with one text pass in the middle)."""
    fix = os.path.join(ROOT, 'tools', 'pk_opsel_fix.py')
    base = obj[:-2]
    asm, fixed, dev_o, dev_out, fatbin = base + '.dev.s', base + '.dev.fixed.s', base + '.dev.o', base + '.dev.out', base + '.hipfb'
    out = _run(cc + ['-S', '--cuda-device-only', src, '-o', asm])
    out += _run([sys.executable, fix, asm, fixed])
    out += _run([sys.executable, fix, '--audit', fixed])          # exit 1 (= build failure) if an unsafe form is left
    out += _run([os.path.join(LLVM_BIN, 'clang'), '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', f'-mcpu={arch}', '-c', fixed, '-o', dev_o])
    out += _run([os.path.join(LLVM_BIN, 'lld'), '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', dev_out, dev_o])
    out += _run([os.path.join(LLVM_BIN, 'clang-offload-bundler'), '-type=o', '-bundle-align=4096',
                 f'-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{arch}', '-input=/dev/null', f'-input={dev_out}',
                 f'-output={fatbin}'])
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
        cc = ['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-x', 'c++', '-include', shim, '-Wno-unknown-pragmas',
              '-Wno-attributes']
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

"""The C-ABI shared library exists after build(), exports every symbol include/*.h declares, and refuses to compute on
the host (no CPU fallback).  No GPU needed: nothing is launched."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'epro-pnp_amd', 'lib', 'libepropnp_hip.so')


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'epropnp_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(epropnp_[a-z_]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(LIB):
        import importlib.util
        spec = importlib.util.spec_from_file_location('epropnp_build', os.path.join(ROOT, 'epro-pnp_amd', 'build.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    return ctypes.CDLL(LIB)


def test_exports_match_header(lib):
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/epropnp_hip.h but not exported'
    from epropnp import _hip
    assert sorted(_hip.EXPORTS) == syms


def test_abi_version_and_noise_stride(lib):
    assert lib.epropnp_abi_version() == 6
    assert lib.epropnp_noise_stride(6) == 8 and lib.epropnp_noise_stride(4) == 52 and lib.epropnp_noise_stride(5) == -1


def test_argument_validation_without_launch(lib):
    from epropnp import _hip
    lib.epropnp_last_error.restype = ctypes.c_char_p
    prob = _hip.Problem(None, None, None, None, None, None, None, 0.1, 4, 16, 5)
    rc = lib.epropnp_evaluate_cost(ctypes.byref(prob), None, 1, None, None)
    assert rc == -1 and b'dof' in lib.epropnp_last_error()
    prob = _hip.Problem(None, None, None, None, None, None, None, 0.1, 4, 16, 6)
    rc = lib.epropnp_evaluate_cost(ctypes.byref(prob), None, 1, None, None)
    assert rc == -1 and b'NULL' in lib.epropnp_last_error()
    # z_min is a depth clamp: the forward sweep's one-instruction max(z, z_min) relies on z_min >= 0
    prob = _hip.Problem(None, None, None, None, None, None, None, -0.1, 0, 16, 6)
    rc = lib.epropnp_evaluate_cost(ctypes.byref(prob), None, 1, None, None)
    assert rc == -1 and b'z_min' in lib.epropnp_last_error()


def test_negative_z_min_is_refused_by_the_python_layer(backend):
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import HuberPnPCost
    z = lambda *s: torch.zeros(*s, device=backend)
    cam = PerspectiveCamera(cam_mats=torch.eye(3, device=backend).expand(2, 3, 3), z_min=-0.1)
    with pytest.raises(ValueError, match='z_min'):
        F.PnPProblem(z(2, 8, 3), z(2, 8, 2), z(2, 8, 2), cam, HuberPnPCost(delta=1.0), 6)


def test_product_path_refuses_cpu_tensors():
    from epropnp import _hip
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import HuberPnPCost
    from epropnp.levenberg_marquardt import LMSolver
    import install as emu
    emu.uninstall()
    z = torch.zeros
    with pytest.raises(RuntimeError, match='HIP device'):
        LMSolver(dof=6, num_iter=1).solve(z(2, 8, 3), z(2, 8, 2), z(2, 8, 2), PerspectiveCamera(cam_mats=torch.eye(3).expand(2, 3, 3)),
                                          HuberPnPCost(), pose_init=z(2, 7))


def test_header_is_plain_c(tmp_path):
    """include/epropnp_hip.h is the drop-in boundary: it must compile as C99 (and C++) with nothing but <stdint.h>."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'abi.c'
    src.write_text('#include "epropnp_hip.h"\nint main(void) { return epropnp_abi_version() == EPROPNP_ABI_VERSION ? 0 : 1; }\n')
    for cmd in (['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror'], ['g++', '-std=c++11', '-Wall', '-Werror', '-x', 'c++']):
        r = subprocess.run(cmd + ['-I', os.path.join(root, 'include'), '-fsyntax-only', str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_c_program_links_and_runs(tmp_path, lib):
    """A C caller: include the header, link -lepropnp_hip, call the two entry points that need no device."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'epro-pnp_amd', 'lib')
    src = tmp_path / 'caller.c'
    src.write_text('#include <stdio.h>\n#include "epropnp_hip.h"\n'
                   'int main(void) { printf("%d %d %d\\n", epropnp_abi_version(), epropnp_noise_stride(6), epropnp_noise_stride(4));'
                   ' return 0; }\n')
    exe = tmp_path / 'caller'
    r = subprocess.run(['gcc', '-std=c99', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe), '-L', libdir,
                        '-lepropnp_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-L', '/opt/rocm/lib'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ['6', '8', '52']


def _build_standalone(tmp_path):
    import shutil
    import subprocess
    if shutil.which('g++') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime_api.h'):
        pytest.skip('g++ / HIP headers not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'epro-pnp_amd', 'lib')
    exe = tmp_path / 'standalone'
    r = subprocess.run(['g++', '-O2', '-std=c++17', '-I', os.path.join(root, 'include'), '-I', '/opt/rocm/include',
                        os.path.join(root, 'examples', 'standalone_c_abi.cpp'), '-o', str(exe), '-L', libdir,
                        '-lepropnp_hip', '-L', '/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,' + libdir,
                        '-Wl,-rpath,/opt/rocm/lib'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_standalone_program_builds(tmp_path, lib):
    """examples/standalone_c_abi.cpp (no torch, only hipMalloc + the C ABI) compiles with g++ and links the library."""
    _build_standalone(tmp_path)


@pytest.mark.gpu
def test_standalone_program_on_gpu(tmp_path, lib):
    """The torch-free host program runs the whole path (delta -> LM -> AMIS -> loss -> backward) on the device and
    checks its own result: LM never worse than the start, pose near the generating pose, all outputs finite."""
    import subprocess
    exe = _build_standalone(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'STANDALONE objects=64' in out.stdout and 'finite=1' in out.stdout


def _integration_stub():
    """The ctypes stub INTEGRATION.md shows a reference maintainer (section B), executed as written against the in-tree library."""
    import re
    from epropnp import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    block = [b for b in re.findall(r'```python\n(.*?)```', text, flags=re.S) if 'def lm_solve_hip' in b]
    assert len(block) == 1
    ns = {}
    exec(block[0].replace("C.CDLL('libepropnp_hip.so')", f'C.CDLL({_hip.LIB_PATH!r})'), ns)
    return ns


def test_integration_stub_matches_the_abi(lib):
    from epropnp import _hip
    ns = _integration_stub()
    for mine, doc in ((_hip.Problem, ns['_Problem']), (_hip.LmParams, ns['_LmParams'])):
        assert ctypes.sizeof(mine) == ctypes.sizeof(doc)
        assert [(n, getattr(mine, n).offset) for n, _ in mine._fields_] == [(n, getattr(doc, n).offset) for n, _ in doc._fields_]


@pytest.mark.gpu
def test_integration_stub_runs_lm_on_the_gpu():
    import install as emu
    import epropnp_oracle as orc
    from helpers import make_layer_objects
    from epropnp import functional as F
    from epropnp.levenberg_marquardt import LMSolver
    emu.uninstall()
    dev = torch.device('cuda:0')
    ns = _integration_stub()
    prob = orc.make_problem(6, 200, 6, seed=8)
    p, cam, cf = make_layer_objects(prob, dev)
    solver = LMSolver(dof=6, num_iter=4)
    pose, cov, cost = ns['lm_solve_hip'](solver, p['x3d'], p['x2d'], p['w2d'], cam, cf, p['pose_init'], True, True, False)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, 6)
    ref = F.lm_solve(hp, p['pose_init'], 4, with_pose_cov=True, with_cost=True)
    for a, b in zip((pose, cov, cost), ref):
        assert torch.equal(a, b)


def test_struct_layouts_of_the_header_match_the_ctypes_binding(tmp_path):
    """sizeof / offsetof of every struct in include/epropnp_hip.h, printed by a C program, against the ctypes Structures of
    epropnp/_hip.py (the C++ binding includes the header itself)."""
    import shutil
    import subprocess
    from epropnp import _hip
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    pairs = (('epropnp_problem', _hip.Problem), ('epropnp_lm_params', _hip.LmParams),
             ('epropnp_amis_params', _hip.AmisParams), ('epropnp_mc_params', _hip.McParams))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "epropnp_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines) + '\n')
    exe = tmp_path / 'layout'
    r = subprocess.run(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got[f'{cname}.{f}']) == getattr(cls, f).offset, (cname, f)
    assert _hip.torch_ext() is None or _hip.torch_ext().mc_params_size() == ctypes.sizeof(_hip.McParams)

"""pip install [-e] . [--no-build-isolation]   ->  importable `epropnp` package (the reference's module names).

The directory `epro-pnp_amd/` is not a Python identifier, so the mapping is declared here.  The HIP library is built for
gfx950 by epro-pnp_amd/build.py (hipcc; no GPU needed to compile) -- EPROPNP_OFFLOAD_ARCH overrides the architecture --
and, for a regular (non-editable) install, copied next to the package as epropnp/_lib/libepropnp_hip.so.
An editable install uses the in-tree epro-pnp_amd/lib/libepropnp_hip.so."""
import importlib.util
import os
import shutil

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


def _build_hip_library():
    spec = importlib.util.spec_from_file_location('epropnp_build', os.path.join(ROOT, 'epro-pnp_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(emu=False)


class BuildWithHip(build_py):
    def run(self):
        lib = _build_hip_library()
        super().run()
        dst = os.path.join(self.build_lib, 'epropnp', '_lib')
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(lib, dst)


setup(
    name='epropnp-hip',
    version='0.2.0',
    description='MI355X-native (HIP, gfx950) implementation of the EPro-PnP layer behind the reference API',
    packages=['epropnp'],
    package_dir={'epropnp': 'epro-pnp_amd/epropnp'},
    python_requires='>=3.9',
    install_requires=['torch'],
    cmdclass={'build_py': BuildWithHip},
)

"""Packaging of the MI355X-native drop-in for lucidrains/flash-cosine-sim-attention (reference: setup.py:1-77).

The reference builds one CUDAExtension with nvcc (setup.py:16-39).  Here the native parts are
  * libfcsa_hip.so   -- hand-written gfx950 HIP kernels + the C ABI (include/fcsa.h), built with hipcc, and
  * _fcsa_torch.so   -- the host-only PyTorch binding over that C ABI (csrc/fcsa_torch.cpp), built with g++,
both by flash_cosine_sim_attention_amd/csrc/Makefile (hipcc cross-compiles without a GPU).  `pip install .` /
`python setup.py build_ext --inplace` run that Makefile and ship the two shared objects as package data; nothing is
hipified, there is no CUDA path.
"""
import os
import subprocess
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = "flash_cosine_sim_attention_amd"


def build_native():
    # PYTHON: the binding must compile and link against the torch of the interpreter that runs this build (and will import the
    # package), not whatever `python3` is first on PATH.  Build with `pip install --no-build-isolation .` so that this IS the
    # ROCm torch of the environment (an isolated build env would pull a different torch wheel).
    subprocess.check_call(["make", "-C", os.path.join(HERE, PKG, "csrc"), "-j", str(min(8, os.cpu_count() or 1)), "PYTHON=" + sys.executable])


class BuildWithNative(build_py):
    def run(self):
        build_native()
        super().run()


try:
    from setuptools.command.build_ext import build_ext

    class BuildExt(build_ext):
        def run(self):
            build_native()
    cmdclass = {"build_py": BuildWithNative, "build_ext": BuildExt}
except ImportError:      # pragma: no cover
    cmdclass = {"build_py": BuildWithNative}

setup(
    name="flash-cosine-sim-attention-amd",
    version="0.3.0",
    description="Fused cosine-similarity attention for AMD MI355X (gfx950): hand-written HIP kernels behind the "
                "flash_cosine_sim_attention(q, k, v, ...) API",
    packages=[PKG],
    # include/fcsa.h: the package's own copy of the C-ABI header (`_lib.py` generates its ctypes structs from it at import time);
    # in the source tree it is a link to the repository's include/fcsa.h, the build copies the file
    package_data={PKG: ["libfcsa_hip.so", "_fcsa_torch.so", "csrc/*", "include/fcsa.h"]},
    python_requires=">=3.9",
    install_requires=["torch>=2.4"],
    cmdclass=cmdclass,
    zip_safe=False,
)

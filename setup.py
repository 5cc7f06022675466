"""In-tree install of the MI355X stepper (`pip install -e . --no-build-isolation`), the counterpart of the reference's setup.py.
The HIP library is built by hipcc for gfx950 at install time (mocca_envs_amd/build.py) and lives inside the package directory; an
sdist / wheel is not supported (the C ABI headers under include/ are part of the build)."""
import os
import sys

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_hip():
    sys.path.insert(0, HERE)
    from mocca_envs_amd.build import build_lib
    print("hipcc --offload-arch=gfx950 ->", build_lib())


class BuildPy(build_py):
    def run(self):
        _build_hip()
        super().run()


class Develop(develop):
    def run(self):
        _build_hip()
        super().run()


setup(
    name="mocca_envs_amd",
    version="0.3.0",
    description="MI355X-native vectorised locomotion stepper behind the mocca_envs gym surface",
    packages=["mocca_envs_amd"],
    package_data={"mocca_envs_amd": ["libmocca_hip.so", "data/*", "csrc/*"]},
    install_requires=["numpy", "torch"],
    python_requires=">=3.8",
    cmdclass={"build_py": BuildPy, "develop": Develop},
)

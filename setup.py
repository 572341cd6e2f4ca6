"""Packaging of the MI355X rasteriser: builds libsgs_hip.so (gfx950, hipcc via the csrc Makefile) and installs the
four Python packages the reference imports -- the counterpart of the reference's three CUDAExtension setup.py files
(submodules/channel-rasterization/setup.py:19-37, rgbd-rasterization, simple-knn).

    pip install --no-build-isolation .        # or:  pip install --no-build-isolation -e .

The shared library is a plain C-ABI .so loaded with ctypes (no torch extension module), so the build needs hipcc
and make only; `PYTORCH_ROCM_ARCH` is not consulted: the code is written for gfx950.
"""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "semantic-gaussians_amd")


class BuildWithHip(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j", str(os.cpu_count() or 4)])
        super().run()


setup(
    name="semantic-gaussians-amd",
    version="0.2.0",
    description="MI355X-native N-channel Gaussian-splat rasteriser (channel_rasterization / rgbd_rasterization / "
                "simple_knn drop-ins) for sharinka0715/semantic-gaussians",
    package_dir={"": "semantic-gaussians_amd"},
    packages=["sgs_hip", "channel_rasterization", "rgbd_rasterization", "simple_knn"],
    package_data={"sgs_hip": ["libsgs_hip.so"]},
    include_package_data=True,
    cmdclass={"build_py": BuildWithHip},
    python_requires=">=3.9",
    install_requires=[],          # torch (ROCm build) and numpy are expected in the environment
    zip_safe=False,
)

"""MI355X-native frame-synthesis hot path of DAIN / VFIDKR.

  lib/libvfi_hip.so   hand-written gfx950 HIP kernels behind the C ABI of include/vfi_hip.h
  ext/*.so            the reference's pybind11 module names (filterinterpolation_cuda, ...)
                      re-exported on top of that ABI
  cabi                ctypes caller of the C ABI (tests and bench call through it)
  my_package/, PWCNet/  host-side mirror of the reference's Layer/Module wrappers
  synthetic           seeded synthetic inputs of SURVEY.md section 8(d)

The directory name contains hyphens; import it through the repo-root alias
module `vfidkr_amd` (vfidkr_amd.py).  Importing the package makes the extension
module names importable (`import filterinterpolation_cuda`), as after the
reference's `python setup.py install`.

There is no CPU fallback anywhere in this package: a missing libvfi_hip.so or a
missing extension module is an ImportError/OSError, never a silent detour.
"""
import os
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
EXT_DIR = os.path.join(PKG_DIR, "ext")
LIB_PATH = os.path.join(PKG_DIR, "lib", "libvfi_hip.so")

if EXT_DIR not in sys.path:
    sys.path.insert(0, EXT_DIR)

__version__ = "0.1.0"

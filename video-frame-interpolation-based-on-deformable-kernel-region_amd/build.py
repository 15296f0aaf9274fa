"""In-tree builders for the two native artefacts of the package.

  lib/libvfi_hip.so                hand-written HIP kernels + the C ABI of include/vfi_hip.h
                                   (hipcc --offload-arch=gfx950, csrc/Makefile)
  ext/<module>.cpython-*.so        the reference's nine pybind11 module names on top of the
                                   C ABI (csrc/shim/vfi_torch_shim.cpp, plain g++ against the
                                   torch headers; no kernels, no hipify)

Both are built in-tree (nothing goes to ~/.cache or site-packages) so the .so
files travel with a snapshot of the repository.
"""
import os
import shutil
import subprocess
import sys
import sysconfig

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
EXT_DIR = os.path.join(PKG_DIR, "ext")
LIB_PATH = os.path.join(LIB_DIR, "libvfi_hip.so")

SHIM_MODULES = (
    "filterinterpolation_cuda", "flowprojection_cuda", "depthflowprojection_cuda", "mindepthflowprojection_cuda",
    "interpolation_cuda",
    "interpolationch_cuda", "separableconv_cuda", "separableconvflow_cuda", "correlation_cuda",
)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip_library(force=False, jobs=4):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def shim_path(name):
    return os.path.join(EXT_DIR, name + sysconfig.get_config_var("EXT_SUFFIX"))


def build_torch_shim(force=False):
    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(CSRC, "shim", "vfi_torch_shim.cpp")
    hdr = os.path.join(os.path.dirname(PKG_DIR), "include", "vfi_hip.h")
    os.makedirs(EXT_DIR, exist_ok=True)
    master = os.path.join(EXT_DIR, "_vfi_torch_shim.so")
    if force or _newer(master, [src, hdr]):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        tlib = ce.library_paths()[0]
        cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-function", src, "-o", master,
               "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
        for inc in ce.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"]]:
            cmd += ["-isystem", inc]
        cmd += ["-L" + tlib, "-L" + LIB_DIR, "-lvfi_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
                "-ltorch", "-ltorch_python",
                "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + tlib]
        subprocess.check_call(cmd)
    for name in SHIM_MODULES:
        dst = shim_path(name)
        if force or _newer(dst, [master]):
            shutil.copyfile(master, dst)
    return [shim_path(n) for n in SHIM_MODULES]


def build_all(force=False):
    build_hip_library(force=force)
    build_torch_shim(force=force)
    # ext/ may not have existed when the package put it on sys.path: drop the import system's cached "nothing there"
    import importlib
    importlib.invalidate_caches()
    sys.path_importer_cache.pop(EXT_DIR, None)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", LIB_PATH)

"""Import alias for the package directory
`video-frame-interpolation-based-on-deformable-kernel-region_amd/` (hyphens are
not importable): `import vfidkr_amd` yields that package, with submodules
(`vfidkr_amd.cabi`, `vfidkr_amd.my_package.FilterInterpolation`, ...)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "video-frame-interpolation-based-on-deformable-kernel-region_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

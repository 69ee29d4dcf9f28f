"""offline_raytracer_amd -- MI355X-native drop-in for the path-trace render call of
gyuhyun-lee/offline_raytracer.  The product is the C-ABI library
offline_raytracer_amd/lib/libort.so (HIP kernels + C++ host side, sources in csrc/);
this package is the thin ctypes binding used by the tests, bench.py and launch scripts."""
from . import api  # noqa: F401
from .api import Scene, OrtError  # noqa: F401

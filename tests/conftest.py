import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

# the library reads its developer knobs (ORT_EXCHANGE, ORT_KERNEL, ...) once per uploaded scene; the tests flip them between
# two renders of one cached scene, so they ask for the environment to be read at every render
os.environ.setdefault("ORT_KNOBS_LIVE", "1")

GOLDEN = os.path.join(HERE, "golden")
DATA = os.path.join(ROOT, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="session")
def api():
    """The product binding; the library must already be built (build() / make)."""
    from offline_raytracer_amd import api as _api
    if not os.path.exists(_api.LIB_PATH):
        _api.build_library()
    _api.lib()
    return _api


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


_scene_cache = {}


@pytest.fixture(scope="session")
def load_scene(api, tmp_path_factory):
    """name -> committed product scene (cached per session).  "c5_heightfield_<n>" (BASELINE.json's
    synthetic-mesh stress config) is generated on the spot by tools/make_heightfield.py."""
    def _load(name):
        if name not in _scene_cache:
            if name.startswith("c5_heightfield_"):
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import make_heightfield
                scn, _, _ = make_heightfield.write_scene(int(name.rsplit("_", 1)[1]), str(tmp_path_factory.mktemp("c5")))
            else:
                scn = os.path.join(DATA, name + ".scn")
            _scene_cache[name] = api.Scene.load_scn(scn).commit()
        return _scene_cache[name]
    return _load


@pytest.fixture(scope="session")
def gpu_scene(api, load_scene):
    def _load(name):
        s = load_scene(name)
        if s.device is None:
            if api.device_count() < 1:
                n = api.C.c_int(0)
                rc = api.lib().ort_device_count(api.C.byref(n))
                pytest.fail("GPU test on a machine without a HIP device (the render path has no CPU fallback): "
                            "ort_device_count rc %d, %r" % (rc, api.lib().ort_last_error()))
            s.upload(0)
        return s
    return _load


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype="<f4")
    b = np.ascontiguousarray(b, dtype="<f4")
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    ne = a.view("<u4") != b.view("<u4")
    if ne.any():
        idx = np.argwhere(ne)[0]
        raise AssertionError("%s: %d of %d values differ bitwise; first at %s: %r vs %r"
                             % (what, int(ne.sum()), ne.size, tuple(idx), a[tuple(idx)], b[tuple(idx)]))

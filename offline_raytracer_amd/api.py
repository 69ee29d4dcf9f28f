"""ctypes binding of libort.so -- the C ABI of include/ort.h.

Python here is plumbing (tests, bench.py, torch.distributed launch); the host side of the
product is C++ (offline_raytracer_amd/csrc) and the compute is HIP.  Every render entry
point raises unless the HIP library is built and the scene is resident on a GPU: there is
no CPU fallback for the render call.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORT_LIB", os.path.join(PKG_DIR, "lib", "libort.so"))  # ORT_LIB: tuning builds only
CSRC_DIR = os.path.join(PKG_DIR, "csrc")

OK, ERR_INVALID, ERR_IO, ERR_PARSE, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE, ERR_NO_MEMORY, ERR_INTERNAL = range(10)
POLICY_TILE32, POLICY_WHOLE, POLICY_PIXEL, POLICY_CHUNK = range(4)
POLICIES = {"tile32": POLICY_TILE32, "whole": POLICY_WHOLE, "pixel": POLICY_PIXEL, "chunk": POLICY_CHUNK}
RENDER_COUNTERS = 1
RENDER_PACKED = 2
COMM_ID_BYTES = 128


class OrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ort error %d: %s" % (code, message))
        self.code = code


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 3), ("specular", "<f4", 4), ("transmission", "<f4", 3),
                           ("ior", "<f4"), ("emit", "<f4", 3), ("is_light", "<i4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("r", "<f4"), ("mat", "<u4")])
BOX_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("mat", "<u4")])
CYLINDER_DTYPE = np.dtype([("base", "<f4", 3), ("axis", "<f4", 3), ("r", "<f4"), ("mat", "<u4")])
LIGHT_DTYPE = np.dtype([("type", "<u4"), ("index", "<u4")])
JOB_DTYPE = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("rng_state", "<u4"),
                      ("spp", "<u4")])


class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint32), ("indices", C.c_void_p),
                ("index_count", C.c_uint32), ("mat", C.c_uint32), ("aabb_min", V3), ("aabb_max", V3)]


class SceneDesc(C.Structure):
    _fields_ = [("materials", C.c_void_p), ("material_count", C.c_uint32),
                ("spheres", C.c_void_p), ("sphere_count", C.c_uint32),
                ("boxes", C.c_void_p), ("box_count", C.c_uint32),
                ("cylinders", C.c_void_p), ("cylinder_count", C.c_uint32),
                ("meshes", C.c_void_p), ("mesh_count", C.c_uint32),
                ("lights", C.c_void_p), ("light_count", C.c_uint32),
                ("camera_p", V3), ("camera_quat_xyzw", C.c_float * 4), ("camera_height_ratio", C.c_float),
                ("screen_width", C.c_int32), ("screen_height", C.c_int32), ("ambient", V3),
                ("with_reference_csg", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [("material_count", C.c_uint32), ("sphere_count", C.c_uint32), ("box_count", C.c_uint32),
                ("cylinder_count", C.c_uint32), ("mesh_count", C.c_uint32), ("light_count", C.c_uint32),
                ("triangle_count", C.c_uint32), ("screen_width", C.c_int32), ("screen_height", C.c_int32),
                ("ambient", V3), ("camera_p", V3), ("camera_quat_xyzw", C.c_float * 4),
                ("camera_height_ratio", C.c_float)]


class TreeInfo(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("leaf_count", C.c_uint32), ("max_leaf_prims", C.c_uint32),
                ("max_depth", C.c_uint32), ("node_bytes", C.c_uint64), ("prim_bytes", C.c_uint64),
                ("sah_cost", C.c_float), ("ref_node_count", C.c_uint32), ("ref_nonempty_leaves", C.c_uint32),
                ("ref_max_leaf_records", C.c_uint32), ("ref_bytes", C.c_uint64), ("prologue_prims", C.c_uint32),
                ("wide_node_count", C.c_uint32), ("wide_max_depth", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("analytic_tests", C.c_uint64), ("fallback_rays", C.c_uint64), ("kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32), ("policy", C.c_int32), ("seed", C.c_uint32),
                ("spp", C.c_uint32), ("chunk", C.c_uint32), ("rr", C.c_float), ("flags", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("p", V3), ("x_axis", V3), ("y_axis", V3), ("z_axis", V3)]


# every symbol include/ort.h declares
EXPORTS = [
    "ort_last_error", "ort_abi_version", "ort_scene_load_scn", "ort_scene_parse_scn", "ort_scene_create",
    "ort_scene_destroy", "ort_scene_get_info", "ort_scene_get_materials", "ort_scene_get_spheres",
    "ort_scene_get_boxes", "ort_scene_get_cylinders", "ort_scene_get_lights", "ort_scene_get_mesh",
    "ort_scene_get_camera", "ort_scene_commit", "ort_scene_get_tree_info", "ort_device_count", "ort_scene_upload",
    "ort_tiled_raytrace", "ort_tiled_raytrace_batch", "ort_render_image", "ort_render_image_device",
    "ort_render_workspace_bytes", "ort_unit_eval_device", "ort_rgbe", "ort_write_hdr",
    "ort_shard_block_count", "ort_pack_blocks_host", "ort_unpack_blocks_host", "ort_unpack_blocks_device",
    "ort_comm_unique_id", "ort_comm_create", "ort_comm_create_local", "ort_comm_destroy", "ort_gather_framebuffer",
    "ort_gather_framebuffer_local"]

_lib = None


def build_library():
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    subprocess.check_call(["make", "-s", "-C", CSRC_DIR])


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  libort.so needs libamdhip64.so.7 and finds the system's (/opt/rocm); PyTorch ships
    its own copy under torch/lib and asks for it by file name, so if libort.so is loaded BEFORE torch the process ends
    up with two runtimes and the second one to initialise finds "no ROCm-capable device".  Loading torch's copy first
    (without importing torch) makes both resolve to the same library whatever the import order.  Only where torch is
    installed and only for this Python binding; the C++ driver uses the system runtime.  ORT_SYSTEM_HIP=1 opts out."""
    if os.environ.get("ORT_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: without it the import order decides, as before
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrtError(ERR_STATE, "libort.so is not built (run offline_raytracer_amd.api.build_library() or "
                                      "make -C offline_raytracer_amd/csrc); there is no fallback implementation")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        L.ort_last_error.restype = C.c_char_p
        L.ort_scene_load_scn.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ort_scene_parse_scn.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ort_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.ort_scene_destroy.argtypes = [C.c_void_p]
        L.ort_scene_destroy.restype = None
        L.ort_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        for name in ("materials", "spheres", "boxes", "cylinders", "lights"):
            getattr(L, "ort_scene_get_" + name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ort_scene_get_mesh.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Mesh)]
        L.ort_scene_get_camera.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Camera)]
        L.ort_scene_commit.argtypes = [C.c_void_p]
        L.ort_scene_get_tree_info.argtypes = [C.c_void_p, C.POINTER(TreeInfo)]
        L.ort_device_count.argtypes = [C.POINTER(C.c_int)]
        L.ort_scene_upload.argtypes = [C.c_void_p, C.c_int]
        L.ort_tiled_raytrace.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_uint32, C.c_float,
                                         C.POINTER(C.c_uint64)]
        L.ort_tiled_raytrace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32,
                                               C.c_float, C.c_void_p, C.POINTER(Stats)]
        L.ort_render_image.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
        L.ort_render_image_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                              C.POINTER(Stats)]
        L.ort_render_workspace_bytes.argtypes = [C.POINTER(RenderParams), C.POINTER(C.c_uint64)]
        L.ort_unit_eval_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ort_rgbe.restype = C.c_uint32
        L.ort_rgbe.argtypes = [C.c_float, C.c_float, C.c_float]
        L.ort_write_hdr.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ort_shard_block_count.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.ort_pack_blocks_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ort_unpack_blocks_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ort_unpack_blocks_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.ort_comm_unique_id.argtypes = [C.c_void_p]
        L.ort_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.ort_comm_create_local.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.ort_comm_destroy.argtypes = [C.c_void_p]
        L.ort_comm_destroy.restype = None
        L.ort_gather_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ort_gather_framebuffer_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_int32,
                                                   C.c_int32, C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        raise OrtError(rc, lib().ort_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    rc = lib().ort_device_count(C.byref(n))
    return n.value if rc == OK else 0


def _v3(a):
    return V3(float(a[0]), float(a[1]), float(a[2]))


class Scene:
    """Owning wrapper of an ort_scene handle."""

    def __init__(self, handle):
        self.handle = C.c_void_p(handle)
        self.device = None

    # -- construction ----------------------------------------------------------------
    @classmethod
    def load_scn(cls, path, base_dir=None):
        if base_dir is None:
            base_dir = os.path.dirname(os.path.abspath(path)) + "/"
        h = C.c_void_p()
        _check(lib().ort_scene_load_scn(os.fsencode(path), os.fsencode(base_dir), C.byref(h)))
        return cls(h.value)

    @classmethod
    def parse_scn(cls, text, base_dir=""):
        data = text if isinstance(text, bytes) else text.encode()
        h = C.c_void_p()
        _check(lib().ort_scene_parse_scn(data, len(data), os.fsencode(base_dir), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_arrays(cls, materials, spheres=None, boxes=None, cylinders=None, lights=None, meshes=(),
                    camera_p=(0, 0, 0), camera_quat_xyzw=(0, 0, 0, 1), camera_height_ratio=0.2, screen=(0, 0),
                    ambient=(0, 0, 0), with_reference_csg=False):
        def arr(a, dt):
            return np.ascontiguousarray(a if a is not None else np.zeros(0, dt), dtype=dt)
        mats, sph, box = arr(materials, MATERIAL_DTYPE), arr(spheres, SPHERE_DTYPE), arr(boxes, BOX_DTYPE)
        cyl, lig = arr(cylinders, CYLINDER_DTYPE), arr(lights, LIGHT_DTYPE)
        keep = []
        ms = (Mesh * max(1, len(meshes)))()
        for i, m in enumerate(meshes):
            v = np.ascontiguousarray(m["vertices"], dtype="<f4").reshape(-1, 3)
            ix = np.ascontiguousarray(m["indices"], dtype="<u4")
            keep += [v, ix]
            lo = m.get("aabb_min", v.min(axis=0) if len(v) else (0, 0, 0))
            hi = m.get("aabb_max", v.max(axis=0) if len(v) else (0, 0, 0))
            ms[i] = Mesh(v.ctypes.data, len(v), ix.ctypes.data, len(ix), int(m["mat"]), _v3(lo), _v3(hi))
        d = SceneDesc(mats.ctypes.data, len(mats), sph.ctypes.data, len(sph), box.ctypes.data, len(box),
                      cyl.ctypes.data, len(cyl), C.addressof(ms), len(meshes), lig.ctypes.data, len(lig),
                      _v3(camera_p), (C.c_float * 4)(*[float(q) for q in camera_quat_xyzw]),
                      float(camera_height_ratio), int(screen[0]), int(screen[1]), _v3(ambient),
                      1 if with_reference_csg else 0)
        h = C.c_void_p()
        _check(lib().ort_scene_create(C.byref(d), C.byref(h)))
        return cls(h.value)

    def close(self):
        if self.handle:
            lib().ort_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inspection ------------------------------------------------------------------
    def info(self):
        si = SceneInfo()
        _check(lib().ort_scene_get_info(self.handle, C.byref(si)))
        return si

    def _get(self, name, dtype, count):
        out = np.zeros(count, dtype)
        _check(getattr(lib(), "ort_scene_get_" + name)(self.handle, out.ctypes.data, count))
        return out

    def flatten(self, width, height):
        """The scene as plain arrays (same field layout as tests/ref_io.SceneDump)."""
        si = self.info()

        class Flat:
            pass
        f = Flat()
        f.width, f.height = width, height
        f.ambient = np.array([si.ambient.x, si.ambient.y, si.ambient.z], "<f4")
        f.materials = self._get("materials", MATERIAL_DTYPE, si.material_count)
        f.spheres = self._get("spheres", SPHERE_DTYPE, si.sphere_count)
        f.boxes = self._get("boxes", BOX_DTYPE, si.box_count)
        f.cylinders = self._get("cylinders", CYLINDER_DTYPE, si.cylinder_count)
        f.lights = self._get("lights", LIGHT_DTYPE, si.light_count)
        f.meshes = []
        for i in range(si.mesh_count):
            m = Mesh()
            _check(lib().ort_scene_get_mesh(self.handle, i, C.byref(m)))
            v = np.ctypeslib.as_array(C.cast(m.vertices, C.POINTER(C.c_float)), shape=(m.vertex_count, 3)).copy() \
                if m.vertex_count else np.zeros((0, 3), "<f4")
            ix = np.ctypeslib.as_array(C.cast(m.indices, C.POINTER(C.c_uint32)), shape=(m.index_count,)).copy() \
                if m.index_count else np.zeros(0, "<u4")
            f.meshes.append(dict(vertices=v.astype("<f4"), indices=ix.astype("<u4"), mat=m.mat,
                                 aabb_min=np.array([m.aabb_min.x, m.aabb_min.y, m.aabb_min.z], "<f4"),
                                 aabb_max=np.array([m.aabb_max.x, m.aabb_max.y, m.aabb_max.z], "<f4")))
        f.camera = self.camera(width, height)
        return f

    def camera(self, width, height):
        cam = Camera()
        _check(lib().ort_scene_get_camera(self.handle, width, height, C.byref(cam)))
        return np.array([[v.x, v.y, v.z] for v in (cam.p, cam.x_axis, cam.y_axis, cam.z_axis)], dtype="<f4")

    # -- build / upload -----------------------------------------------------------------
    def commit(self):
        _check(lib().ort_scene_commit(self.handle))
        return self

    def tree_info(self):
        ti = TreeInfo()
        _check(lib().ort_scene_get_tree_info(self.handle, C.byref(ti)))
        return {k: getattr(ti, k) for k, _ in ti._fields_}

    def upload(self, device=0):
        _check(lib().ort_scene_upload(self.handle, device))
        self.device = device
        return self

    # -- render ---------------------------------------------------------------------------
    @staticmethod
    def params(width, height, spp, seed, policy="chunk", chunk=0, rect=None, rr=0.8, counters=False, shard=(0, 1), packed=False):
        x0, y0, x1, y1 = rect if rect else (0, 0, width, height)
        pol = POLICIES[policy] if isinstance(policy, str) else policy
        if pol == POLICY_CHUNK and not chunk:
            chunk = spp
        return RenderParams(width, height, x0, y0, x1, y1, pol, seed & 0xFFFFFFFF, spp, chunk, rr,
                            (RENDER_COUNTERS if counters else 0) | (RENDER_PACKED if packed else 0), shard[0], shard[1])

    def render(self, width, height, spp, seed, policy="chunk", chunk=0, rect=None, rr=0.8, counters=False,
               shard=(0, 1), out=None):
        """Host framebuffer in/out.  Returns (image[H,W,3] float32, stats dict)."""
        p = self.params(width, height, spp, seed, policy, chunk, rect, rr, counters, shard)
        if out is None:
            out = np.zeros((height, width, 3), dtype="<f4")
        st = Stats()
        _check(lib().ort_render_image(self.handle, C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def render_device(self, d_out_ptr, params, stream=None, want_stats=False):
        """Device framebuffer (raw device pointer, e.g. torch_tensor.data_ptr())."""
        st = Stats() if want_stats else None
        _check(lib().ort_render_image_device(self.handle, C.byref(params), C.c_void_p(d_out_ptr),
                                             C.c_void_p(stream) if stream else None,
                                             C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def tiled_raytrace(self, out, x0, y0, x1, y1, rng_state, spp, rr=0.8):
        """Exact analogue of one reference call (ray.cpp:1178); returns (shape_tests, new_rng_state)."""
        height, width = out.shape[:2]
        st = C.c_uint32(rng_state)
        n = C.c_uint64(0)
        _check(lib().ort_tiled_raytrace(self.handle, out.ctypes.data, width, height, x0, y0, x1, y1, C.byref(st), spp,
                                        rr, C.byref(n)))
        return n.value, st.value

    def tiled_raytrace_batch(self, out, jobs, rr=0.8):
        height, width = out.shape[:2]
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        finals = np.zeros(len(jobs), "<u4")
        st = Stats()
        _check(lib().ort_tiled_raytrace_batch(self.handle, out.ctypes.data, width, height, jobs.ctypes.data, len(jobs),
                                              rr, finals.ctypes.data, C.byref(st)))
        return finals, st.as_dict()


def unit_eval_device(records, device=0):
    """records: structured array {op: u4, a: f4[24]} -> (n, 8) float32, evaluated by the HIP device functions."""
    records = np.ascontiguousarray(records)
    assert records.dtype.itemsize == 100
    out = np.zeros((len(records), 8), "<f4")
    _check(lib().ort_unit_eval_device(device, records.ctypes.data, len(records), out.ctypes.data))
    return out


def rgbe(r, g, b):
    return lib().ort_rgbe(r, g, b)


def write_hdr(path, image):
    image = np.ascontiguousarray(image, dtype="<f4")
    h, w = image.shape[:2]
    _check(lib().ort_write_hdr(os.fsencode(path), image.ctypes.data, w, h))


def workspace_bytes(params):
    n = C.c_uint64(0)
    _check(lib().ort_render_workspace_bytes(C.byref(params), C.byref(n)))
    return n.value


# ---- multi-GPU: block sharding and the one collective (ort_comm.cpp) -----------------------------
def shard_block_count(width, height, index, count):
    n = C.c_uint64(0)
    _check(lib().ort_shard_block_count(width, height, index, count, C.byref(n)))
    return n.value


def pack_blocks_host(image, index, count):
    """[H, W, 3] float32 -> this shard's packed blocks [n_blocks, 64, 3] (CPU)."""
    image = np.ascontiguousarray(image, dtype="<f4")
    h, w = image.shape[:2]
    out = np.zeros((shard_block_count(w, h, index, count), 64, 3), "<f4")
    _check(lib().ort_pack_blocks_host(image.ctypes.data, w, h, index, count, out.ctypes.data))
    return out


def unpack_blocks_host(packed, width, height, index, count, out=None):
    packed = np.ascontiguousarray(packed, dtype="<f4")
    if out is None:
        out = np.zeros((height, width, 3), "<f4")
    _check(lib().ort_unpack_blocks_host(packed.ctypes.data, width, height, index, count, out.ctypes.data))
    return out


def unpack_blocks_device(d_packed_ptr, width, height, index, count, d_full_ptr, stream=None):
    _check(lib().ort_unpack_blocks_device(C.c_void_p(d_packed_ptr), width, height, index, count, C.c_void_p(d_full_ptr),
                                          C.c_void_p(stream) if stream else None))


class Comm:
    """One rank's handle on the gather (RCCL underneath for world > 1, loaded on first use)."""

    def __init__(self, handle):
        self.handle = handle

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        _check(lib().ort_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, unique_id, rank, world, device):
        h = C.c_void_p()
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id) if unique_id else None
        _check(lib().ort_comm_create(buf, rank, world, device, C.byref(h)))
        return cls(h.value)

    def gather(self, d_packed_ptr, d_full_ptr, width, height, stream=None):
        _check(lib().ort_gather_framebuffer(self.handle, C.c_void_p(d_packed_ptr), C.c_void_p(d_full_ptr) if d_full_ptr else None,
                                            width, height, C.c_void_p(stream) if stream else None))

    def close(self):
        if self.handle:
            lib().ort_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

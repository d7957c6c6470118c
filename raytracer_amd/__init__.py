"""ctypes view of librt_amd.so, the MI355X path tracer.

The product is the shared library (HIP kernels + C++ host scene layer, built by
``raytracer_amd/csrc/Makefile``); this module only binds its C ABI
(``include/rt_amd.h``) for the test-suite and the benchmark.  ``Scene`` mirrors
the public interface of the reference's ``class Scene`` (reference
include/scene.cuh:8-31).  There is no CPU fallback: if the library is missing
the import fails.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: another build of the library (tuning experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")
REPO_ROOT = os.path.dirname(_HERE)
ASSETS = os.path.join(REPO_ROOT, "assets")

RT_OK = 0
RT_ERR_INVALID, RT_ERR_DEVICE, RT_ERR_IO, RT_ERR_PARSE, RT_ERR_NOMEM = -1, -2, -3, -4, -5
DIFFUSE, MIRROR, GLASS = 0, 1, 2

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


class SceneDesc(C.Structure):
    """rt_scene_desc"""
    _fields_ = [
        ("n_spheres", C.c_uint32), ("sphere_pos", _f32p), ("sphere_radius", _f32p), ("sphere_material", _i32p),
        ("n_materials", C.c_uint32), ("material_type", _i32p), ("material_color", _f32p),
        ("material_ro", _f32p), ("material_ri", _f32p),
        ("n_vertices", C.c_uint32), ("vertices", _f32p), ("normals", _f32p),
        ("n_triangles", C.c_uint32), ("tri_v", _i32p), ("tri_n", _i32p), ("tri_material", _i32p),
        ("n_boxes", C.c_uint32), ("box_min", _f32p), ("box_max", _f32p),
        ("n_lights", C.c_uint32), ("light_pos", _f32p), ("light_intensity", _f32p),
    ]


class Camera(C.Structure):
    """rt_camera"""
    _fields_ = [
        ("rot", C.c_float * 9), ("pos", C.c_float * 3),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("width", C.c_uint32), ("height", C.c_uint32), ("num_rays", C.c_uint32),
    ]


class Options(C.Structure):
    """rt_options"""
    _fields_ = [
        ("indirect", C.c_int32), ("antialiasing", C.c_int32), ("bounding_box", C.c_int32),
        ("max_depth", C.c_int32), ("seed", C.c_uint32), ("profile", C.c_int32),
        ("cluster_culling", C.c_int32), ("arithmetic", C.c_int32),
    ]


class Stats(C.Structure):
    """rt_stats"""
    _fields_ = [
        ("gpu_ms", C.c_double), ("mesh_kernel_ms", C.c_double), ("bin_kernel_ms", C.c_double),
        ("mesh_kernel_launches", C.c_uint32), ("batches", C.c_uint32),
        ("paths", C.c_uint64), ("mesh_rays", C.c_uint64), ("triangle_tests", C.c_uint64),
        ("sphere_casts", C.c_uint64), ("shade_kernel_ms", C.c_double), ("shade_kernel_launches", C.c_uint32),
        ("reserved", C.c_uint32), ("path_bounces", C.c_uint64), ("path_records", C.c_uint64),
    ]


def make_options(indirect=True, antialiasing=True, bounding_box=True, max_depth=5, seed=1, profile=False,
                 cluster_culling=True, arithmetic=0) -> Options:
    """arithmetic: 0 = the reference's float arithmetic bit for bit, 1 = fused multiply-adds and hardware reciprocals / roots."""
    return Options(int(indirect), int(antialiasing), int(bounding_box), int(max_depth), int(seed) & 0xFFFFFFFF, int(profile),
                   int(cluster_culling), int(arithmetic))


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt_amd error {code}: {message}")
        self.code = code


def _preload_torch_hip_runtime() -> None:
    """PyTorch's ROCm wheels bundle their own libamdhip64.so with the same soname as the
    system one; whichever is loaded first serves the whole process, and PyTorch finds no
    GPU when it is not its own.  So when PyTorch is installed its copy is loaded first
    (without importing torch), which makes the import order of the two packages
    irrelevant.  RT_AMD_SYSTEM_HIP=1 keeps the system runtime."""
    if os.environ.get("RT_AMD_SYSTEM_HIP") == "1":
        return
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    candidate = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(candidate):
        try:
            C.CDLL(candidate, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint64, C.c_uint64, C.c_void_p)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sig = {
        "rt_init": [C.c_int32],
        "rt_scene_upload": [C.POINTER(SceneDesc), C.POINTER(vp)],
        "rt_scene_free": [vp],
        "rt_render": [vp, C.POINTER(Camera), C.POINTER(Options), _f32p, C.POINTER(Stats)],
        "rt_accumulate": [vp, C.POINTER(Camera), C.POINTER(Options), C.c_uint32, C.c_uint32, vp, vp, C.POINTER(Stats)],
        "rt_accumulate_rows": [vp, C.POINTER(Camera), C.POINTER(Options), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp,
                               C.POINTER(Stats)],
        "rt_accumulate_static": [vp, C.POINTER(Camera), C.POINTER(Options), C.c_uint32, C.c_uint32, vp, vp],
        "rt_resolve": [vp, C.c_uint32, vp, vp],
        "rt_add_sums": [vp, vp, C.c_uint32, vp],
        "rt_device_group_create": [C.POINTER(SceneDesc), _i32p, C.c_uint32, C.POINTER(vp)],
        "rt_device_group_render": [vp, C.POINTER(Camera), C.POINTER(Options), _f32p, C.POINTER(Stats)],
        "rt_device_group_free": [vp],
        "rt_host_scene_set_devices": [vp, C.c_uint32],
        "rt_cast": [vp, C.POINTER(Options), C.c_uint32, _f32p, _f32p, _f32p, _i32p, _i32p, _f32p],
        "rt_visible": [vp, C.POINTER(Options), C.c_uint32, _f32p, _f32p, _i32p],
        "rt_random_cos": [C.c_uint32, _f32p, _f32p, _f32p, _f32p],
        "rt_host_scene_create": [C.POINTER(vp)],
        "rt_host_scene_destroy": [vp],
        "rt_host_scene_load": [vp, C.c_char_p],
        "rt_host_scene_set_camera_intrinsics": [vp, C.c_float, C.c_size_t, C.c_size_t],
        "rt_host_scene_rotate_camera": [vp, C.c_float, C.c_float, C.c_float],
        "rt_host_scene_transform_camera": [vp, C.c_float, C.c_float, C.c_float],
        "rt_host_scene_set_num_rays": [vp, C.c_uint32],
        "rt_host_scene_add_sphere": [vp, _f32p, C.c_float, _f32p, C.c_int32, C.c_float, C.c_float],
        "rt_host_scene_add_object": [vp, C.c_char_p, _f32p, _f32p, _f32p, _f32p, C.c_int32, C.c_float, C.c_float],
        "rt_host_scene_add_light": [vp, _f32p, C.c_float],
        "rt_host_scene_set_options": [vp, C.POINTER(Options)],
        "rt_host_scene_render": [vp, C.c_char_p, C.c_size_t],
        "rt_host_scene_describe": [vp, C.POINTER(SceneDesc), C.POINTER(Camera)],
        "rt_write_png": [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)],
        "rt_write_pfm": [C.c_char_p, C.c_uint32, C.c_uint32, _f32p],
        "rt_set_progress": [vp, PROGRESS_FN, vp],
        "rt_host_scene_render_frames": [vp, C.c_char_p, C.c_size_t, C.c_uint32, _f32p, _f32p, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.POINTER(C.c_double)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.rt_frame_partition.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32 * 4)]
    lib.rt_frame_partition.restype = None
    lib.rt_last_error.argtypes = []
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_host_scene_image.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rt_host_scene_image.restype = _f32p
    return lib


lib = _load()

# every symbol include/rt_amd.h declares
EXPORTS = [
    "rt_init", "rt_last_error", "rt_scene_upload", "rt_scene_free", "rt_render", "rt_accumulate", "rt_resolve",
    "rt_cast", "rt_visible", "rt_host_scene_create", "rt_host_scene_destroy", "rt_host_scene_load",
    "rt_host_scene_set_camera_intrinsics", "rt_host_scene_rotate_camera", "rt_host_scene_transform_camera",
    "rt_host_scene_set_num_rays", "rt_host_scene_add_sphere", "rt_host_scene_add_object", "rt_host_scene_add_light",
    "rt_host_scene_set_options", "rt_host_scene_render", "rt_host_scene_describe", "rt_host_scene_image",
    "rt_write_png", "rt_write_pfm", "rt_set_progress", "rt_host_scene_render_frames", "rt_accumulate_rows",
    "rt_accumulate_static", "rt_add_sums", "rt_frame_partition", "rt_device_group_create", "rt_device_group_render",
    "rt_device_group_free", "rt_host_scene_set_devices", "rt_random_cos", "rt_profile_collect",
]


def _check(code: int) -> None:
    if code != RT_OK:
        raise RtError(code, (lib.rt_last_error() or b"").decode(errors="replace"))


def _vec3(v) -> "C.Array[C.c_float]":
    a = np.asarray(v, dtype=np.float32).reshape(3)
    return (C.c_float * 3)(*a.tolist())


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _iptr(a: np.ndarray):
    return a.ctypes.data_as(_i32p)


def desc_arrays(desc: SceneDesc) -> dict:
    """Copies of the arrays a SceneDesc points at, as numpy arrays."""
    def f(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    def i(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.int32)

    return {
        "sphere_pos": f(desc.sphere_pos, 3 * desc.n_spheres).reshape(-1, 3),
        "sphere_radius": f(desc.sphere_radius, desc.n_spheres),
        "sphere_material": i(desc.sphere_material, desc.n_spheres),
        "material_type": i(desc.material_type, desc.n_materials),
        "material_color": f(desc.material_color, 3 * desc.n_materials).reshape(-1, 3),
        "material_ro": f(desc.material_ro, desc.n_materials),
        "material_ri": f(desc.material_ri, desc.n_materials),
        "vertices": f(desc.vertices, 3 * desc.n_vertices).reshape(-1, 3),
        "normals": f(desc.normals, 3 * desc.n_vertices).reshape(-1, 3),
        "tri_v": i(desc.tri_v, 3 * desc.n_triangles).reshape(-1, 3),
        "tri_n": i(desc.tri_n, 3 * desc.n_triangles).reshape(-1, 3),
        "tri_material": i(desc.tri_material, desc.n_triangles),
        "box_min": f(desc.box_min, 3 * desc.n_boxes).reshape(-1, 3),
        "box_max": f(desc.box_max, 3 * desc.n_boxes).reshape(-1, 3),
        "light_pos": f(desc.light_pos, 3 * desc.n_lights).reshape(-1, 3),
        "light_intensity": f(desc.light_intensity, desc.n_lights),
    }


def desc_from_arrays(a: dict) -> Tuple[SceneDesc, list]:
    """Builds a SceneDesc over numpy arrays; returns (desc, keepalive)."""
    keep = {}

    def f(key, width=None):
        arr = np.ascontiguousarray(a.get(key, np.zeros(0)), dtype=np.float32)
        keep[key] = arr
        return arr

    def i(key):
        arr = np.ascontiguousarray(a.get(key, np.zeros(0)), dtype=np.int32)
        keep[key] = arr
        return arr

    d = SceneDesc()
    d.n_spheres = len(f("sphere_radius"))
    d.sphere_pos, d.sphere_radius, d.sphere_material = _fptr(f("sphere_pos")), _fptr(keep["sphere_radius"]), _iptr(i("sphere_material"))
    d.n_materials = len(i("material_type"))
    d.material_type, d.material_color = _iptr(keep["material_type"]), _fptr(f("material_color"))
    d.material_ro, d.material_ri = _fptr(f("material_ro")), _fptr(f("material_ri"))
    d.n_vertices = f("vertices").size // 3
    d.vertices, d.normals = _fptr(keep["vertices"]), _fptr(f("normals"))
    d.n_triangles = len(i("tri_material"))
    d.tri_v, d.tri_n, d.tri_material = _iptr(i("tri_v")), _iptr(i("tri_n")), _iptr(keep["tri_material"])
    d.n_boxes = f("box_min").size // 3
    d.box_min, d.box_max = _fptr(keep["box_min"]), _fptr(f("box_max"))
    d.n_lights = len(f("light_intensity"))
    d.light_pos, d.light_intensity = _fptr(f("light_pos")), _fptr(keep["light_intensity"])
    return d, list(keep.values())


class DeviceScene:
    """A scene resident on the current HIP device (rt_scene)."""

    def __init__(self, desc: SceneDesc):
        self._h = C.c_void_p()
        _check(lib.rt_scene_upload(C.byref(desc), C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib.rt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam: Camera, opt: Options) -> Tuple[np.ndarray, Stats]:
        out = np.empty((cam.height, cam.width, 3), dtype=np.float32)
        st = Stats()
        _check(lib.rt_render(self._h, C.byref(cam), C.byref(opt), _fptr(out), C.byref(st)))
        return out, st

    def accumulate(self, cam: Camera, opt: Options, sample_begin: int, sample_end: int, d_sum_ptr: int,
                   stream: int = 0, want_stats: bool = False, rows: Optional[Tuple[int, int]] = None) -> Optional[Stats]:
        """Adds samples [sample_begin, sample_end) of the image rows `rows` (default: all) to the device buffer."""
        st = Stats() if want_stats else None
        row_begin, row_end = rows if rows is not None else (0, cam.height)
        _check(lib.rt_accumulate_rows(self._h, C.byref(cam), C.byref(opt), sample_begin, sample_end, row_begin, row_end,
                                      C.c_void_p(d_sum_ptr), C.c_void_p(stream), C.byref(st) if st is not None else None))
        return st

    def profile_collect(self) -> Stats:
        """The kernel timings of the last accumulate() made with make_options(profile=2) and want_stats=False:
        that call only recorded its marks on the stream; this one waits for its work and reads them."""
        st = Stats()
        _check(lib.rt_profile_collect(self._h, C.byref(st)))
        return st

    def accumulate_static(self, cam: Camera, opt: Options, sample_begin: int, sample_end: int, d_sum_ptr: int,
                          stream: int = 0) -> None:
        """rt_accumulate_static: the same sums from the library's fixed work area, without any device allocation."""
        _check(lib.rt_accumulate_static(self._h, C.byref(cam), C.byref(opt), sample_begin, sample_end,
                                        C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def set_progress(self, callback) -> None:
        """callback(paths_done, paths_total) after every finished batch of a frame; None removes it."""
        self._progress = PROGRESS_FN(lambda done, total, _user: callback(done, total)) if callback else PROGRESS_FN()
        _check(lib.rt_set_progress(self._h, self._progress, None))

    def cast(self, opt: Options, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        t = np.empty(n, np.float32)
        sid = np.empty(n, np.int32)
        tid = np.empty(n, np.int32)
        abg = np.zeros((n, 3), np.float32)
        _check(lib.rt_cast(self._h, C.byref(opt), n, _fptr(o), _fptr(d), _fptr(t), _iptr(sid), _iptr(tid), _fptr(abg)))
        return t, sid, tid, abg

    def visible(self, opt: Options, origins, targets):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, 3)
        vis = np.empty(len(o), np.int32)
        _check(lib.rt_visible(self._h, C.byref(opt), len(o), _fptr(o), _fptr(d), _iptr(vis)))
        return vis


class DeviceGroup:
    """A scene resident on several HIP devices of this node, rendered from this one process (rt_device_group)."""

    def __init__(self, desc: SceneDesc, devices):
        self._h = C.c_void_p()
        arr = np.ascontiguousarray(devices, dtype=np.int32)
        _check(lib.rt_device_group_create(C.byref(desc), _iptr(arr), len(arr), C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib.rt_device_group_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam: Camera, opt: Options) -> Tuple[np.ndarray, Stats]:
        out = np.empty((cam.height, cam.width, 3), dtype=np.float32)
        st = Stats()
        _check(lib.rt_device_group_render(self._h, C.byref(cam), C.byref(opt), _fptr(out), C.byref(st)))
        return out, st


def frame_partition(rank: int, world: int, spp: int, height: int) -> Tuple[int, int, int, int]:
    """rt_frame_partition: (sample_begin, sample_end, row_begin, row_end) of `rank`'s share of a frame."""
    out = (C.c_uint32 * 4)()
    lib.rt_frame_partition(rank, world, spp, height, C.byref(out))
    return tuple(int(v) for v in out)


def add_sums(d_dst_ptr: int, d_src_ptr: int, n_floats: int, stream: int = 0) -> None:
    _check(lib.rt_add_sums(C.c_void_p(d_dst_ptr), C.c_void_p(d_src_ptr), n_floats, C.c_void_p(stream)))


def random_cos(normals, r1, r2) -> np.ndarray:
    """rt_random_cos: the device's cosine-weighted direction for every (normal, r1, r2)."""
    nrm = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
    a = np.ascontiguousarray(r1, dtype=np.float32)
    b = np.ascontiguousarray(r2, dtype=np.float32)
    out = np.empty_like(nrm)
    _check(lib.rt_random_cos(len(nrm), _fptr(nrm), _fptr(a), _fptr(b), _fptr(out)))
    return out


def resolve(d_sum_ptr: int, n_pixels: int, d_out_ptr: int, stream: int = 0) -> None:
    _check(lib.rt_resolve(C.c_void_p(d_sum_ptr), n_pixels, C.c_void_p(d_out_ptr), C.c_void_p(stream)))


def init(device: int = 0) -> None:
    _check(lib.rt_init(device))


class Scene:
    """Python face of the C++ ``Scene`` (load_scene / render / add_* / camera)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib.rt_host_scene_create(C.byref(self._h)))
        self.options = make_options()

    def close(self) -> None:
        if self._h:
            lib.rt_host_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_scene(self, filepath: str) -> None:
        _check(lib.rt_host_scene_load(self._h, os.fsencode(filepath)))

    def set_camera_intrinsics(self, fov: float, width: int, height: int) -> None:
        _check(lib.rt_host_scene_set_camera_intrinsics(self._h, fov, width, height))

    def rotate_camera(self, alpha: float, beta: float, gamma: float) -> None:
        _check(lib.rt_host_scene_rotate_camera(self._h, alpha, beta, gamma))

    def transform_camera(self, x: float, y: float, z: float) -> None:
        _check(lib.rt_host_scene_transform_camera(self._h, x, y, z))

    def set_devices(self, n: int) -> None:
        """Scene::settings.devices: render() splits every frame over HIP devices 0 .. n-1."""
        _check(lib.rt_host_scene_set_devices(self._h, n))

    def set_num_rays(self, n: int) -> None:
        _check(lib.rt_host_scene_set_num_rays(self._h, n))

    def add_sphere(self, pos, radius, color, material_type, ro=0.0, ri=0.0) -> None:
        _check(lib.rt_host_scene_add_sphere(self._h, _vec3(pos), radius, _vec3(color), material_type, ro, ri))

    def add_object(self, obj_path, pos, scale, rotation, color, material_type, ro=0.0, ri=0.0) -> None:
        _check(lib.rt_host_scene_add_object(self._h, os.fsencode(obj_path), _vec3(pos), _vec3(scale), _vec3(rotation),
                                            _vec3(color), material_type, ro, ri))

    def add_light(self, pos, intensity) -> None:
        _check(lib.rt_host_scene_add_light(self._h, _vec3(pos), intensity))

    def describe(self) -> Tuple[SceneDesc, Camera]:
        d, c = SceneDesc(), Camera()
        _check(lib.rt_host_scene_describe(self._h, C.byref(d), C.byref(c)))
        d._owner = self  # the pointers refer to this scene's arrays: keep it alive with the view
        return d, c

    def arrays(self) -> dict:
        return desc_arrays(self.describe()[0])

    def render(self, imagepath: Optional[str], seed: int) -> np.ndarray:
        """Scene::render; returns the unclamped float image (H, W, 3)."""
        _check(lib.rt_host_scene_set_options(self._h, C.byref(self.options)))
        _check(lib.rt_host_scene_render(self._h, os.fsencode(imagepath) if imagepath else None, seed))
        w, h = C.c_uint32(), C.c_uint32()
        p = lib.rt_host_scene_image(self._h, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()

    def render_frames(self, pattern: Optional[str], first_seed: int, n_frames: int, rotate_step=(0, 0, 0),
                      move_step=(0, 0, 0), first: int = 0, stride: int = 1, writers: int = 4) -> float:
        """Scene::render_frames: frame i after i camera steps, seed first_seed + i, file pattern % i.

        One process per GPU renders its share with first=rank, stride=world size.  Returns seconds."""
        _check(lib.rt_host_scene_set_options(self._h, C.byref(self.options)))
        seconds = C.c_double()
        _check(lib.rt_host_scene_render_frames(self._h, os.fsencode(pattern) if pattern else None, first_seed, n_frames,
                                               _vec3(rotate_step), _vec3(move_step), first, stride, writers,
                                               C.byref(seconds)))
        return seconds.value

    def upload(self) -> DeviceScene:
        return DeviceScene(self.describe()[0])


def write_png(path: str, rgb8: np.ndarray) -> None:
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = a.shape
    _check(lib.rt_write_png(os.fsencode(path), w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))))


def write_pfm(path: str, image: np.ndarray) -> None:
    a = np.ascontiguousarray(image, dtype=np.float32)
    h, w, _ = a.shape
    _check(lib.rt_write_pfm(os.fsencode(path), w, h, _fptr(a)))


def read_pfm(path: str) -> np.ndarray:
    """Reads a colour PFM written by write_pfm / Scene.render_frames back as (H, W, 3), row 0 on top."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError(f"{path}: not a colour PFM")
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3)[::-1].astype(np.float32)

"""rabbit-transcoding_amd: ctypes binding of librbt.so (include/rbt.h), the MI355X-native V-PCC transcoding hot path.

The package directory name contains a hyphen (mandated layout), so load it with
    importlib.util.spec_from_file_location("rabbit_transcoding_amd", ".../rabbit-transcoding_amd/__init__.py")
or through tests/rbt_lib.py. There is no CPU fallback: creating a Context without a HIP device raises RbtError.
"""
import ctypes as C
import os
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RBT_LIB_PATH") or os.path.join(_DIR, "librbt.so")   # RBT_LIB_PATH: another build of the same library (experiments)

RBT_VIDEO_OCCUPANCY, RBT_VIDEO_GEOMETRY, RBT_VIDEO_ATTRIBUTE = 0, 1, 19


class RbtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rbt error {code}: {msg}")
        self.code = code


class Memory(C.Structure):
    """rbt_memory (include/rbt.h)"""
    _fields_ = [(n, C.c_size_t) for n in ("total_bytes", "free_bytes", "cached_bytes", "in_use_bytes", "reserve_bytes")]


class StreamParams(C.Structure):
    _fields_ = [("video_type", C.c_int), ("qp", C.c_int), ("occupancy_precision", C.c_int), ("log2_ctb", C.c_int),
                ("ctb_rows_per_slice", C.c_int), ("md5_sei", C.c_int), ("verify_md5", C.c_int), ("occupancy_rd", C.c_int), ("preset", C.c_int)]


class Video(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("n_frames", C.c_int),
                ("data", C.POINTER(C.c_uint16)), ("md5_checked", C.c_int), ("md5_failed", C.c_int)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("host_parse_ms", "h2d_ms", "gpu_ms", "d2h_ms", "host_pack_ms", "total_ms", "k_parse_ms",
                                          "k_recon_ms", "k_filter_ms", "k_analyse_ms", "k_encode_ms", "k_entropy_ms")] + [("algorithmic_bytes", C.c_uint64)]


class Patch(C.Structure):
    """rbt_patch: the fields of PCCPatch the reconstruction reads"""
    _fields_ = [(n, C.c_int32) for n in ("u0", "v0", "size_u0", "size_v0", "u1", "v1", "d1", "normal_axis", "tangent_axis", "bitangent_axis", "projection_mode", "orientation", "lod_x", "lod_y")]


class AtlasParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "occupancy_resolution", "occupancy_precision", "map_count", "absolute_d1", "remove_duplicate_points", "threshold_lossy_om",
                                            "geometry_smoothing", "grid_size", "threshold_smoothing")]      # the last three default to 0 = no smoothing; CTC: 1, 8, 64


def ctc_smoothing(atlas):
    """the same atlas with the geometry smoothing of the CTC switched on (cfg/common/ctc-common.cfg:57-60: gridSmoothing, gridSize 8, thresholdSmoothing 64)"""
    a = AtlasParams(*[getattr(atlas, n) for n, _ in AtlasParams._fields_])
    a.geometry_smoothing, a.grid_size, a.threshold_smoothing = 1, 8, 64
    return a


class Cloud(C.Structure):
    _fields_ = [("n_points", C.c_int), ("xyz", C.POINTER(C.c_int16)), ("yuv", C.POINTER(C.c_uint16)), ("occupancy_map", C.POINTER(C.c_uint8)), ("block_to_patch", C.POINTER(C.c_uint32)), ("n_smoothed", C.c_int)]


class D1Result(C.Structure):
    _fields_ = [("n_a", C.c_int), ("n_b", C.c_int), ("sse_ab", C.c_uint64), ("sse_ba", C.c_uint64), ("max_ab", C.c_uint64), ("max_ba", C.c_uint64),
                ("mse_ab", C.c_float), ("mse_ba", C.c_float), ("psnr_ab", C.c_float), ("psnr_ba", C.c_float), ("psnr", C.c_float)]


class D2Result(C.Structure):
    _fields_ = [("n_a", C.c_int), ("n_b", C.c_int), ("sse_ab", C.c_double), ("sse_ba", C.c_double), ("max_ab", C.c_double), ("max_ba", C.c_double),
                ("mse_ab", C.c_float), ("mse_ba", C.c_float), ("psnr_ab", C.c_float), ("psnr_ba", C.c_float), ("psnr", C.c_float)]


class V3CUnit(C.Structure):
    """rbt_v3c_unit: one unit of a V3C sample stream"""
    _fields_ = [(n, C.c_int) for n in ("type", "gof", "parameter_set_id", "atlas_id", "attribute_index", "attribute_dimension_index", "map_index", "auxiliary_video", "video_type")] + \
               [("offset", C.c_size_t), ("size", C.c_size_t)]


class V3CParams(C.Structure):
    """rbt_v3c_params: PCCTranscoderParameters as the container walk needs them"""
    _fields_ = [(n, C.c_int) for n in ("occupancy_precision", "geometry_qp", "attribute_qp", "forced_unit_size_precision_bytes", "log2_ctb", "ctb_rows_per_slice", "md5_sei", "verify_md5", "gofs_per_job", "occupancy_rd", "preset")]


class V3CStat(C.Structure):
    """rbt_v3c_stat: PCCBitstreamStat of a V3C sample stream"""
    _fields_ = [("n_units", C.c_int), ("n_gofs", C.c_int), ("unit_size_precision_bytes", C.c_int), ("header", C.c_uint64), ("unit_size", C.c_uint64 * 5)] + \
               [(n, C.c_uint64) for n in ("occupancy_video", "geometry_video", "geometry_aux_video", "attribute_video", "attribute_aux_video", "total_metadata", "total_geometry", "total_attribute", "total")]


V3C_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))   # rbt_v3c_sink
RBT_V3C_VPS, RBT_V3C_AD, RBT_V3C_OVD, RBT_V3C_GVD, RBT_V3C_AVD = range(5)


def load(path=None):
    """Loads the shared library and declares the C ABI. Raises OSError if the HIP extension has not been built."""
    # 16 HIP streams shared by the jobs in flight: the ROCm runtime multiplexes streams onto 4 hardware queues unless told otherwise, and
    # it reads this when it initialises (librbt also sets it in rbt_create, which is too late if torch touched the GPU first)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    L = C.CDLL(path or LIB_PATH)
    L.rbt_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    L.rbt_destroy.argtypes = [C.c_void_p]
    L.rbt_owns_gof.argtypes = [C.c_void_p, C.c_int]
    L.rbt_world.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rbt_strerror.restype = C.c_char_p
    L.rbt_strerror.argtypes = [C.c_int]
    L.rbt_last_error.restype = C.c_char_p
    L.rbt_last_error.argtypes = [C.c_void_p]
    L.rbt_version.restype = C.c_char_p
    L.rbt_free.argtypes = [C.c_void_p]
    L.rbt_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Video)]
    L.rbt_encode.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_transcode_substream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(StreamParams), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_transcode_gof.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(StreamParams), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_submit_gof.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(StreamParams), C.POINTER(C.c_void_p)]
    L.rbt_set_depth.argtypes = [C.c_void_p, C.c_int]
    L.rbt_trim.argtypes = [C.c_void_p]
    L.rbt_get_depth.argtypes = [C.c_void_p]
    L.rbt_job_shape.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rbt_preset_from_name.argtypes = [C.c_char_p]
    L.rbt_wait_gof.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_or_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.rbt_sample_to_byte_stream.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_byte_to_sample_stream.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.rbt_selftest_transform32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.rbt_reconstruct.argtypes = [C.c_void_p, C.POINTER(AtlasParams), C.POINTER(Patch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Cloud)]
    L.rbt_cloud_free.argtypes = [C.POINTER(Cloud)]
    L.rbt_d1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(D1Result)]
    L.rbt_d2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(D2Result)]
    L.rbt_v3c_index.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(V3CUnit)), C.POINTER(C.c_int)]
    L.rbt_v3c_write.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rbt_v3c_stats.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(V3CStat)]
    L.rbt_device_memory.argtypes = [C.c_void_p, C.POINTER(Memory)]
    L.rbt_job_memory.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    L.rbt_transcode_v3c_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(V3CParams), V3C_SINK, C.c_void_p]
    L.rbt_transcode_v3c.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(V3CParams), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    return L


def _convert(fn, data, L):
    out, n = C.c_void_p(), C.c_size_t()
    rc = fn(data, len(data), C.byref(out), C.byref(n))
    if rc != 0:
        raise RbtError(rc, L.rbt_strerror(rc).decode())
    res = C.string_at(out, n.value)
    L.rbt_free(out)
    return res


RBT_PRESET_DEFAULT, RBT_PRESET_FAST = 0, 1


def preset_from_name(name, lib=None):
    """rbt_preset_from_name: the reference's `preset` string (an x265 preset name) as RBT_PRESET_*"""
    L = lib or load()
    rc = L.rbt_preset_from_name(name.encode() if name is not None else None)
    if rc < 0:
        raise RbtError(rc, L.rbt_strerror(rc).decode())
    return rc


def job_shape(n_gofs, max_jobs=16, lib=None):
    """rbt_job_shape: (GOFs per job, jobs in flight) for a walk of n_gofs GOFs on one GPU"""
    L = lib or load()
    g, d = C.c_int(), C.c_int()
    rc = L.rbt_job_shape(n_gofs, max_jobs, C.byref(g), C.byref(d))
    if rc != 0:
        raise RbtError(rc, L.rbt_strerror(rc).decode())
    return g.value, d.value


def byte_to_sample_stream(data: bytes, lib=None):
    """rbt_byte_to_sample_stream (PCCVideoBitstream::byteStreamToSampleStream): Annex-B -> 4-byte sizes (host only)"""
    L = lib or load()
    return _convert(L.rbt_byte_to_sample_stream, data, L)


def sample_to_byte_stream(data: bytes, lib=None):
    """rbt_sample_to_byte_stream (PCCVideoBitstream::sampleStreamToByteStream): 4-byte sizes -> Annex-B (host only)"""
    L = lib or load()
    return _convert(L.rbt_sample_to_byte_stream, data, L)


def v3c_index(data: bytes, lib=None):
    """rbt_v3c_index: the units of a V3C sample stream as a list of dicts (host only: works without a GPU)"""
    L = lib or load()
    u, n = C.POINTER(V3CUnit)(), C.c_int()
    rc = L.rbt_v3c_index(data, len(data), C.byref(u), C.byref(n))
    if rc != 0:
        raise RbtError(rc, L.rbt_strerror(rc).decode())
    out = [{f: getattr(u[i], f) for f, _ in V3CUnit._fields_} for i in range(n.value)]
    L.rbt_free(u)
    return out


def v3c_stats(data: bytes, lib=None):
    """rbt_v3c_stats: the PCCBitstreamStat figures of a V3C sample stream as a dict (host only)"""
    L = lib or load()
    st = V3CStat()
    rc = L.rbt_v3c_stats(data, len(data), C.byref(st))
    if rc != 0:
        raise RbtError(rc, L.rbt_strerror(rc).decode())
    return {n: (list(getattr(st, n)) if n == "unit_size" else getattr(st, n)) for n, _ in V3CStat._fields_}


def v3c_write(units, forced_precision_bytes=0, lib=None):
    """rbt_v3c_write: a V3C sample stream from a list of units (bytes, each with its 4-byte header; host only)"""
    L = lib or load()
    k = len(units)
    arr = (C.c_char_p * max(1, k))(*units)
    sizes = (C.c_size_t * max(1, k))(*[len(x) for x in units])
    out, n = C.c_void_p(), C.c_size_t()
    rc = L.rbt_v3c_write(arr, sizes, k, forced_precision_bytes, C.byref(out), C.byref(n))
    if rc != 0:
        raise RbtError(rc, L.rbt_strerror(rc).decode())
    res = C.string_at(out, n.value)
    L.rbt_free(out)
    return res


class Context:
    """One transcoder context (rbt_create). Mirrors how PCCTranscoder is used: construct, then transcode per GOF."""

    def __init__(self, device=0, rank=0, world=1, lib_path=None):
        self.L = load(lib_path)
        self.h = C.c_void_p()
        rc = self.L.rbt_create(C.byref(self.h), device, rank, world)
        if rc != 0:
            raise RbtError(rc, self.L.rbt_strerror(rc).decode())

    def owns_gof(self, g):
        """rbt_owns_gof: GOF g of a sequence belongs to rank g mod world"""
        return bool(self.L.rbt_owns_gof(self.h, g))

    def close(self):
        if self.h:
            self.L.rbt_destroy(self.h)
            self.h = C.c_void_p()

    def _chk(self, rc):
        if rc != 0:
            detail = self.L.rbt_last_error(self.h).decode()
            raise RbtError(rc, self.L.rbt_strerror(rc).decode() + (": " + detail if detail else ""))

    def _take(self, ptr, n):
        out = C.string_at(ptr, n.value) if ptr.value else b""
        self.L.rbt_free(ptr)
        return out

    def decode(self, stream: bytes, verify_md5=True):
        v = Video()
        self._chk(self.L.rbt_decode(self.h, stream, len(stream), int(verify_md5), C.byref(v)))
        fs = v.width * v.height * 3 // 2
        arr = np.ctypeslib.as_array(v.data, shape=(v.n_frames, fs)).copy()
        self.L.rbt_free(v.data)
        return arr, v.width, v.height, v.bit_depth, v.md5_checked, v.md5_failed

    def encode(self, frames, w, h, bit_depth, qp, gop=2, lossless=0, log2_ctb=5, rows_per_slice=1, md5_sei=1):
        frames = np.ascontiguousarray(frames, dtype=np.uint16)
        out, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.rbt_encode(self.h, frames.ctypes.data, w, h, bit_depth, frames.shape[0], qp, gop, lossless, log2_ctb, rows_per_slice, md5_sei, C.byref(out), C.byref(n)))
        return self._take(out, n)

    def transcode_substream(self, stream: bytes, video_type, qp, occupancy_precision=4, log2_ctb=5, rows_per_slice=1, md5_sei=1, verify_md5=0, preset=0):
        p = StreamParams(video_type, qp, occupancy_precision, log2_ctb, rows_per_slice, md5_sei, verify_md5, 0, preset)
        out, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.rbt_transcode_substream(self.h, stream, len(stream), C.byref(p), C.byref(out), C.byref(n)))
        return self._take(out, n)

    def transcode_gof(self, streams, params):
        k = len(streams)
        ins = (C.c_char_p * k)(*streams)
        sizes = (C.c_size_t * k)(*[len(s) for s in streams])
        ps = (StreamParams * k)(*params)
        outs = (C.c_void_p * k)()
        ns = (C.c_size_t * k)()
        self._chk(self.L.rbt_transcode_gof(self.h, k, ins, sizes, ps, outs, ns))
        res = []
        for i in range(k):
            res.append(C.string_at(outs[i], ns[i]) if outs[i] else b"")
            self.L.rbt_free(outs[i])
        return res

    def transcode_v3c(self, data: bytes, geometry_qp, attribute_qp, occupancy_precision=4, forced_precision_bytes=0, log2_ctb=5, rows_per_slice=-1, md5_sei=0,
                      verify_md5=0, gofs_per_job=1, occupancy_rd=0, preset=0):
        """rbt_transcode_v3c: a whole V3C sample stream (every GOF this context owns) -> transcoded sample stream; gofs_per_job=0: job shape by rbt_job_shape"""
        p = V3CParams(occupancy_precision, geometry_qp, attribute_qp, forced_precision_bytes, log2_ctb, rows_per_slice, md5_sei, verify_md5, gofs_per_job, occupancy_rd, preset)
        out, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.rbt_transcode_v3c(self.h, data, len(data), C.byref(p), C.byref(out), C.byref(n)))
        return self._take(out, n)

    def transcode_v3c_stream(self, data: bytes, sink, geometry_qp, attribute_qp, occupancy_precision=4, log2_ctb=5, rows_per_slice=-1, md5_sei=0, verify_md5=0, gofs_per_job=0):
        """rbt_transcode_v3c_stream: sink(gof, [unit bytes, ...]) is called once per GOF this context owns, in GOF order, while later GOFs are still on the GPU; a truthy
        return of the sink ends the walk"""
        p = V3CParams(occupancy_precision, geometry_qp, attribute_qp, 0, log2_ctb, rows_per_slice, md5_sei, verify_md5, gofs_per_job)

        def cb(_user, gof, n_units, unit, unit_size):
            return 1 if sink(gof, [C.string_at(unit[i], unit_size[i]) for i in range(n_units)]) else 0
        self._chk(self.L.rbt_transcode_v3c_stream(self.h, data, len(data), C.byref(p), V3C_SINK(cb), None))

    def device_memory(self):
        """rbt_device_memory: {total, free, cached, in_use, reserve} bytes as the library sees the device"""
        m = Memory()
        self._chk(self.L.rbt_device_memory(self.h, C.byref(m)))
        return {"total": m.total_bytes, "free": m.free_bytes, "cached": m.cached_bytes, "in_use": m.in_use_bytes, "reserve": m.reserve_bytes}

    def job_memory(self, job):
        """rbt_job_memory: device bytes a submitted job holds"""
        b = C.c_size_t()
        self._chk(self.L.rbt_job_memory(self.h, job[0] if isinstance(job, tuple) else job, C.byref(b)))
        return b.value

    def set_depth(self, n):
        """rbt_set_depth: how many GOFs the caller will keep in flight (1..16 = RBT_MAX_JOBS, default 4)"""
        self._chk(self.L.rbt_set_depth(self.h, n))

    def get_depth(self):
        """rbt_get_depth: the depth announced with set_depth"""
        return self.L.rbt_get_depth(self.h)

    def trim(self):
        """rbt_trim: hand the cached device memory of earlier jobs back to the driver (call when the workload changes shape)"""
        self._chk(self.L.rbt_trim(self.h))

    def submit_gof(self, streams, params):
        """rbt_submit_gof: enqueue one GOF; returns a job for wait_gof. Up to set_depth() jobs may be in flight."""
        k = len(streams)
        ins = (C.c_char_p * k)(*streams)
        sizes = (C.c_size_t * k)(*[len(s) for s in streams])
        ps = (StreamParams * k)(*params)
        job = C.c_void_p()
        self._chk(self.L.rbt_submit_gof(self.h, k, ins, sizes, ps, C.byref(job)))
        return (job, k)

    def wait_gof(self, job):
        h, k = job
        outs = (C.c_void_p * k)()
        ns = (C.c_size_t * k)()
        self._chk(self.L.rbt_wait_gof(self.h, h, outs, ns))
        res = []
        for i in range(k):
            res.append(C.string_at(outs[i], ns[i]) if outs[i] else b"")
            self.L.rbt_free(outs[i])
        return res

    def or_pool(self, plane, factor=2):
        plane = np.ascontiguousarray(plane, dtype=np.uint16)
        h, w = plane.shape
        out = np.zeros((h // factor, w // factor), np.uint16)
        self._chk(self.L.rbt_or_pool(self.h, plane.ctypes.data, w, h, factor, out.ctypes.data))
        return out

    def reconstruct(self, atlas, patches, occ, d0, d1, geo_bd=10, t0=None, t1=None, attr_bd=10):
        """rbt_reconstruct: (xyz int16 [n,3], yuv uint16 [n,3], occupancy_map uint8 [h,w], block_to_patch uint32 [h/res, w/res])
        atlas: AtlasParams; patches: list of Patch; occ / d0 / d1: luma planes (2-D arrays); t0 / t1: planar 4:2:0 frames (1-D) or None"""
        ps = (Patch * max(1, len(patches)))(*patches)
        arr = [np.ascontiguousarray(x, dtype=np.uint16) if x is not None else None for x in (occ, d0, d1, t0, t1)]
        ptr = [x.ctypes.data if x is not None else None for x in arr]
        c = Cloud()
        self._chk(self.L.rbt_reconstruct(self.h, C.byref(atlas), ps, len(patches), ptr[0], ptr[1], ptr[2], geo_bd, ptr[3], ptr[4], attr_bd, C.byref(c)))
        n, w, h, res = c.n_points, atlas.width, atlas.height, atlas.occupancy_resolution
        xyz = np.ctypeslib.as_array(c.xyz, shape=(max(n, 1), 3))[:n].copy(); yuv = np.ctypeslib.as_array(c.yuv, shape=(max(n, 1), 3))[:n].copy()
        om = np.ctypeslib.as_array(c.occupancy_map, shape=(h, w)).copy(); b2p = np.ctypeslib.as_array(c.block_to_patch, shape=(h // res, w // res)).copy()
        self.n_smoothed = c.n_points and c.n_smoothed      # points the geometry smoothing moved in this call
        self.L.rbt_cloud_free(C.byref(c))
        return xyz, yuv, om, b2p

    def d1(self, a, b, peak=1023):
        """rbt_d1: point-to-point metric between two clouds (int16 [n,3]) -> dict"""
        a = np.ascontiguousarray(a, dtype=np.int16); b = np.ascontiguousarray(b, dtype=np.int16)
        r = D1Result()
        self._chk(self.L.rbt_d1(self.h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], peak, C.byref(r)))
        return {n: getattr(r, n) for n, _ in D1Result._fields_}

    def d2(self, a, normals_a, b, peak=1023):
        """rbt_d2: point-to-plane metric; a, b int16 [n,3]; normals_a int16 [n_a,3] in Q14 (16384 = 1.0) -> dict"""
        a = np.ascontiguousarray(a, dtype=np.int16); b = np.ascontiguousarray(b, dtype=np.int16); na = np.ascontiguousarray(normals_a, dtype=np.int16)
        if na.shape != a.shape:
            raise ValueError("one normal per point of a")
        r = D2Result()
        self._chk(self.L.rbt_d2(self.h, a.ctypes.data, na.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], peak, C.byref(r)))
        return {n: getattr(r, n) for n, _ in D2Result._fields_}

    def selftest_transform32(self, blocks, bit_depth=10):
        """rbt_selftest_transform32: matrix-core vs vector-ALU 32-point transforms on int16 blocks [n, 1024]; returns the number of differing samples"""
        blocks = np.ascontiguousarray(blocks, dtype=np.int16)
        bad = C.c_uint32()
        self._chk(self.L.rbt_selftest_transform32(self.h, blocks.ctypes.data, blocks.shape[0], bit_depth, C.byref(bad)))
        return bad.value

    def stats(self):
        s = Stats()
        self._chk(self.L.rbt_get_stats(self.h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (the CPU checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class OracleVideo(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("bit_depth", C.c_int), ("n_frames", C.c_int),
                ("data", C.POINTER(C.c_uint16)), ("md5_checked", C.c_int), ("md5_failed", C.c_int)]


class TranscodeParams(C.Structure):
    _fields_ = [("video_type", C.c_int), ("qp", C.c_int), ("occupancy_precision", C.c_int), ("log2_ctb", C.c_int),
                ("ctb_rows_per_slice", C.c_int), ("md5_sei", C.c_int), ("occupancy_rd", C.c_int), ("preset", C.c_int)]


class EncParams(C.Structure):
    """oracle_enc_params (oracle/hevc_enc.h)"""
    _fields_ = [(n, C.c_int) for n in ("width", "height", "bit_depth", "qp", "i_qp_offset", "gop", "lossless", "log2_ctb", "ctb_rows_per_slice", "md5_sei")] + \
               [("stress_seed", C.c_uint32)] + [(n, C.c_int) for n in ("conf_win_right", "conf_win_bottom", "hm_like", "p_qp_offset")] + \
               [("hint_modes", C.c_void_p), ("hint_w4", C.c_int), ("hint_h4", C.c_int), ("occ4", C.c_void_p), ("occ4_w", C.c_int), ("occ4_h", C.c_int), ("ctc_gop", C.c_int), ("log2_max_poc_lsb", C.c_int), ("first_idx", C.c_int), ("weighted_pred", C.c_int), ("tools_off", C.c_int)]


class OPatch(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("u0", "v0", "size_u0", "size_v0", "u1", "v1", "d1", "normal_axis", "tangent_axis", "bitangent_axis", "projection_mode", "orientation", "lod_x", "lod_y")]


class OAtlas(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "occupancy_resolution", "occupancy_precision", "map_count", "absolute_d1", "remove_duplicate_points", "threshold_lossy_om",
                                            "geometry_smoothing", "grid_size", "threshold_smoothing")]


class OCloud(C.Structure):
    _fields_ = [("n", C.c_int), ("xyz", C.POINTER(C.c_int16)), ("yuv", C.POINTER(C.c_uint16)), ("occupancy_map", C.POINTER(C.c_uint8)), ("block_to_patch", C.POINTER(C.c_uint32)), ("n_smoothed", C.c_int)]


LAST_SMOOTHED = 0      # points the geometry smoothing moved in the last reconstruct() call


class OD1(C.Structure):
    _fields_ = [("n_a", C.c_int), ("n_b", C.c_int), ("sse_ab", C.c_uint64), ("sse_ba", C.c_uint64), ("max_ab", C.c_uint64), ("max_ba", C.c_uint64),
                ("mse_ab", C.c_float), ("mse_ba", C.c_float), ("psnr_ab", C.c_float), ("psnr_ba", C.c_float), ("psnr", C.c_float)]


class OD2(C.Structure):
    _fields_ = [("n_a", C.c_int), ("n_b", C.c_int), ("sse_ab", C.c_double), ("sse_ba", C.c_double), ("max_ab", C.c_double), ("max_ba", C.c_double),
                ("mse_ab", C.c_float), ("mse_ba", C.c_float), ("psnr_ab", C.c_float), ("psnr_ba", C.c_float), ("psnr", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(OracleVideo)]
        L.oracle_encode.argtypes = [C.c_int] * 10 + [C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        L.oracle_encode_ex.argtypes = [C.POINTER(EncParams), C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        L.oracle_transcode_substream.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(TranscodeParams), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.oracle_transcode_data.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(TranscodeParams), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.oracle_reconstruct.argtypes = [C.POINTER(OAtlas), C.POINTER(OPatch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OCloud)]
        L.oracle_cloud_free.argtypes = [C.POINTER(OCloud)]
        L.oracle_d1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(OD1)]
        L.oracle_d2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(OD2)]
        L.oracle_sample_to_byte_stream.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.oracle_byte_to_sample_stream.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.oracle_or_pool.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_md5.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_v3c_transcode.argtypes = [C.c_char_p, C.c_size_t] + [C.c_int] * 9 + [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _LIB = L
    return _LIB


def _take(ptr, n):
    out = C.string_at(ptr, n.value) if ptr.value else b""
    lib().oracle_free(ptr)
    return out


def decode(stream: bytes):
    """-> (frames uint16 [n, w*h*3/2], w, h, bit_depth, md5_checked, md5_failed)"""
    v = OracleVideo()
    rc = lib().oracle_decode(stream, len(stream), C.byref(v))
    if rc != 0:
        raise RuntimeError(f"oracle decode failed rc={rc}")
    fs = v.w * v.h * 3 // 2
    arr = np.ctypeslib.as_array(v.data, shape=(v.n_frames, fs)).copy()
    lib().oracle_free(v.data)
    return arr, v.w, v.h, v.bit_depth, v.md5_checked, v.md5_failed


def encode(frames: np.ndarray, w, h, bit_depth, qp, gop=2, i_qp_offset=-3, lossless=0, log2_ctb=5, rows_per_slice=1,
           md5_sei=1, stress_seed=0, want_recon=True):
    frames = np.ascontiguousarray(frames, dtype=np.uint16)
    n = frames.shape[0]
    assert frames.shape[1] == w * h * 3 // 2
    out = C.c_void_p()
    n_out = C.c_size_t()
    recon = np.zeros_like(frames) if want_recon else None
    rc = lib().oracle_encode(w, h, bit_depth, qp, i_qp_offset, gop, lossless, log2_ctb, rows_per_slice, md5_sei, stress_seed,
                             frames.ctypes.data, n, C.byref(out), C.byref(n_out), recon.ctypes.data if want_recon else None)
    if rc != 0:
        raise RuntimeError(f"oracle encode failed rc={rc}")
    return _take(out, n_out), recon


def encode_ex(frames: np.ndarray, want_recon=True, **kw):
    """oracle_encode_ex with oracle_enc_params fields by name (width, height, bit_depth, qp, ..., ctc_gop, log2_max_poc_lsb, first_idx)"""
    frames = np.ascontiguousarray(frames, dtype=np.uint16)
    p = EncParams()
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    assert frames.shape[1] == p.width * p.height * 3 // 2
    out, n_out = C.c_void_p(), C.c_size_t()
    recon = np.zeros_like(frames) if want_recon else None
    rc = lib().oracle_encode_ex(C.byref(p), frames.ctypes.data, frames.shape[0], C.byref(out), C.byref(n_out), recon.ctypes.data if want_recon else None)
    if rc != 0:
        raise RuntimeError(f"oracle encode failed rc={rc}")
    return _take(out, n_out), recon


def encode_ctc(frames, w, h, bit_depth, qp, ctc_gop=1, log2_max_poc_lsb=0, first_idx=0, hm=1, gop=2, i_qp_offset=-3, p_qp_offset=0, lossless=0, log2_ctb=0,
               md5_sei=1, stress_seed=0, rows_per_slice=0, want_recon=True):
    """Streams in the structure of the CTC's HM encoder (oracle_enc_params.ctc_gop: one IDR, then trailing pictures with reference picture sets, POC running on):
    hm=1 the HM-like toolset, stress_seed != 0 random syntax, otherwise RBT-E1."""
    return encode_ex(frames, want_recon, width=w, height=h, bit_depth=bit_depth, qp=qp, i_qp_offset=i_qp_offset, gop=gop, lossless=lossless,
                     log2_ctb=log2_ctb or (0 if stress_seed else 6 if hm else 5), ctb_rows_per_slice=rows_per_slice, md5_sei=md5_sei, stress_seed=stress_seed,
                     hm_like=1 if (hm and not stress_seed) else 0, p_qp_offset=p_qp_offset, ctc_gop=ctc_gop, log2_max_poc_lsb=log2_max_poc_lsb, first_idx=first_idx)


def encode_hm(frames: np.ndarray, w, h, bit_depth, qp, gop=2, i_qp_offset=-3, p_qp_offset=0, lossless=0, log2_ctb=6, md5_sei=1, want_recon=True):
    """The oracle's HM-like encoder (oracle/hevc_enc.c, hm_like): the coding tools of the CTC input streams, one slice per picture."""
    frames = np.ascontiguousarray(frames, dtype=np.uint16)
    n = frames.shape[0]
    assert frames.shape[1] == w * h * 3 // 2
    p = EncParams(w, h, bit_depth, qp, i_qp_offset, gop, lossless, log2_ctb, 0, md5_sei, 0, 0, 0, 1, p_qp_offset)
    out, n_out = C.c_void_p(), C.c_size_t()
    recon = np.zeros_like(frames) if want_recon else None
    rc = lib().oracle_encode_ex(C.byref(p), frames.ctypes.data, n, C.byref(out), C.byref(n_out), recon.ctypes.data if want_recon else None)
    if rc != 0:
        raise RuntimeError(f"oracle encode failed rc={rc}")
    return _take(out, n_out), recon


def transcode_substream(stream: bytes, video_type, qp, occupancy_precision=4, log2_ctb=5, rows_per_slice=1, md5_sei=1, preset=0):
    p = TranscodeParams(video_type, qp, occupancy_precision, log2_ctb, rows_per_slice, md5_sei, 0, preset)
    out = C.c_void_p()
    n_out = C.c_size_t()
    rc = lib().oracle_transcode_substream(stream, len(stream), C.byref(p), C.byref(out), C.byref(n_out))
    if rc != 0:
        raise RuntimeError(f"oracle transcode failed rc={rc}")
    return _take(out, n_out)


def transcode_data(streams, params):
    """PCCTranscoder::transcodeData on Annex-B sub-bitstreams; params: list of (video_type, qp, occupancy_precision, log2_ctb, rows_per_slice, md5_sei)"""
    k = len(streams)
    ins = (C.c_char_p * k)(*streams)
    sizes = (C.c_size_t * k)(*[len(s) for s in streams])
    ps = (TranscodeParams * k)(*[TranscodeParams(*p) for p in params])
    outs = (C.c_void_p * k)()
    ns = (C.c_size_t * k)()
    rc = lib().oracle_transcode_data(k, ins, sizes, ps, outs, ns)
    if rc != 0:
        raise RuntimeError(f"oracle transcode_data failed rc={rc}")
    res = []
    for i in range(k):
        res.append(C.string_at(outs[i], ns[i]) if outs[i] else b"")
        lib().oracle_free(outs[i])
    return res


def v3c_transcode(data: bytes, geometry_qp, attribute_qp, occupancy_precision=4, forced_precision_bytes=0, log2_ctb=5, rows_per_slice=-1, md5_sei=0, occupancy_rd=0, preset=0):
    """oracle_v3c_transcode: the V3C sample stream walk of PccAppTranscoder around transcodeData"""
    out, n = C.c_void_p(), C.c_size_t()
    rc = lib().oracle_v3c_transcode(data, len(data), occupancy_precision, geometry_qp, attribute_qp, forced_precision_bytes, log2_ctb, rows_per_slice, md5_sei, occupancy_rd, preset, C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"oracle v3c transcode failed rc={rc}")
    return _take(out, n)


def sample_to_byte_stream(b: bytes):
    out = C.c_void_p(); n = C.c_size_t()
    if lib().oracle_sample_to_byte_stream(b, len(b), C.byref(out), C.byref(n)) != 0:
        raise RuntimeError("bad sample stream")
    return _take(out, n)


def byte_to_sample_stream(b: bytes):
    out = C.c_void_p(); n = C.c_size_t()
    if lib().oracle_byte_to_sample_stream(b, len(b), C.byref(out), C.byref(n)) != 0:
        raise RuntimeError("bad byte stream")
    return _take(out, n)


def or_pool(plane: np.ndarray, factor=2):
    plane = np.ascontiguousarray(plane, dtype=np.uint16)
    h, w = plane.shape
    out = np.zeros((h // factor, w // factor), dtype=np.uint16)
    lib().oracle_or_pool(plane.ctypes.data, w, h, factor, out.ctypes.data)
    return out


def md5(b: bytes):
    out = C.create_string_buffer(16)
    lib().oracle_md5(b, len(b), out)
    return out.raw


def reconstruct(atlas, patches, occ, d0, d1, geo_bd=10, t0=None, t1=None, attr_bd=10):
    """oracle_reconstruct; atlas / patches: any ctypes structures with the rbt_atlas_params / rbt_patch field layout"""
    a = OAtlas(*[getattr(atlas, n) for n, _ in OAtlas._fields_])
    ps = (OPatch * max(1, len(patches)))(*[OPatch(*[getattr(p, n) for n, _ in OPatch._fields_]) for p in patches])
    arr = [np.ascontiguousarray(x, dtype=np.uint16) if x is not None else None for x in (occ, d0, d1, t0, t1)]
    ptr = [x.ctypes.data if x is not None else None for x in arr]
    c = OCloud()
    rc = lib().oracle_reconstruct(C.byref(a), ps, len(patches), ptr[0], ptr[1], ptr[2], geo_bd, ptr[3], ptr[4], attr_bd, C.byref(c))
    if rc != 0:
        raise RuntimeError(f"oracle reconstruct failed rc={rc}")
    n, w, h, res = c.n, a.width, a.height, a.occupancy_resolution
    xyz = np.ctypeslib.as_array(c.xyz, shape=(max(n, 1), 3))[:n].copy(); yuv = np.ctypeslib.as_array(c.yuv, shape=(max(n, 1), 3))[:n].copy()
    om = np.ctypeslib.as_array(c.occupancy_map, shape=(h, w)).copy(); b2p = np.ctypeslib.as_array(c.block_to_patch, shape=(h // res, w // res)).copy()
    global LAST_SMOOTHED
    LAST_SMOOTHED = c.n_smoothed
    lib().oracle_cloud_free(C.byref(c))
    return xyz, yuv, om, b2p


def d1(a, b, peak=1023):
    a = np.ascontiguousarray(a, dtype=np.int16); b = np.ascontiguousarray(b, dtype=np.int16)
    r = OD1()
    if lib().oracle_d1(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], peak, C.byref(r)) != 0:
        raise RuntimeError("oracle d1 failed")
    return {n: getattr(r, n) for n, _ in OD1._fields_}


def d2(a, normals_a, b, peak=1023):
    """oracle_d2: point-to-plane metric; normals_a int16 Q14 (16384 = 1.0), three per point of a"""
    a = np.ascontiguousarray(a, dtype=np.int16); b = np.ascontiguousarray(b, dtype=np.int16); na = np.ascontiguousarray(normals_a, dtype=np.int16)
    assert na.shape == a.shape
    r = OD2()
    if lib().oracle_d2(a.ctypes.data, na.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], peak, C.byref(r)) != 0:
        raise RuntimeError("oracle d2 failed")
    return {n: getattr(r, n) for n, _ in OD2._fields_}


def sps_fields(stream: bytes):
    out = (C.c_int * 5)()
    if lib().oracle_sps_fields(stream, len(stream), out) != 0:
        raise RuntimeError("no SPS")
    return dict(zip(("log2_max_poc_lsb", "log2_ctb", "sao", "tmvp", "num_st_rps"), list(out)))


SLICE_FIELDS = ("nal_type", "address", "slice_type", "poc", "tmvp", "sao_luma", "sao_chroma", "num_ref_idx", "cabac_init", "col_ref_idx", "max_merge_cand", "qp",
                "cb_qp_offset", "cr_qp_offset", "deblocking_disabled", "beta_offset_div2", "tc_offset_div2", "lf_across", "dependent")


def slice_headers(stream: bytes, fn=None):
    """every slice segment header as the oracle's parser reads it (fn: another library's accessor with the same signature, e.g. the product's host parser)"""
    cap = 4096
    out = (C.c_int * (64 * cap))()
    f = fn or lib().oracle_slice_headers
    n = f(stream, C.c_size_t(len(stream)), out, cap)
    if n < 0:
        raise RuntimeError("slice header parse failed")
    res = []
    for k in range(n):
        r = out[64 * k:64 * k + 64]
        d = dict(zip(SLICE_FIELDS, r[:19]))
        d["rps"] = [[r[20 + 2 * q], r[21 + 2 * q]] for q in range(min(r[19], 4))]   # the slice's short-term reference picture set: [delta POC, used by the current picture]
        # pred_weight_table of a P slice under weighted_pred_flag: [luma denominator, chroma denominator, per RefPicList0 entry [luma flag, chroma flag, wY, oY, wCb, oCb, wCr, oCr]]
        d["wp"] = [r[29], r[30]] + [list(r[32 + 8 * q:40 + 8 * q]) for q in range(min(d["num_ref_idx"], 4))] if r[28] else []
        res.append(d)
    return res

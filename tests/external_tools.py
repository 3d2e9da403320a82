"""Third-party HEVC tools, if the box has any (none of the build container's or the GPU boxes' images did: row N1 of the round-3 review). Used by
tests/test_external_decoder.py and bench.py to pin this project's reading of H.265 against an independent implementation the moment one is available."""
import os
import shutil
import subprocess
import tempfile
import numpy as np


def find_decoder():
    """-> (name, argv builder) of the first decoder found on PATH, or None. The builder maps (input .265 path, output .yuv path, bit depth) to a command line."""
    if shutil.which("ffmpeg"):
        return "ffmpeg", lambda i, o, bd: ["ffmpeg", "-loglevel", "error", "-y", "-i", i, "-f", "rawvideo", "-pix_fmt", "yuv420p10le" if bd > 8 else "yuv420p", o]
    for exe in ("TAppDecoderStatic", "TAppDecoder"):                    # HM
        if shutil.which(exe):
            return exe, lambda i, o, bd, exe=exe: [exe, "-b", i, "-o", o, "-d", str(bd)]
    if shutil.which("dec265"):                                          # libde265
        return "dec265", lambda i, o, bd: ["dec265", "-q", "-o", o, i]
    return None


def have_libx265():
    if not shutil.which("ffmpeg"):
        return False
    try:
        return "libx265" in subprocess.run(["ffmpeg", "-hide_banner", "-encoders"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return False


def describe():
    d = find_decoder()
    return {"external_decoder": d[0] if d else "absent", "libx265": "present" if have_libx265() else "absent"}


def decode(stream: bytes, w: int, h: int, bit_depth: int):
    """The stream through the external decoder -> frames as uint16 [n, w * h * 3 / 2] (display size, i.e. after the conformance window)."""
    name, argv = find_decoder()
    with tempfile.TemporaryDirectory() as td:
        i, o = os.path.join(td, "in.265"), os.path.join(td, "out.yuv")
        open(i, "wb").write(stream)
        subprocess.run(argv(i, o, bit_depth), check=True, capture_output=True, timeout=600)
        raw = np.fromfile(o, dtype=np.uint16 if bit_depth > 8 else np.uint8)
    fs = w * h * 3 // 2
    assert raw.size and raw.size % fs == 0, (name, raw.size, fs)
    return raw.reshape(-1, fs).astype(np.uint16)


def x265_transcode(stream: bytes, w: int, h: int, bit_depth: int, qp: int, preset: str = "veryfast", lossless: bool = False):
    """What PCCTranscoder::transcodeVideo does with libavcodec + libx265 (setEncoderOptions, PCCTranscoder.cpp:825-904): decode, re-encode with preset / tune=ssim,
    x265-params qp=<qp>:keyint=2:bframes=0 (geometry / attribute) or lossless=1:keyint=1 (occupancy) -> Annex-B bytes."""
    with tempfile.TemporaryDirectory() as td:
        i, o = os.path.join(td, "in.265"), os.path.join(td, "out.265")
        open(i, "wb").write(stream)
        xp = "lossless=1:keyint=1:bframes=0" if lossless else f"qp={qp}:keyint=2:bframes=0"
        subprocess.run(["ffmpeg", "-loglevel", "error", "-y", "-i", i, "-c:v", "libx265", "-preset", preset, "-tune", "ssim", "-x265-params", xp, "-pix_fmt", "yuv420p10le" if bit_depth > 8 else "yuv420p",
                        "-f", "hevc", o], check=True, capture_output=True, timeout=3600)
        return open(o, "rb").read()

"""Loads the hyphen-named package rabbit-transcoding_amd/ as module `rabbit_transcoding_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.join(_ROOT, "rabbit-transcoding_amd", "__init__.py")
HOSTEMU_LIB = os.path.join(_ROOT, "tests", "hostemu", "librbt_hostemu.so")


def module():
    if "rabbit_transcoding_amd" not in sys.modules:
        spec = importlib.util.spec_from_file_location("rabbit_transcoding_amd", _PKG)
        m = importlib.util.module_from_spec(spec)
        sys.modules["rabbit_transcoding_amd"] = m
        spec.loader.exec_module(m)
    return sys.modules["rabbit_transcoding_amd"]


def module_file(name):
    """Loads rabbit-transcoding_amd/<name>.py as rabbit_transcoding_amd_<name>."""
    key = "rabbit_transcoding_amd_" + name
    if key not in sys.modules:
        spec = importlib.util.spec_from_file_location(key, os.path.join(_ROOT, "rabbit-transcoding_amd", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[key] = m
        spec.loader.exec_module(m)
    return sys.modules[key]

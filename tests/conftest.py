import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

try:  # PyTorch ships its own HIP runtime: load it before libvr_hip.so pulls in the system one, whatever tests are selected
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_fills_before_library_kernels():
    """The library's streams are non-blocking: they do not wait for torch's default stream.  A test that fills a tensor
    with torch (torch.zeros, torch.full, a slice copy) and hands its pointer to the library in the next statement races
    with that fill.  Ordered here once, for every entry point that takes device pointers: torch's work is waited for
    before the library's call is made."""
    if torch is None or not torch.cuda.is_available():
        yield
        return
    from volumerenderer_amd import capi
    L = capi.lib()
    for name in ("vr_render_device", "vr_render_device_noalloc", "vr_render_device_bands", "vr_refine_device", "vr_trilerp_device"):
        fn = getattr(L, name)

        def ordered(*args, _fn=fn):
            torch.cuda.synchronize()
            return _fn(*args)
        setattr(L, name, ordered)
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def c60_vox(oracle):
    return oracle.load_ucd(os.path.join(DATA, "C60.vol"))


@pytest.fixture(scope="session")
def skull_vox():
    raw = gzip.open(os.path.join(DATA, "Skull.vol.gz"), "rb").read()
    return np.frombuffer(raw, dtype=np.uint8, count=68 * 256 * 256, offset=28).reshape(68, 256, 256).copy()


@pytest.fixture(scope="session")
def c60_tree(oracle, c60_vox):
    nodes, rd, st = oracle.build_kdt(c60_vox)
    return oracle.trim_nodes(nodes), rd, st


@pytest.fixture(scope="session")
def skull_tree(oracle, skull_vox):
    nodes, rd, st = oracle.build_kdt(skull_vox)
    return oracle.trim_nodes(nodes), rd, st


@pytest.fixture(scope="session")
def golden_gray():
    return np.load(os.path.join(GOLDEN, "c60_renders_512.npz"))["gray"]


@pytest.fixture(scope="session")
def emu_lib():
    """Host build of the lane code (tests/emu): the same header the GPU kernels compile."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "emu", "vr_emu.cpp")
    out = os.path.join(ROOT, "tests", "emu", "libvr_emu.so")
    hdrs = [os.path.join(ROOT, "volumerenderer_amd", "csrc", h) for h in ("vr_trace_core.h", "vr_trace_wide.h", "vr_trace_union.h", "vr_trace_union_wide.h", "vr_trilerp.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(q) for q in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared",
                               "-o", out, src])
    return C.CDLL(out)

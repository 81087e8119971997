"""The C-ABI library loads on a machine without a GPU and exports every symbol include/vr.h declares."""
import ctypes as C
import os
import re

from volumerenderer_amd import capi
from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "vr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = capi.lib()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), "libvr_hip.so does not export %s" % n
    assert sorted(capi.EXPORTS) == names


def test_error_strings_and_kernel_names():
    assert capi.strerror(0) == "ok"
    for code in (-1, -2, -3, -4, -5, -6, -7, -1001):
        assert capi.strerror(code)
    L = capi.lib()
    assert L.vr_kernel_name(3) == b"resume"
    assert L.vr_kernel_name(99) is None


def test_argument_checks_need_no_gpu():
    L = capi.lib()
    assert L.vr_ctx_create(0, None) == capi.VR_EARG
    assert L.vr_upload_kdt(None, None, 0, None) == capi.VR_EARG
    assert L.vr_render(None, None, 0, None, None, None) == capi.VR_EARG
    assert L.vr_sync(None) == capi.VR_EARG
    L.vr_ctx_destroy(None)
    L.vr_free(None)


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        return
    ctx = C.c_void_p()
    rc = capi.lib().vr_ctx_create(0, C.byref(ctx))
    assert rc == capi.VR_ENODEVICE and not ctx

"""The C-ABI library loads and exports every symbol include/ubdpt.h declares; the ctypes mirrors have
the sizes the header promises. No compute call is made (runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

from util import ROOT, capi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ubdpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ubdpt_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(capi.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_struct_sizes_match_reference_layouts():
    assert C.sizeof(capi.RenderParams) == 56          # URenderParameters, uengine/uutils.h:11-20
    assert C.sizeof(capi.MeshVertex) == 44 == capi.MESH_VERTEX_DTYPE.itemsize
    assert capi.RAY_DTYPE.itemsize == 48
    assert capi.HIT_DTYPE.itemsize == 4 + 4 + 8 + 8 + 15 * 8 + 16
    assert capi.SEGMENT_DTYPE.itemsize == 48
    assert C.sizeof(capi.Object) == 8 + 128 + 16 + 8 + 24 + 16
    assert C.sizeof(capi.Camera) == 64
    assert capi.BSDF_QUERY_DTYPE.itemsize == 8 + 8 + 24 + 96 + 24 + 24 + 16
    assert capi.BSDF_RESULT_DTYPE.itemsize == 24 + 8 + 24 + 8


def test_version_and_status_strings():
    lib = capi.load_library()
    assert lib.ubdpt_version() >= 100
    assert lib.ubdpt_status_string(capi.SUCCESS) == b"success"
    assert lib.ubdpt_status_string(capi.STOPPED) == b"stopped"
    assert b"unknown" in lib.ubdpt_status_string(12345)


def test_argument_validation_needs_no_device():
    lib = capi.load_library()
    assert lib.ubdpt_create(None, None, None, 0, None) == capi.INVALID_ARG
    h = C.c_void_p()
    assert lib.ubdpt_create(None, None, None, 0, C.byref(h)) == capi.INVALID_ARG
    assert lib.ubdpt_destroy(None) == capi.INVALID_ARG
    assert lib.ubdpt_request_stop(None) == capi.INVALID_ARG
    assert lib.ubdpt_query_progress(None, None, None) == capi.INVALID_ARG
    assert lib.ubdpt_read_sum(None, None, 0) == capi.INVALID_ARG

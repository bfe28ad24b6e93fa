"""CPU: the C-ABI library loads, exports every symbol include/ss4k.h declares, and its host-only
entry points work; anything that needs the GPU fails loudly (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _capi.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ss4k.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ss4k_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    assert sorted(_capi.SYMBOLS) == syms, "python binding list and header disagree"
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ss4k.h but not exported"
    assert lib.ss4k_abi_version() == 3


def test_dev_library_is_a_superset_and_product_has_no_bench_hooks(lib):
    """include/ss4k_dev.h: ss4k_bench_conv and the instrumented conv builds live in
    libss4k_hip_dev.so only; the product library does not export them."""
    from sharkshark4k_amd import build as B
    assert not hasattr(lib, "ss4k_bench_conv")
    if not os.path.exists(B.LIB_DEV):
        B.build(dev=True, verbose=False)
    dev = C.CDLL(B.LIB_DEV)
    for s in header_symbols() + _capi.DEV_SYMBOLS:
        assert hasattr(dev, s), f"{s} missing from the dev library"
    text = open(os.path.join(ROOT, "include", "ss4k_dev.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert sorted(set(re.findall(r"\b(ss4k_[a-z0-9_]+)\s*\(", text))) == sorted(_capi.DEV_SYMBOLS)
    # the product library is the smaller one (no debug instantiations)
    assert os.path.getsize(_capi.LIB_PATH) < os.path.getsize(B.LIB_DEV)


def test_struct_layout_matches_header():
    assert C.sizeof(_capi.ModelDesc) == 16 * 4
    assert C.sizeof(_capi.UpscaleCfg) == 8 * 4 + 4 + 4 + 8 + 6 * 4
    assert _capi.UpscaleCfg.denoise_rate.offset == 40
    # bsvd_stream took the first reserved word of ss4k_model_desc (include/ss4k.h)
    assert _capi.ModelDesc.bsvd_stream.offset == 11 * 4
    d = _capi.make_desc(_capi.BSVD, bsvd_stream=True)
    assert d.bsvd_stream == 1 and _capi.make_desc(_capi.BSVD).bsvd_stream == 0


@pytest.mark.parametrize("desc,table", [
    (dict(kind=_capi.FSRCNN, scale=2), lambda: W.fsrcnn_table(0)),
    (dict(kind=_capi.RRDBNET, scale=2, num_block=3), lambda: W.rrdbnet_table(0, scale=2, num_block=3)),
    (dict(kind=_capi.RRDBNET, scale=4, num_block=1), lambda: W.rrdbnet_table(0, scale=4, num_block=1)),
    (dict(kind=_capi.RRDBNET, scale=1, num_block=1), lambda: W.rrdbnet_table(0, scale=1, num_block=1)),
    (dict(kind=_capi.SRVGG, scale=4, num_feat=64, num_block=32), lambda: W.srvgg_table(0)),
    (dict(kind=_capi.SRVGG, scale=2, num_feat=16, num_block=2), lambda: W.srvgg_table(0, num_feat=16, num_conv=2, upscale=2)),
    (dict(kind=_capi.BSVD), lambda: W.bsvd_table(0)),
])
def test_param_count_matches_state_dict(lib, desc, table):
    d = _capi.make_desc(**desc)
    assert _capi.param_count(d) == W.num_params(table())


def test_full_size_param_counts(lib):
    assert _capi.param_count(_capi.make_desc(_capi.RRDBNET, scale=2)) == 16_703_171
    assert _capi.param_count(_capi.make_desc(_capi.RRDBNET, scale=4)) == 16_697_987
    assert _capi.param_count(_capi.make_desc(99)) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(lib):
    h = C.c_void_p()
    rc = lib.ss4k_ctx_create(0, C.byref(h))
    assert rc == -19 and b"no HIP device" in lib.ss4k_last_error()
    with pytest.raises(Exception):
        _capi.Context(0)


def test_weight_tables_are_deterministic():
    a, b = W.rrdbnet_table(3, num_block=1), W.rrdbnet_table(3, num_block=1)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = W.rrdbnet_table(4, num_block=1)
    assert not np.array_equal(a["conv_first.weight"], c["conv_first.weight"])
    flat = W.flatten(a, W.rrdbnet_keys(1))
    assert flat.dtype == np.float32 and flat.size == W.num_params(a)
    blend = W.dni_blend(W.srvgg_table(0, num_conv=1), W.srvgg_table(1, num_conv=1), 0.25)
    k = "body.0.weight"
    assert np.allclose(blend[k], 0.25 * W.srvgg_table(0, num_conv=1)[k] + 0.75 * W.srvgg_table(1, num_conv=1)[k])


def test_model_flag_constants_match_header():
    """ss4k_model_desc.flags: the SS4K_MODEL_* bits the Python binding uses are the header's."""
    text = open(os.path.join(ROOT, "include", "ss4k.h")).read()
    hdr = {m.group(1): int(m.group(2)) for m in re.finditer(r"\bSS4K_MODEL_([A-Z0-9_]+)\s*=\s*(\d+)", text)}
    want = {"FS_EXACT": _capi.MODEL_FS_EXACT, "ONE_CHAIN": _capi.MODEL_ONE_CHAIN, "TWO_CHAINS": _capi.MODEL_TWO_CHAINS,
            "TILE_ROWS_16": _capi.MODEL_TILE_ROWS_16, "TILE_ROWS_20": _capi.MODEL_TILE_ROWS_20, "NO_PAIR": _capi.MODEL_NO_PAIR, "HR_F32": _capi.MODEL_HR_F32,
            "NO_DENSE": _capi.MODEL_NO_DENSE, "NO_WIDE": _capi.MODEL_NO_WIDE, "NO_UPS_PRESUM": _capi.MODEL_NO_UPS_PRESUM, "NO_W16": _capi.MODEL_NO_W16}
    for k, v in want.items():
        assert hdr[k] == v, k
    assert set(hdr) - {"FLAGS_ALL"} == set(want), "a header bit without a Python constant (or the other way round)"
    all_expr = re.search(r"SS4K_MODEL_FLAGS_ALL\s*=\s*([0-9 |]+)", text).group(1)
    assert eval(all_expr) == sum(want.values()) == _capi.MODEL_FLAGS_ALL
    # the kernels that left the product library with ABI 3 keep their bit values in the dev header, outside the product's mask
    dev = open(os.path.join(ROOT, "include", "ss4k_dev.h")).read()
    dhdr = {m.group(1): int(m.group(2)) for m in re.finditer(r"\bSS4K_DEV_MODEL_([A-Z0-9_]+)\s*=\s*(\d+)", dev)}
    assert dhdr["CHAIN"] == _capi.DEV_MODEL_CHAIN and dhdr["CONV5_RS"] == _capi.DEV_MODEL_CONV5_RS
    assert (_capi.DEV_MODEL_CHAIN | _capi.DEV_MODEL_CONV5_RS) & _capi.MODEL_FLAGS_ALL == 0
    assert _capi.ModelDesc.flags.offset == 12 * 4 and _capi.make_desc(_capi.RRDBNET, flags=_capi.MODEL_NO_W16).flags == 32768

"""The C ABI without a GPU: libgptq_amd.so loads, exports every function include/gptq_amd.h
declares (and the ctypes table binds exactly those), and rejects bad arguments before touching the
device.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gptq_amd.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments out
    text = re.sub(r"typedef struct .*?\} \w+;", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gq_\w+)\s*\(", text)))


def test_header_declares_what_we_think():
    names = _declared()
    for must in ("gq_xtx_accumulate", "gq_hessian_invert", "gq_quantize_range", "gq_quantize_layer",
                 "gq_quantize_layer_ex", "gq_quantize_rtn", "gq_dequantize", "gq_pack_rows", "gq_w4a16_gemm", "gq_w4a16_gemm_dt",
                 "gq_xtx_accumulate_f32"):
        assert must in names
    assert len(names) >= 19


def test_library_exports_every_declared_symbol():
    from gptq_v2_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} is declared in gptq_amd.h but not exported"


def test_ctypes_table_matches_the_header():
    from gptq_v2_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load()   # binds restype / argtypes of every entry; AttributeError if one is missing


def test_grid_options_struct_layout():
    """struct gq_grid_options: int, int, int, pointer, int, float -- 8-byte aligned pointer."""
    from gptq_v2_amd import _lib
    o = _lib.GridOptions
    assert (o.sym.offset, o.scale_bits.offset, o.mse.offset) == (0, 4, 8)
    assert o.shrink.offset == 16 and o.n_shrink.offset == 24 and o.norm.offset == 28
    assert ctypes.sizeof(o) == 32


def test_bad_arguments_are_rejected_before_the_device_is_touched():
    from gptq_v2_amd import _lib
    lib = _lib.load()
    assert lib.gq_version() == _lib.ABI_VERSION == int(re.search(r'#define GQ_ABI_VERSION (\d+)', open(HEADER).read()).group(1))
    assert lib.gq_hessian_invert_workspace(0) == 0 and lib.gq_hessian_invert_workspace(4096) > 3 * 4096 * 4096 * 4
    assert lib.gq_quantize_layer_workspace(0, 5) == 0 and lib.gq_quantize_layer_workspace(64, 128) > 0
    assert lib.gq_xtx_accumulate(None, None, 0, 16, 16, 0, 1, None) == -1
    assert b"NULL" in lib.gq_last_error()
    one = ctypes.c_void_p(256)   # never dereferenced: the argument checks come first
    assert lib.gq_xtx_accumulate(one, one, 7, 16, 16, 0, 1, None) == -1          # dtype code
    assert lib.gq_xtx_accumulate(one, one, 0, 16, 16, 1, 0, None) == -1          # save_lower_only without compute_lower_only
    assert lib.gq_w4a16_gemm(one, one, one, one, one, 8, 64, 100, 128, None) == -1   # in_features not a multiple of 64
    assert lib.gq_pack_rows(one, 4, 64, 5, 0, one, None) == -1                    # 5-bit values
    assert lib.gq_xtx_accumulate_f32(one, one, 6, 16, None) == -1                 # token count not a multiple of 4
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(_lib.GqError):
        _lib.check(-2, "x")


def test_dropin_modules_carry_the_references_public_names():
    """dropin/quant.py and dropin/gptq_py.py shadow the reference's modules of the same name: every
    function and class those define at module level (quant.py:7-279, gptq_py.py:9-251) exists here."""
    import importlib.util
    import sys
    drop = os.path.join(ROOT, "gptq_v2_amd", "dropin")

    def load(name):
        spec = importlib.util.spec_from_file_location("dropin_" + name, os.path.join(drop, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    quant = load("quant")
    for name in ("EPSILON", "quantize", "dequantize", "dequantize_quantized", "quantize2", "dequantize2",
                 "dequantize2_quantized2", "Quantizer", "collate_quantizers", "construct_matrix", "construct_matrix_2",
                 "reconstruct_nn_linear"):
        assert hasattr(quant, name), name
    gptq_py = load("gptq_py")
    for name in ("HessianHook", "gptq_quant"):
        assert hasattr(gptq_py, name), name
    q = quant.Quantizer()
    for name in ("find_params", "mse", "quantize", "dequantize", "dequantize_quantized"):
        assert callable(getattr(q, name)), name


def test_quantlinear_concat_and_expand_policy_on_cpu():
    """Host logic of the packed layer that needs no GPU: which token counts take the expand-once forward, and
    that a concatenation stacks the packed words, scales and zero points of its parts row-wise and refuses
    parts that do not share input width, act-order or storage."""
    import torch
    from gptq_v2_amd.quant import QuantLinear

    def fake(rows, c=256, group=128, perm=None, dense3=False):
        layer = QuantLinear(c, rows, group, 3 if dense3 else 4, torch.float16, dense3=dense3)
        layer.qweight = torch.randint(0, 2 ** 31 - 1, (c // 64, rows, 6 if dense3 else 8), dtype=torch.int32)
        layer.scales = torch.rand(c // group, rows).half()
        layer.qzeros = torch.randint(0, 16, (c // group, rows), dtype=torch.uint8)
        layer.perm = perm
        return layer

    small, big = fake(4096), fake(8192)
    least = QuantLinear.expand_min_tokens
    assert least > 0
    assert big._expands(least) and not big._expands(least - 1)
    assert small._expands(2 * least) and not small._expands(2 * least - 1)
    small.expand_min_tokens = 0
    assert not small._expands(1 << 20)

    perm = torch.randperm(256).to(torch.int32)
    parts = [fake(64, perm=perm), fake(32, perm=perm), fake(96, perm=perm)]
    fused = QuantLinear.concat(parts)
    assert fused.out_features == 192 and fused.in_features == 256 and fused.perm is perm
    assert torch.equal(fused.qweight[:, 64:96], parts[1].qweight) and torch.equal(fused.scales[:, 96:], parts[2].scales)
    assert torch.equal(fused.qzeros[:, :64], parts[0].qzeros) and fused.qweight.is_contiguous()
    import pytest
    for other in (fake(64, c=128, perm=perm[:128].contiguous()), fake(64, perm=torch.roll(perm, 1)), fake(64, perm=None),
                  fake(64, group=64, perm=perm), fake(64, perm=perm, dense3=True)):
        with pytest.raises(ValueError):
            QuantLinear.concat([parts[0], other])

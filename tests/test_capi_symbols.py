"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/vaa.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from roboticattack_amd import _lib


def _declared(header="vaa.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vaa_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.EXPORTS)
    assert _declared("vaa_model_ops.h") == sorted(_lib.MODEL_OP_EXPORTS)  # the optional model-side operators


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared() + _declared("vaa_model_ops.h"):
        assert hasattr(L, name), f"libvaa_hip.so does not export {name}"
    assert _lib.lib().vaa_version() >= 100


def test_sizes_and_argument_errors_without_gpu():
    """Pure host-side entry points and argument validation (no kernel is launched)."""
    L = _lib.lib()
    assert L.vaa_patch_grad_ws_bytes(64, 50, 50) == 64 * 3 * 50 * 50 * 4  # one partial tile per image (its row bands are separate workgroups)
    assert L.vaa_patch_grad_ws_bytes(4096, 50, 50) == 512 * 3 * 50 * 50 * 4
    assert L.vaa_patch_grad_ws_bytes(0, 50, 50) == 0
    # the larger of the label-driven schedule's per-position statistics (36 B) and the ROWS route (row map 16 + 16 B/position, rounded to
    # 256 B, + 4 part statistics and one slice statistic per row)
    assert L.vaa_loss_ws_bytes(64, 45) == max(64 * 44 * 36, (16 + 64 * 44 * 16 + 255) // 256 * 256 + 64 * 44 * (4 * 16 + 16))
    assert L.vaa_loss_rowmap_bytes(64, 45) == 16 + 64 * 44 * 16 and L.vaa_loss_rows_ws_bytes(128) == 128 * 80
    assert L.vaa_patch_resize_ws_bytes(4, 100, 100) == 4 * 3 * 100 * 100 * 4 and L.vaa_patch_resize_ws_bytes(1, 100, 100) == 0  # one partial per image group; a single group writes gpatch itself
    # K2': packed weights = 37 column blocks x 16 columns (588 + 4 of padding) x D; towers must be multiples of 64 wide
    assert L.vaa_patch_embed_packed_elems(1024) == 592 * 1024 and L.vaa_patch_embed_packed_elems(96) == 0 and L.vaa_patch_embed_packed_elems(0) == 0
    assert L.vaa_patch_embed_pack_weights(None, 1024, None, None) == -1 and b"null pointer" in L.vaa_last_error()
    assert L.vaa_patch_embed_grad_ws_bytes(64, 50, 50) == L.vaa_patch_grad_ws_bytes(64, 50, 50) + 2 * 64 * 256 * 588 * 4 + 256  # partials + a tile-gradient buffer per tower
    assert L.vaa_patch_embed_grad_multi_ws_bytes(4) == 2 * 4 * 256 * 588 * 4 + 256
    # K3s: shapes it covers and its scratch (two logits + the launch tag per 64-bit word, whole row blocks)
    assert L.vaa_head_slice_applies(128, 4096, 32064) == 1 and L.vaa_head_slice_applies(129, 4096, 32064) == 0 and L.vaa_head_slice_applies(16, 4160, 32064) == 0
    assert L.vaa_head_slice_applies(16, 64, 32064) == 1 and L.vaa_head_slice_applies(16, 64, 40000) == 0 and L.vaa_head_slice_applies(0, 64, 32064) == 0
    assert L.vaa_head_slice_ws_bytes(1) == 16 * 1024 and L.vaa_head_slice_ws_bytes(128) == 128 * 1024 and L.vaa_head_slice_ws_bytes(0) == 0
    assert L.vaa_head_slice_pack(None, 4096, 32064, None, None) == -1 and b"null pointer" in L.vaa_last_error()
    assert L.vaa_head_slice_fwd_bwd(None, None, None, 4096, None, 16, 8, 30, 32064, 1, _lib.f32x([5, .8, .2, 1]), None, None, None, 0, None, None, None, None, 0, None) == -1
    # empty batches return before anything is validated or launched
    assert L.vaa_patch_embed_grad_gather_multi(None, 64, None, 64, None, None, None, None, None, None, None, 0, 50, 50, 1, 0, _lib.f32x([1] * 6), 1, None, None, 0, None) == 0
    assert L.vaa_patch_grad_gather_multi(None, None, None, None, None, None, 0, 50, 50, 1, 0, _lib.f32x([1] * 6), None, None) == 0
    assert L.vaa_patch_embed_grad_gather_multi(None, 64, None, 64, None, None, None, None, None, None, None, 2, 50, 50, 1, 0, _lib.f32x([1] * 6), 1, None, None, 0, None) == -1
    rc = L.vaa_patch_apply_fwd(None, None, None, None, 1, 50, 50, 1, 0, _lib.f32x([0] * 6), _lib.f32x([1] * 6), None, None, None)
    assert rc == -1 and b"null pointer" in L.vaa_last_error()
    rc = L.vaa_patch_update(None, None, None, None, 10, 0, 1e-3, 0.9, 0.999, 1e-6, 1, 0.0, 1.0, None, None)
    assert rc == -1
    with pytest.raises(_lib.VaaError):
        _lib.check(rc, "vaa_patch_update")


def test_ops_refuse_cpu_tensors():
    import torch

    from roboticattack_amd import ops

    with pytest.raises(_lib.VaaError, match="no CPU fallback"):
        ops.patch_apply_fwd(torch.zeros(1, 224, 224, 3, dtype=torch.uint8), torch.zeros(3, 50, 50), torch.zeros(1, 2, dtype=torch.int32),
                            torch.zeros(1, 6), True)


def test_missing_library_fails_loudly(tmp_path):
    """No HIP extension -> the product path raises (there is no CPU fallback to drop into)."""
    import subprocess
    import sys

    code = ("import os; os.environ['VAA_LIB_PATH'] = %r\n"
            "from roboticattack_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept _lib.VaaError as e:\n    print('RAISED', e)\n") % str(tmp_path / "absent.so")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "RAISED" in out.stdout and "no CPU fallback" in out.stdout, out.stdout + out.stderr

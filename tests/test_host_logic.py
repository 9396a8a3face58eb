"""CPU-only tests of the PRODUCT host code (roboticattack_amd/*) against the reference's golden vectors:
RNG draw order, label masking, tokenizer, LR schedule, CLI surface, patch.pt format, synthetic batch contract."""
import json
import os
import random
import runpy
import zipfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from roboticattack_amd import synthetic
from roboticattack_amd.action_tokenizer import ActionTokenizer
from roboticattack_amd.labels import mask_labels, tma_target_labels, tma_target_tokens
from roboticattack_amd.optim import cosine_with_warmup_lambda
from roboticattack_amd.transform import RandomPatchTransform


def test_transform_rng_draw_order_seed42():
    d = np.load(os.path.join(GOLDEN, "rng_stream_seed42.npz"))
    random.seed(42)
    np.random.seed(42)
    t = RandomPatchTransform("cpu")
    xy, th = t._draw(len(d["xy"]), 50, 50, True)
    assert np.array_equal(xy, d["xy"])
    assert np.array_equal(th.reshape(-1, 2, 3), d["theta"][:, :2, :])
    # geometry=False consumes only the two randint draws per image
    random.seed(42)
    np.random.seed(42)
    st = np.random.get_state()[1].copy()
    xy2, th2 = t._draw(4, 50, 50, False)
    assert np.array_equal(xy2, d["xy"][:4]) and np.array_equal(np.random.get_state()[1], st)
    assert np.array_equal(th2, np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (4, 1)))


def test_labels_and_tokenizer_vs_reference():
    d = np.load(os.path.join(GOLDEN, "labels_tokenizer.npz"))
    lab = torch.from_numpy(d["labels_in"])
    for tag, mi in (("0", [0]), ("012", [0, 1, 2]), ("6", [6]), ("all", list(range(7))), ("25", [2, 5])):
        assert np.array_equal(mask_labels(lab.clone(), mi).numpy(), d[f"uada_mask_{tag}"])
    at = ActionTokenizer()
    assert at.action_token_begin_idx == int(d["begin_idx"]) == 31743
    assert np.array_equal(at.bin_centers, d["bin_centers"])
    assert np.array_equal(at.decode_token_ids_to_actions(d["tokens"]), d["decoded"])
    with pytest.raises(RuntimeError):
        mask_labels(torch.full((2, 10), -100, dtype=torch.int64), [0])  # the reference's view(-1, 7) would raise too
    for tag in ("t0", "t012"):
        t = np.load(os.path.join(GOLDEN, f"k3_tma_{tag}.npz"))
        tt = tma_target_tokens(np.ones(7) * float(t["target_action"]), list(t["maskidx"]))
        assert np.array_equal(tt.numpy(), t["target_tokens"])
        assert np.array_equal(tma_target_labels(torch.from_numpy(t["labels"]), tt).numpy(), t["newlabels"])


def test_relative_distance_metric():
    from roboticattack_amd.attack.engine import AttackBase

    d = np.load(os.path.join(GOLDEN, "labels_tokenizer.npz"))
    rd = AttackBase.calculate_relative_distance(None, torch.from_numpy(d["rd_pred"]), torch.from_numpy(d["rd_gt"]), [0, 3], {"0": [], "3": []})
    np.testing.assert_allclose(rd["0"], d["rd_0"], rtol=1e-12)
    np.testing.assert_allclose(rd["3"], d["rd_3"], rtol=1e-12)


def test_cosine_schedule_vs_transformers_golden():
    d = np.load(os.path.join(GOLDEN, "sched.npz"))
    for tag, warm, total in (("w20_t2000", 20, 2000), ("w200_t10000", 200, 10000), ("w2_t4", 2, 4)):
        got = np.array([cosine_with_warmup_lambda(s, warm, total) for s in range(len(d[tag]))])
        np.testing.assert_allclose(got, d[tag], rtol=0, atol=1e-15)
    assert cosine_with_warmup_lambda(0, 20, 2000) == 0.0  # lr = 0 during the whole first outer iteration


def test_cli_surface_matches_reference_defaults():
    want = {
        "UADA_wrapper": dict(maskidx=[0], lr=1e-3, device=1, iter=2000, accumulate=1, bs=8, warmup=20, geometry=True, patch_size=[3, 50, 50],
                             innerLoop=50, dataset="bridge_orig", resize_patch=False, filterGripTrainTo1=False),
        "UADA_wrapper_ddp": dict(maskidx=[0], lr=1e-3, iter=2000, MSE_weights=5, bs=8, warmup=20, innerLoop=50, geometry=True),
        "TMA_wrapper": dict(maskidx=[0], lr=2e-3, device=0, iter=2000, warmup=20, innerLoop=50, targetAction=0, server="xxx"),
        "UPA_wrapper": dict(maskidx=[0, 1, 2], lr=2e-3, device=1, iter=10000, warmup=200, innerLoop=100, alpha=0.8, belta=0.2,
                            reverse_direction=True),
    }
    for w, exp in want.items():
        mod = runpy.run_path(os.path.join(ROOT, "VLAAttacker", f"{w}.py"), run_name="not_main")
        a = mod["arg_parser"]([])
        for k, v in exp.items():
            assert getattr(a, k) == v, (w, k, getattr(a, k), v)
    a = runpy.run_path(os.path.join(ROOT, "VLAAttacker", "UADA_wrapper_ddp.py"), run_name="x")["arg_parser"](
        ["--maskidx", "0,1,2", "--geometry", "false", "--patch_size", "3,100,100", "--wandb_project", "false"])
    assert a.maskidx == [0, 1, 2] and a.geometry is False and a.patch_size == [3, 100, 100] and not hasattr(a, "device")


def test_patch_pt_format_matches_released_patch(tmp_path):
    """a-12: `torch.save(patch.detach().cpu())` -> fp32 contiguous [3,ph,pw], same zip entries/sizes as a released file."""
    from roboticattack_amd.attack.engine import AttackBase

    meta = json.load(open(os.path.join(GOLDEN, "released_patch_meta.json")))
    rel = torch.load(os.path.join(GOLDEN, "released_patch_T-dof1.pt"), map_location="cpu")
    assert str(rel.dtype) == meta["dtype"] and list(rel.shape) == meta["shape"] and 0.0 <= float(rel.min()) and float(rel.max()) <= 1.0
    obj = AttackBase.__new__(AttackBase)
    obj.save_dir = str(tmp_path)
    p = torch.rand(3, 50, 50, requires_grad=True)
    d = AttackBase.save_patch(obj, p, "last")
    f = os.path.join(d, "patch.pt")
    back = torch.load(f, map_location="cpu")
    assert back.dtype == torch.float32 and tuple(back.shape) == (3, 50, 50) and not back.requires_grad and torch.equal(back, p.detach())
    with zipfile.ZipFile(f) as z:
        ours = {os.path.basename(i.filename) if "/data/" not in i.filename else "data/0": i.file_size for i in z.infolist()}
    theirs = {os.path.basename(n) if "/data/" not in n else "data/0": s for n, s in meta["entries"]}
    assert ours["data/0"] == theirs["data/0"] == 30000
    assert {"data.pkl", "byteorder", "version"} <= set(ours) and {"data.pkl", "byteorder", "version"} <= set(theirs)


def test_synthetic_batch_contract():
    b = synthetic.synth_batch(3, 5)
    assert len(b["pixel_values"]) == 5 and b["pixel_values"][0].size == (224, 224) and b["pixel_values"][0].mode == "RGB"
    ids, lab, att = b["input_ids"], b["labels"], b["attention_mask"]
    assert ids.dtype == lab.dtype == torch.int64 and att.dtype == torch.bool and ids.shape == lab.shape == att.shape
    assert bool((ids[:, 0] == 1).all()) and bool(((ids == 32000) == ~att).all())
    for r in range(5):
        real = lab[r][lab[r] != -100]
        assert len(real) == 8 and real[-1] == 2 and bool(((real[:7] >= 31744) & (real[:7] <= 31999)).all())
    assert np.array_equal(synthetic.synth_images(9, 2), synthetic.synth_images(9, 2))


def test_tiny_model_rows_equal_full_logits():
    from roboticattack_amd.openvla_model import OpenVLAShaped, tiny_cfg

    m = OpenVLAShaped(tiny_cfg()).init_random(0).eval()
    ids, labels, attn = synthetic.synth_text_batch(1, 3, 18, 24)
    labels = mask_labels(labels, [0, 2])
    pix = torch.randn(3, 6, 224, 224)
    out = m(ids, attn, pix, labels)
    rows = m.forward_rows(ids, pix, labels)
    L = labels.shape[1]
    sel = [(b, 256 + k) for b in range(3) for k in range(L - 1) if labels[b, k + 1] != -100]
    full = torch.stack([out.logits[b, p] for b, p in sel])
    assert torch.allclose(full, rows, atol=1e-5) and out.logits.shape == (3, 256 + L, 32064)
    # the rows-only last layer gives the same pixel gradient as the full sequence
    gs = []
    for use_rows in (True, False):
        p = pix.clone().requires_grad_(True)
        if use_rows:
            z = m.forward_rows(ids, p, labels)
        else:
            lg = m(ids, attn, p, labels).logits
            z = torch.stack([lg[b, q] for b, q in sel])
        z.square().mean().backward()
        gs.append(p.grad)
    assert torch.allclose(gs[0], gs[1], rtol=1e-4, atol=1e-9 + 1e-4 * float(gs[1].abs().max()))
    assert all(not p.requires_grad for p in m.parameters())


def test_prompt_length_bucket_keeps_the_labelled_rows():
    """cfg.seq_floor / seq_multiple pad the prompts of the rows-only path to a canonical length (stable GEMM shapes -> the shipped
    hipBLASLt selections apply to every batch); causal attention: the labelled rows must not change."""
    import dataclasses

    from roboticattack_amd.openvla_model import OpenVLAShaped, openvla_7b_cfg, tiny_cfg

    assert openvla_7b_cfg().seq_floor == 44 and openvla_7b_cfg().seq_multiple == 4
    plain = OpenVLAShaped(tiny_cfg()).init_random(0).eval()
    buck = OpenVLAShaped(dataclasses.replace(tiny_cfg(), seq_floor=32, seq_multiple=4)).eval()
    buck.load_state_dict(plain.state_dict())
    ids, labels, attn = synthetic.synth_text_batch(1, 3, 18, 24)
    labels = mask_labels(labels, [0, 2])
    L = labels.shape[1]
    assert plain.seq_bucket(L) == L and buck.seq_bucket(L) == 32 and buck.seq_bucket(33) == 36 and buck.seq_bucket(44) == 44
    pix = torch.randn(3, 6, 224, 224)
    a = plain.forward_rows(ids, pix, labels)
    b = buck.forward_rows(ids, pix, labels)
    assert a.shape == b.shape and torch.allclose(a, b, atol=1e-5)
    ri = buck.label_row_index(labels)
    assert int(ri.max()) < 3 * (256 + 32) and torch.equal(ri % (256 + 32), plain.label_row_index(labels) % (256 + L))


def test_upa_change_target_matches_reference():
    """UPA.py:358-364 (guide mode): the reference's sequential in-place assignment and its single torch.randint draw."""
    import types

    from roboticattack_amd.attack.upa import OpenVLAAttacker

    d = np.load(os.path.join(GOLDEN, "labels_tokenizer.npz"))
    torch.manual_seed(5)
    got = OpenVLAAttacker.change_target(types.SimpleNamespace(), torch.from_numpy(d["change_target_in"]).clone())
    assert np.array_equal(got.numpy(), d["change_target_out"])
    assert np.array_equal(torch.rand(1).numpy(), d["change_target_rng_after"]), "RNG consumption differs from the reference"


def test_hf_checkpoint_name_mapping_roundtrip(tmp_path):
    """`load_hf_openvla` maps an HF-OpenVLA-named safetensors checkpoint (modeling_prismatic.py module tree: vision_backbone.
    {featurizer,fused_featurizer}.blocks.N.attn.qkv ..., projector.fcK, language_model.model.layers.N.self_attn.q_proj ...) onto the
    plain module tree. No real weights exist in this image, so a tiny checkpoint with HF names is synthesised from a reference
    instance and must load back bit-identically (logits equal)."""
    from safetensors.torch import save_file

    from roboticattack_amd.openvla_model import OpenVLAShaped, load_hf_openvla, tiny_cfg

    src = OpenVLAShaped(tiny_cfg()).init_random(3).eval()
    hf = {}
    for name, p in src.named_parameters():
        n = name
        if n.startswith("featurizer.") or n.startswith("fused_featurizer."):
            pre, rest = n.split(".", 1)
            if rest == "prefix":
                hf[f"vision_backbone.{pre}.cls_token"] = p[:, :1].clone()
                hf[f"vision_backbone.{pre}.reg_token"] = p[:, 1:].clone()
                continue
            rest = rest.replace("patch_embed.", "patch_embed.proj.")
            rest = rest.replace(".qkv.", ".attn.qkv.").replace(".proj.weight", ".attn.proj.weight").replace(".proj.bias", ".attn.proj.bias")
            rest = rest.replace("patch_embed.attn.proj", "patch_embed.proj")  # undo the attn rename on the patch embed
            rest = rest.replace(".fc1.", ".mlp.fc1.").replace(".fc2.", ".mlp.fc2.").replace(".ls1", ".ls1.scale_factor").replace(".ls2", ".ls2.scale_factor")
            hf[f"vision_backbone.{pre}.{rest}"] = p.detach().clone()
        elif n.startswith("fc"):
            hf[f"projector.{n}"] = p.detach().clone()
        elif n.startswith("lm_head."):
            hf[f"language_model.{n}"] = p.detach().clone()
        else:
            n2 = n
            for proj in ("q_proj", "k_proj", "v_proj", "o_proj"):
                n2 = n2.replace(f".{proj}.", f".self_attn.{proj}.")
            for proj in ("gate_proj", "up_proj", "down_proj"):
                n2 = n2.replace(f".{proj}.", f".mlp.{proj}.")
            hf[f"language_model.model.{n2}"] = p.detach().clone()
    save_file({k: v.contiguous() for k, v in hf.items()}, str(tmp_path / "model-00001-of-00001.safetensors"))
    # the checkpoint's config.json is read and checked against the model's constants (configuration_prismatic.py:83-123)
    import json

    tc = tiny_cfg()
    with pytest.raises(ValueError, match="no config.json"):
        load_hf_openvla(OpenVLAShaped(tiny_cfg()).eval(), str(tmp_path))
    conf = {"pad_token_id": 32000, "image_sizes": [224, 224], "use_fused_vision_backbone": True, "vision_backbone_id": "dinosiglip-vit-so-224px",
            "text_config": {"hidden_size": tc.llm_dim, "num_hidden_layers": tc.llm_layers, "num_attention_heads": tc.llm_heads, "intermediate_size": tc.llm_mlp,
                            "vocab_size": 32064, "rms_norm_eps": 1e-06, "pad_token_id": 32000}}
    for key, val, where in (("rms_norm_eps", 1e-05, "rms_norm_eps"), ("rope_theta", 500000.0, "rope_theta"), ("hidden_size", 4096, "hidden_size"),
                            ("pad_token_id", 0, "text_config.pad_token_id"), ("num_key_value_heads", 1, "num_key_value_heads")):
        badc = json.loads(json.dumps(conf))
        badc["text_config"][key] = val
        (tmp_path / "config.json").write_text(json.dumps(badc))
        with pytest.raises(ValueError, match=where):
            load_hf_openvla(OpenVLAShaped(tiny_cfg()).eval(), str(tmp_path))
    (tmp_path / "config.json").write_text(json.dumps(dict(conf, image_sizes=[384, 384])))
    with pytest.raises(ValueError, match="image_sizes"):
        load_hf_openvla(OpenVLAShaped(tiny_cfg()).eval(), str(tmp_path))
    (tmp_path / "config.json").write_text(json.dumps(conf))
    dst = load_hf_openvla(OpenVLAShaped(tiny_cfg()).eval(), str(tmp_path))
    for (n1, p1), (n2, p2) in zip(src.named_parameters(), dst.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    ids, labels, attn = synthetic.synth_text_batch(2, 2, 18, 20)
    pix = torch.randn(2, 6, 224, 224)
    assert torch.equal(src(ids, attn, pix, labels).logits, dst(ids, attn, pix, labels).logits)
    # the shapes a real openvla-7b checkpoint holds for the DINOv2-reg4 tower (timm no_embed_class): pos_embed over the 256 patch tokens
    # only, a 1-token cls_token and a 4-token reg_token; SigLIP: pos_embed 256, no prefix tokens
    assert tuple(hf["vision_backbone.featurizer.pos_embed"].shape) == (1, 256, 64)
    assert tuple(hf["vision_backbone.featurizer.cls_token"].shape) == (1, 1, 64) and tuple(hf["vision_backbone.featurizer.reg_token"].shape) == (1, 4, 64)
    assert tuple(hf["vision_backbone.fused_featurizer.pos_embed"].shape) == (1, 256, 128)
    from roboticattack_amd.openvla_model import openvla_7b_cfg

    c7 = openvla_7b_cfg()
    assert (c7.dino.n_prefix, c7.dino.cls_pos, c7.siglip.n_prefix) == (5, False, 0)


def test_dinov2_reg4_prefix_and_pos_embed_order():
    """timm `no_embed_class` semantics of the DINOv2-reg4 tower: x = cat([cls, reg x4, patches + pos_embed]) — the position embedding
    never touches the prefix tokens (a 257-entry pos_embed cannot load the real checkpoint)."""
    from roboticattack_amd.openvla_model import Vit, VitCfg

    torch.manual_seed(0)
    v = Vit(VitCfg(16, 2, 2, 32, 5, False, True)).eval()
    with torch.no_grad():
        v.pos_embed.normal_()
        v.prefix.normal_()
    assert tuple(v.pos_embed.shape) == (1, 256, 16)
    img = torch.randn(2, 3, 224, 224)
    seen = {}
    v.blocks[0].register_forward_pre_hook(lambda m, a: seen.setdefault("x", a[0].detach().clone()))
    v(img)
    x = seen["x"]
    tiles = img.reshape(2, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(2, 256, 588)
    emb = torch.nn.functional.linear(tiles, v.patch_embed.weight.reshape(16, 588), v.patch_embed.bias)
    assert x.shape == (2, 261, 16)
    assert torch.equal(x[:, :5], v.prefix.expand(2, -1, -1)) and torch.allclose(x[:, 5:], emb + v.pos_embed, atol=1e-6)


def test_frozen_linears_fn_matches_autograd():
    """model_ops.FrozenLinearsFn (TN-layout dgrad through a resident W^T, epilogue-fused sums) == plain autograd, fp32 on CPU."""
    import torch

    from roboticattack_amd.model_ops import FrozenLinearsFn

    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 8, generator=g, requires_grad=True)
    res = torch.randn(3, 5, 6, generator=g, requires_grad=True)
    Ws = [torch.randn(6, 8, generator=g) for _ in range(3)]
    ws = []
    for w in Ws:
        ws += [w, w.t().contiguous()]
    outs = FrozenLinearsFn.apply(x, None, *ws)
    sum((o * (i + 1)).sin().sum() for i, o in enumerate(outs)).backward()
    gx = x.grad.clone()
    x.grad = None
    ref = [torch.nn.functional.linear(x, w) for w in Ws]
    sum((o * (i + 1)).sin().sum() for i, o in enumerate(ref)).backward()
    for o, r in zip(outs, ref):
        assert torch.allclose(o, r, atol=1e-6)
    assert torch.allclose(gx, x.grad, atol=1e-5)
    x.grad = None
    (y,) = FrozenLinearsFn.apply(x, res, Ws[0], Ws[0].t().contiguous())
    y.cos().sum().backward()
    gx, gr = x.grad.clone(), res.grad.clone()
    x.grad = res.grad = None
    (res + torch.nn.functional.linear(x, Ws[0])).cos().sum().backward()
    assert torch.allclose(gx, x.grad, atol=1e-5) and torch.allclose(gr, res.grad, atol=1e-6)


def test_vit_accepts_precomputed_patch_embeds():
    """`patch_embeds=` (the hand-over format of ops.PatchApplyEmbed) is the same function of the pixels as `pixel_values=`."""
    from roboticattack_amd import ops
    from roboticattack_amd.openvla_model import OpenVLAShaped, tiny_cfg

    m = OpenVLAShaped(tiny_cfg()).init_random(1).eval()
    ids, labels, _ = synthetic.synth_text_batch(2, 2, 18, 24)
    labels = mask_labels(labels, [0])
    pix = torch.randn(2, 6, 224, 224)
    w0, b0, _ = m.featurizer.embed_params()
    w1, b1, _ = m.fused_featurizer.embed_params()
    e = (torch.nn.functional.linear(ops.unfold_tiles(pix[:, :3]), w0, b0), torch.nn.functional.linear(ops.unfold_tiles(pix[:, 3:]), w1, b1))
    assert torch.allclose(m.forward_rows(ids, pix, labels), m.forward_rows(ids, None, labels, patch_embeds=e), atol=1e-5)
    assert m.patch_embed_params() is None  # CPU / fp32 / widths not multiples of 64: the fused backward does not apply


def test_bench_compact_line_is_strict_json():
    """bench.py's stdout record (the driver parses it): < 4 KB, ASCII, strict JSON — non-finite floats become null, optional keys are dropped
    in order when the line would not fit, a line that still does not fit or carries NaN / Infinity anywhere is refused."""
    import importlib.util
    import json
    import os

    import pytest

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def refuse(name):
        raise AssertionError(name)

    rec = {"metric": "m", "value": float("nan"), "ms_per_step": float("inf"), "config": {"a": np.float32(1.23456789), "n": np.int64(3), "pad": "x" * 3000,
                                                                                          "pad2": "y" * 3000}, "t": (1, 2.5)}
    line = bench.encode_line(rec, optional=["config.pad", "config.pad2"])
    assert len(line) < bench.LINE_LIMIT and line.isascii() and "NaN" not in line and "Infinity" not in line and "\n" not in line
    d = json.loads(line, parse_constant=refuse)
    assert d["value"] is None and d["ms_per_step"] is None and d["config"]["n"] == 3 and abs(d["config"]["a"] - 1.23457) < 1e-6
    assert "pad" not in d["config"] and "pad2" in d["config"] and d["t"] == [1, 2.5]  # least important first, only as many as needed
    with pytest.raises(RuntimeError):
        bench.encode_line({"note": "x" * 5000})
    with pytest.raises(RuntimeError):
        bench.encode_line({"note": "256 MB Infinity Cache"})
    with pytest.raises(RuntimeError):
        bench.encode_line({"note": "café NaN"})

"""Checkpoint I/O against the reference's own converter output
(tests/golden/convert_hf_llama.safetensors, made by tests/golden/make_golden.py --only convert)."""
import json

import pytest
import torch

from conftest import load_golden
from dualhyp_amd import Config
from dualhyp_amd import checkpoint as ck


def _fixture():
    t, meta = load_golden("convert_hf_llama")
    cfg = Config(**meta["config"])
    hf = {k[3:]: v for k, v in t.items() if k.startswith("hf.")}
    lit = {k[4:]: v for k, v in t.items() if k.startswith("lit.")}
    return t, meta, cfg, hf, lit


def test_convert_equals_the_reference_converter():
    t, meta, cfg, hf, lit = _fixture()
    got = ck.convert_hf_llama(hf, cfg)
    assert set(got) == set(lit)
    for k in lit:
        assert torch.equal(got[k], lit[k]), k
    # q and k/v of a layer arriving in different shards, in either order
    late = set(meta["shard2_keys"])
    s1, s2 = {k: v for k, v in hf.items() if k not in late}, {k: v for k, v in hf.items() if k in late}
    for shards in ([s1, s2], [s2, s1]):
        got = ck.convert_hf_llama(shards, cfg)
        assert all(torch.equal(got[k], lit[k]) for k in lit)
    # tied embeddings: no lm_head in the HF files
    got = ck.convert_hf_llama({k: v for k, v in hf.items() if k != "lm_head.weight"}, cfg)
    assert torch.equal(got["lm_head.weight"], t["lit_tied.lm_head.weight"])
    # dtype conversion and a missing projection
    assert ck.convert_hf_llama(hf, cfg, dtype=torch.bfloat16)["transformer.h.1.attn.attn.weight"].dtype == torch.bfloat16
    with pytest.raises(ValueError):
        ck.convert_hf_llama({k: v for k, v in hf.items() if "layers.1.self_attn.v_proj" not in k}, cfg)
    with pytest.raises(KeyError):
        ck.convert_hf_llama({**hf, "model.layers.0.self_attn.q_norm.weight": hf["model.norm.weight"]}, cfg)


def test_interleave_roundtrip_and_export():
    _, _, cfg, hf, lit = _fixture()
    q, k, v = (hf[f"model.layers.0.self_attn.{n}_proj.weight"] for n in "qkv")
    fused = ck.interleave_qkv(q, k, v, cfg)
    assert torch.equal(fused, lit["transformer.h.0.attn.attn.weight"])
    q2, k2, v2 = ck.split_qkv(fused, cfg)
    assert torch.equal(q, q2) and torch.equal(k, k2) and torch.equal(v, v2)
    # group layout: rows of group g are [q heads g*q_per_kv.., k head g, v head g]
    hs, qpk = cfg.head_size, cfg.n_head // cfg.n_query_groups
    g = 1
    blk = fused[g * (qpk + 2) * hs:(g + 1) * (qpk + 2) * hs]
    assert torch.equal(blk[: qpk * hs], q[g * qpk * hs:(g + 1) * qpk * hs])
    assert torch.equal(blk[qpk * hs:(qpk + 1) * hs], k[g * hs:(g + 1) * hs])
    assert torch.equal(blk[(qpk + 1) * hs:], v[g * hs:(g + 1) * hs])
    back = ck.export_hf_llama(lit, cfg)
    assert set(back) == {k for k in hf if "inv_freq" not in k}
    assert all(torch.equal(back[k], hf[k]) for k in back)


def test_checkpoint_files_roundtrip(tmp_path):
    """lit_model.pth (flat) and best_model.pth ({"model": sd}) as the reference reads / writes them,
    through the model's own load_state_dict (old `attn.attn.weight`-style keys included)."""
    from dualhyp_amd import GPT
    from dualhyp_amd.synth import synth_state_dict
    base = Config.from_name("parity-tiny")
    sd0 = {k: v for k, v in synth_state_dict(base, seed=3, device="cpu").items()}
    hf = ck.export_hf_llama(sd0, base)
    assert "model.layers.1.self_attn.k_proj.weight" in hf and not any("lora" in k for k in hf)
    d = tmp_path / "parity-tiny"
    d.mkdir()
    items = list(hf.items())
    torch.save(dict(items[:9]), d / "pytorch_model-00001-of-00002.bin")
    torch.save(dict(items[9:]), d / "pytorch_model-00002-of-00002.bin")
    torch.save({"args": 1}, d / "training_args.bin")
    out = ck.convert_hf_checkpoint(d, dtype="bfloat16")
    assert out.name == "lit_model.pth" and json.loads((d / "lit_config.json").read_text())["n_embd"] == base.n_embd
    sd = ck.load_checkpoint(out)
    assert "transformer.h.0.attn.attn.weight" in sd          # the reference's base-checkpoint key style
    cfg = Config.from_checkpoint(d, r=4, alpha=8, to_query=True, to_key=True, to_value=True, to_projection=True)
    m = GPT(cfg).to(torch.bfloat16)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lora_" in k or "adapter_" in k for k in missing)
    for k, v in m.state_dict().items():
        if "lora_" not in k and "adapter_" not in k:
            assert torch.equal(v, sd0[k].bfloat16()), k
    torch.nn.init.normal_(m.transformer.h[0].attn.attn.lora_B)
    ck.save_checkpoint(m, tmp_path / "runs" / "best_model.pth")
    raw = torch.load(tmp_path / "runs" / "best_model.pth", weights_only=True)
    assert set(raw) == {"model"} and "transformer.h.0.attn.attn.lora_B" in raw["model"]
    m2 = GPT(cfg).to(torch.bfloat16)
    x, y = m2.load_state_dict(ck.load_checkpoint(tmp_path / "runs" / "best_model.pth"), strict=False)
    assert not x and not y
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    ck.save_checkpoint(m, tmp_path / "lora.pth", lora_only=True)
    assert all("lora_" in k for k in ck.load_checkpoint(tmp_path / "lora.pth"))
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(tmp_path / "nope.pth")

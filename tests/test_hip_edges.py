"""Edge cases of the decode path on the GPU against the oracle (CPU restatement, test infrastructure): the limits
generate/base.py:41-47 draws (full context, `max_returned_tokens` one past the cache, empty continuation), one-token
prompts, EOS on the first generated token, and one ragged batch mixing all of them.  Weights are the un-tied synthetic
ones (norm jitter, weight scale 4): logits and ids depend on the whole context, so a wrong rope row or cache slot at
the last position shows.  Logit gate as in test_hip_model.py: HIP vs oracle-bf16 no further apart than oracle-bf16 vs
oracle-fp32 on the same teacher-forced ids."""
import pytest
import torch

from conftest import record_parity
from dualhyp_amd import GPT, Config, generate, generate_batch
from dualhyp_amd.synth import synth_state_dict, synth_prompts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SAFE_MARGIN_ULPS = 4


def rel_rms(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def margin_ulps(row: torch.Tensor) -> float:
    """top-1 minus top-2 of a bf16 logit row in bf16 ulps at the top value"""
    v = row.float().topk(2).values
    ulp = 2.0 ** (torch.floor(torch.log2(v[0].abs().clamp_min(1e-30))) - 7)
    return float((v[0] - v[1]) / ulp)


@pytest.fixture(scope="module")
def tiny():
    from oracle import ger_oracle as O
    cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
    sd = synth_state_dict(cfg, seed=1337, norm_jitter=0.25, weight_scale=4.0)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
    m.eval()
    m.cpu_rsqrt_vec_width = 32          # the oracle runs on the CPU: torch's CPU bf16 rsqrt tail rounding (Q11)
    return cfg, m, O, O.OracleGPT(cfg, sd), O.OracleGPT(cfg, {k: v.float() for k, v in sd.items()})


def teacher_forced(tiny, idx, n_new):
    """prefill + n_new single-token steps fed the bf16 oracle's greedy ids: (hip, oracle bf16, oracle fp32) logits per
    step [n_new, V] and the ids"""
    cfg, m, O, o16, o32 = tiny
    T = idx.numel()
    rows = ([], [], [])
    ids = []
    with torch.no_grad():
        m.reset_cache(); o16.reset_cache(); o32.reset_cache()
        x, pos = idx.view(1, -1), torch.arange(T)
        for s in range(n_new):
            got = m(x.to(DEV), pos.to(DEV))[0, -1].cpu()
            r16, r32 = o16(x, pos)[0, -1], o32(x, pos)[0, -1]
            for acc, r in zip(rows, (got, r16, r32)):
                acc.append(r)
            tok = int(torch.nonzero((r16 / 0.2) == (r16 / 0.2).max())[0])     # pick_token(mode="argmax")
            ids.append(tok)
            x, pos = torch.tensor([[tok]]), torch.tensor([T + s])
        m.reset_cache()
    return tuple(torch.stack(r) for r in rows), ids


def test_full_context_up_to_the_last_cache_slot(tiny):
    """A 120-token prompt decoded to 129 returned tokens: the last forward runs at position block_size - 1 = 127 (the
    last rope row, the last KV slot).  generate/base.py:43-47 allows exactly this and refuses one more."""
    cfg, m, O, o16, o32 = tiny
    assert cfg.block_size == 128
    T, total = 120, 129
    idx = synth_prompts(1, T, cfg.padded_vocab_size, seed=21)[0]
    (got, r16, r32), ids = teacher_forced(tiny, idx, total - T)
    yard, dist = rel_rms(r16, r32), rel_rms(got, r16)
    record_parity("edges.full_context", steps=total - T, rel_rms_hip_vs_oracle=dist, rel_rms_oracle_bf16_vs_fp32=yard,
                  last_step_rel_rms=rel_rms(got[-1], r16[-1]))
    assert dist <= yard, f"full-context logits: {dist:.3e} > oracle's own bf16 distance {yard:.3e}"
    assert rel_rms(got[-1], r16[-1]) <= 1.5 * rel_rms(r16[-1], r32[-1]), "the step at position block_size - 1 is off"
    # free-running greedy ids, as far as the oracle's own arg-max is outside its bf16 noise
    safe = 0
    for s in range(total - T):
        if margin_ulps(r16[s]) < SAFE_MARGIN_ULPS:
            break
        safe += 1
    out = generate(m, idx.to(DEV), total, temperature=0.2, top_k=1).cpu()
    assert out.numel() == total and torch.equal(out[:T], idx)
    assert out[T:T + safe].tolist() == ids[:safe], f"greedy ids differ inside the tie-free prefix ({safe} steps)"
    o16.reset_cache()
    want = O.generate(o16, idx, total, temperature=0.2, top_k=1, mode="argmax")
    assert want[T:T + safe].tolist() == ids[:safe]
    # one token more does not fit the cache: the reference's error, not a silent wrap (generate/base.py:43-47)
    with pytest.raises(NotImplementedError):
        generate(m, idx.to(DEV), total + 1, temperature=0.2, top_k=1)
    with pytest.raises(NotImplementedError):
        O.generate(o16, idx, total + 1, temperature=0.2, top_k=1, mode="argmax")
    # nothing to generate (generate/base.py:42)
    with pytest.raises(AssertionError):
        generate(m, idx.to(DEV), T, temperature=0.2, top_k=1)
    # a prompt filling the whole context still gets its one token
    full = synth_prompts(1, 128, cfg.padded_vocab_size, seed=22)[0]
    o16.reset_cache()
    w = O.generate(o16, full, 129, temperature=0.2, top_k=1, mode="argmax", return_logits=True)
    g = generate(m, full.to(DEV), 129, temperature=0.2, top_k=1).cpu()
    assert g.numel() == 129
    if margin_ulps(w[1][0]) >= SAFE_MARGIN_ULPS:
        assert int(g[-1]) == int(w[0][-1])


def test_one_token_prompt(tiny):
    """T = 1: the prefill IS a single-token forward at position 0 (no key but itself)."""
    cfg, m, O, o16, o32 = tiny
    idx = torch.tensor([7], dtype=torch.int64)
    (got, r16, r32), ids = teacher_forced(tiny, idx, 6)
    yard, dist = rel_rms(r16, r32), rel_rms(got, r16)
    record_parity("edges.one_token_prompt", rel_rms_hip_vs_oracle=dist, rel_rms_oracle_bf16_vs_fp32=yard)
    assert dist <= yard
    safe = 0
    for s in range(6):
        if margin_ulps(r16[s]) < SAFE_MARGIN_ULPS:
            break
        safe += 1
    out = generate(m, idx.to(DEV), 7, temperature=0.2, top_k=1).cpu()
    assert out.numel() == 7 and out[1:1 + safe].tolist() == ids[:safe]


def test_eos_on_the_first_generated_token(tiny):
    """generate/base.py:79-80 with the very first pick: the result is the prompt itself (EOS excluded, Q7); in a batch
    the other sequences go on, and the finished one neither grows nor disturbs them."""
    cfg, m, O, o16, o32 = tiny
    p = synth_prompts(3, 33, cfg.padded_vocab_size, seed=23)
    first = int(generate(m, p[0].to(DEV), 34, temperature=0.2, top_k=1)[-1])
    out = generate(m, p[0].to(DEV), 40, temperature=0.2, top_k=1, eos_id=first).cpu()
    assert torch.equal(out, p[0])
    o16.reset_cache()
    r = O.generate(o16, p[0], 34, temperature=0.2, top_k=1, mode="argmax", return_logits=True)
    if margin_ulps(r[1][0]) >= SAFE_MARGIN_ULPS:
        assert first == int(r[0][-1])
        o16.reset_cache()
        assert torch.equal(O.generate(o16, p[0], 40, temperature=0.2, top_k=1, eos_id=first, mode="argmax"), p[0])
    alone = [generate(m, q.to(DEV), q.numel() + 7, temperature=0.2, top_k=1, eos_id=first).cpu() for q in p]
    both = [o.cpu() for o in generate_batch(m, [q.to(DEV) for q in p], 7, temperature=0.2, top_k=1, eos_id=first)]
    assert torch.equal(both[0], p[0]) and all(torch.equal(a, b) for a, b in zip(alone, both))


def test_ragged_batch_of_extremes(tiny):
    """One joint prefill + decode over prompt lengths 1 .. 120 (a length-1 row, both sides of the 32- and 64-key tile edges,
    a row that ends on the last cache slot): every row equals its own run, ids and final state."""
    cfg, m, O, o16, o32 = tiny
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 97, 120]
    new = 8                                            # 120 + 8 = 128 = block_size: the longest row fills the cache
    ps = [synth_prompts(1, n, cfg.padded_vocab_size, seed=100 + n)[0] for n in lens]
    alone = [generate(m, q.to(DEV), q.numel() + new, temperature=0.2, top_k=1).cpu() for q in ps]
    for pb in (32, 3):                                 # one packed prefill; chunked prefill of three prompts at a time
        joint = [o.cpu() for o in generate_batch(m, [q.to(DEV) for q in ps], new, temperature=0.2, top_k=1, prefill_batch=pb)]
        for n, a, b in zip(lens, alone, joint):
            assert torch.equal(a, b), f"prompt of {n} tokens: joint run (prefill_batch {pb}) differs from the alone run"
    # the oracle's ids for the shortest and the longest row, inside their tie-free prefixes
    for q, a in ((ps[0], alone[0]), (ps[-1], alone[-1])):
        o16.reset_cache()
        w, lg = O.generate(o16, q, q.numel() + new, temperature=0.2, top_k=1, mode="argmax", return_logits=True)
        safe = 0
        for s in range(new):
            if margin_ulps(lg[s]) < SAFE_MARGIN_ULPS:
                break
            safe += 1
        T = q.numel()
        record_parity(f"edges.ragged_len{T}", generated=new, tie_free_prefix=safe)
        assert torch.equal(a[:T + safe], w[:T + safe])


@pytest.mark.parametrize("shape", ["tiny", "tiny-fp8", "tinyllama-width"])
def test_last_block_on_the_last_rows_only_changes_nothing(shape):
    """A prefill asked for the last position's logits (generate's prompt forward, generate/base.py:57-60) runs the LAST
    block's attention output, projection and MLP on each sequence's last row only (csrc/engine.hip g_prune_last_layer; the
    K / V rows of every token still go to the cache).  Against the same call with every row computed (dh_set_tuning 23 = 0):
    logits bit-equal, the whole KV cache bit-equal, the decode that follows identical — for a ragged pack, for a chunk that
    continues a cached prefix, and at a width / row count that takes the 256-tile GEMMs."""
    from dualhyp_amd import _lib
    lib = _lib.load()
    if shape == "tiny":
        cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
        lens = [1, 2, 31, 32, 33, 64, 65, 97, 120]
    elif shape == "tiny-fp8":      # the fp8 serving engine (merged LoRA, e4m3 weights): run_layers_fp8 takes the same short cut
        cfg = Config.from_name("parity-hs128", r=16, alpha=16, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
        lens = [1, 2, 31, 32, 33, 64, 65, 97, 120]
    else:      # TinyLlama's layer shape (d 2048, 32 heads / 4 groups, I 5632), 2 layers, 2 x 700 + 1 rows: the 256-tile kernels
        cfg = Config.from_name("tiny-llama-1.1b-chat", r=16, alpha=16, dropout=0.0, to_query=True, to_key=True, to_value=True,
                               to_projection=True, n_layer=2)
        lens = [700, 1, 700]
    sd = synth_state_dict(cfg, seed=11, norm_jitter=0.25, weight_scale=4.0 if shape.startswith("tiny") and "llama" not in shape else 1.0, device=DEV)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(sd)
    m.eval()
    if shape == "tiny-fp8":
        from dualhyp_amd import quantize_model_fp8
        quantize_model_fp8(m)
    ps = [synth_prompts(1, n, cfg.padded_vocab_size, seed=200 + i)[0].to(DEV) for i, n in enumerate(lens)]
    packed = torch.cat(ps)
    S = 768 if shape == "tinyllama-width" else 128
    G, hs, L = cfg.n_query_groups, cfg.head_size, cfg.n_layer

    def run(prune: int):
        _lib.check(lib.dh_set_tuning(23, prune))
        m.refresh_engine()          # a fresh (zeroed) KV cache for each run: the caches are compared whole
        try:
            eng = m.engine(len(ps), S, int(packed.numel()), exact=True)
            _, last = eng.forward(packed, lens, [0] * len(ps), want_all=False, want_last=True)
            kv = [eng.read(w, l, (len(ps), G, S, hs)).clone() for l in range(L) for w in (1, 2)]
            # a second chunk on top of the cached prefix of the first two sequences (chunked prefill), last logits only
            more = [synth_prompts(1, 5, cfg.padded_vocab_size, seed=300 + i)[0].to(DEV) for i in range(2)]
            _, last2 = eng.forward(torch.cat(more), [5, 5], lens[:2], want_all=False, want_last=True)
            ids = generate_batch(m, ps, 6, temperature=0.2, top_k=1, prefill_batch=4)
            return last.clone(), last2.clone(), kv, [o.cpu() for o in ids]
        finally:
            _lib.check(lib.dh_set_tuning(23, 1))

    full, pruned = run(0), run(1)
    assert torch.equal(full[0], pruned[0]), "last-position logits differ"
    assert torch.equal(full[1], pruned[1]), "last-position logits of the continued chunk differ"
    assert all(torch.equal(a, b) for a, b in zip(full[2], pruned[2])), "KV cache differs"
    assert all(torch.equal(a, b) for a, b in zip(full[3], pruned[3])), "generated ids differ"


def test_eos_early_exit_in_chunks_of_steps(tiny):
    """With an eos_id the decode loop is issued 16 steps at a time and stops once every sequence has finished
    (dualhyp_amd.generate.EOS_CHECK_EVERY).  Against the loop issued in one piece: the same tokens for greedy and for
    sampled decoding (the per-step RNG counter continues across chunks), rows that finish early stay frozen while the
    others go on, and the steps actually run are fewer."""
    import importlib
    G = importlib.import_module("dualhyp_amd.generate")      # (the package binds the name `generate` to the function)
    cfg, m, O, o16, o32 = tiny
    ps = [q.to(DEV) for q in synth_prompts(3, 20, cfg.padded_vocab_size, seed=31)]
    new = 60
    free = [o.cpu() for o in generate_batch(m, ps, new, temperature=0.2, top_k=1)]
    # an EOS that row 0 produces early (its 5th generated token) and the others later or never
    eos = int(free[0][20 + 4])
    def cut(o):
        g = o[20:]
        hit = (g == eos).nonzero().flatten()
        return o[: 20 + int(hit[0])] if hit.numel() else o
    want = [cut(o) for o in free]
    tm = {}
    got = [o.cpu() for o in generate_batch(m, ps, new, temperature=0.2, top_k=1, eos_id=eos, timing=tm)]
    assert all(torch.equal(a, b) for a, b in zip(want, got)), "EOS run differs from the cut free run"
    # tokens until the last row has produced its EOS (the first token comes from the prompt forward), in whole chunks of 16 steps
    need = max((int((o[20:] == eos).nonzero().flatten()[0]) + 1) if bool((o[20:] == eos).any()) else new for o in free)
    expect = new - 1 if need >= new else min(new - 1, max(16, -(-(need - 1) // 16) * 16))
    assert tm["decode_steps"] == expect, f"{tm['decode_steps']} decode steps run, {expect} expected (all rows done after {need} tokens)"
    # sampled decoding: chunked == one piece (RNG step counter continuity); eos chosen so that nothing stops early
    kw = dict(temperature=1.0, top_k=5, seed=77)
    never = cfg.padded_vocab_size - 1
    chunked = [o.cpu() for o in generate_batch(m, ps, new, eos_id=never, **kw)]
    old = G.EOS_CHECK_EVERY
    try:
        G.EOS_CHECK_EVERY = 1 << 20
        whole = [o.cpu() for o in generate_batch(m, ps, new, eos_id=never, **kw)]
    finally:
        G.EOS_CHECK_EVERY = old
    plain = [o.cpu() for o in generate_batch(m, ps, new, **kw)]
    if all((o[20:] != never).all() for o in plain):
        assert all(torch.equal(a, b) for a, b in zip(chunked, whole)) and all(torch.equal(a, b) for a, b in zip(chunked, plain))

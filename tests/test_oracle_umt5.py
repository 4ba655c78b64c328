"""UMT5 text-encoder oracle pinned against the golden vector produced by `transformers.UMT5EncoderModel` itself
(tests/golden/make_golden_umt5.py), plus host-side prompt handling."""
from pathlib import Path
from types import SimpleNamespace

import torch
from safetensors.torch import load_file

from oracle import umt5 as OU

GOLD = Path(__file__).parent / "golden" / "umt5_tiny.safetensors"
TINY = dict(vocab_size=120, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)


def test_oracle_matches_transformers_golden():
    g = load_file(str(GOLD))
    cfg = OU.UMT5Config(**TINY)
    sd = OU.make_weights(cfg, seed=13)
    out = OU.encode(sd, cfg, g["input_ids"], g["attention_mask"])
    valid = g["attention_mask"].bool()
    assert torch.allclose(out[valid], g["out"][valid], atol=2e-5, rtol=1e-5)
    pe = OU.prompt_embeds(sd, cfg, g["input_ids"], g["attention_mask"], 48)
    assert torch.allclose(pe, g["out"], atol=2e-5, rtol=1e-5) and float(pe[1, 9:].abs().max()) == 0


def test_relative_position_buckets_known_values():
    rel = torch.tensor([0, 1, 7, 8, 9, 16, 64, 127, 128, 500, -1, -7, -8, -16, -128, -500])
    b = OU.relative_position_bucket(rel, 32, 128)
    # bidirectional: 16 buckets per sign; exact below 8, log-spaced to 128, clamped to 15 (+16 for keys after the query)
    assert b.tolist() == [0, 17, 23, 24, 24, 26, 30, 31, 31, 31, 1, 7, 8, 10, 15, 15]
    from vist3a_amd.wan.text_encoder import _bucket
    allrel = torch.arange(-511, 512)
    assert torch.equal(_bucket(allrel, 32, 128), OU.relative_position_bucket(allrel, 32, 128))


def test_padding_does_not_leak_into_valid_rows():
    cfg = OU.UMT5Config(**TINY)
    sd = OU.make_weights(cfg, seed=2)
    ids = torch.randint(2, 120, (1, 20), generator=torch.Generator().manual_seed(0))
    mask = torch.zeros(1, 20, dtype=torch.long)
    mask[0, :11] = 1
    a = OU.encode(sd, cfg, ids, mask)[0, :11]
    b = OU.encode(sd, cfg, ids[:, :11], mask[:, :11])[0]   # trimming to the valid tokens is the same function
    assert torch.allclose(a, b, atol=1e-5)
    ids2 = ids.clone()
    ids2[0, 11:] = 7
    assert torch.allclose(OU.encode(sd, cfg, ids2, mask)[0, :11], a, atol=1e-6)


def test_compute_wan_text_embeddings_host_logic():
    """wan_utils.py:25-60 mirror with a stand-in tokenizer/encoder: cleaning, max-length padding, zero rows past seq_len."""
    from vist3a_amd.wan.text_encoder import compute_wan_text_embeddings, prompt_clean
    assert prompt_clean("  a  &amp;amp; b \n\t c ") == "a & b c"
    seen = {}

    def tok(prompts, padding, max_length, truncation, add_special_tokens, return_attention_mask, return_tensors):
        seen["prompts"] = prompts
        assert (padding, truncation, add_special_tokens, return_attention_mask, return_tensors) == ("max_length", True, True, True, "pt")
        ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
        mask = torch.zeros_like(ids)
        for i, p in enumerate(prompts):
            n = min(len(p.split()) + 1, max_length)
            ids[i, :n] = torch.arange(1, n + 1)
            mask[i, :n] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)

    class Enc:
        dtype = torch.float32

        def __call__(self, ids, mask):
            return SimpleNamespace(last_hidden_state=ids.float()[..., None].repeat(1, 1, 4) + 100.0)  # non-zero on padding too

    out = compute_wan_text_embeddings(["one  two three", "x"], Enc(), tok, max_sequence_length=6, device="cpu")
    assert seen["prompts"] == ["one two three", "x"] and out.shape == (2, 6, 4)
    assert out[0, :4, 0].tolist() == [101.0, 102.0, 103.0, 104.0] and float(out[0, 4:].abs().max()) == 0
    assert out[1, :2, 0].tolist() == [101.0, 102.0] and float(out[1, 2:].abs().max()) == 0

"""CPU: the inference oracle (oracle/decode_ref.py: beam search, CTC prefix scores, prefix decoding) against the golden vectors the
imported reference produced (tests/golden/g11_decode.npz)."""
import torch

from conftest import sub
from oracle import cpu_ref as R
from oracle import decode_ref as D

CFG = dict(d_model=128, num_heads=2, dec_layers=2, activation="GELU", emb_scale=False)
V = 23
CASES = dict(a=dict(beam_size=4, min_f2t_ratio=1.5),
             b=dict(beam_size=4, min_f2t_ratio=-8, ctc_weight=0.3, temperature=1.3, length_penalty=1.2, ctc_temperature=1.1),
             c=dict(beam_size=3, min_f2t_ratio=2.0, eos_filtering=True, eos_threshold=1.2),
             d=dict(beam_size=2, min_f2t_ratio=-9, ctc_weight=0.5))


def decode_weights(g):
    """Fixture keys (the reference ARASRDecoder's state_dict) -> the flat names oracle.decode_ref.asr_decode_logits reads."""
    sd = sub(g, "sd.decoder.")
    P = {"embed.weight": sd["embedding.embed.weight"]}
    P.update({k: v for k, v in sd.items() if k.startswith(("decoder.", "postnet."))})
    P["decoder.posenc.posenc"] = R.posenc_table(100, CFG["d_model"])
    return P, sub(g, "sd.ctc_layer.")


def test_prefix_logits(golden):
    g = golden("g11_decode")
    P, _ = decode_weights(g)
    mask = R.make_mask_from_len(g["enc_len"])
    got = D.asr_decode_logits(g["enc"], mask, g["pref_text"], g["pref_text_len"], P, CFG)
    assert torch.allclose(got, g["pref_logits"], atol=2e-5, rtol=1e-5), float((got - g["pref_logits"]).abs().max())


def test_ctc_prefix_scores(golden):
    g = golden("g11_decode")
    sc = D.CtcPrefixScores(g["ctc_x"], g["ctc_lens"], batch_size=2, beam_size=3, blank_index=0, eos_index=8)
    text, state = torch.zeros(6, 0, dtype=torch.long), None
    for step in range(3):
        psi, state = sc.forward_step(text, state)
        assert torch.allclose(psi, g[f"ctc_psi{step}"], atol=1e-4, rtol=1e-5), (step, float((psi - g[f"ctc_psi{step}"]).abs().max()))
        beam_idx, tokens = g[f"ctc_beam_idx{step}"], g[f"ctc_tokens{step}"]
        text = torch.cat([text[beam_idx], tokens[:, None]], dim=1)
        state = sc.permute_mem(state, beam_idx, tokens.view(2, 3) + (torch.arange(3) * 9)[None, :])
        assert torch.allclose(state[0], g[f"ctc_r{step}"], atol=1e-4, rtol=1e-5)
        assert torch.allclose(state[1][:, 0], g[f"ctc_psimem{step}"], atol=1e-4, rtol=1e-5)
    psi, _ = sc.forward_step(text, state)
    assert torch.allclose(psi, g["ctc_psi3"], atol=1e-4, rtol=1e-5)


def test_beam_search(golden):
    g = golden("g11_decode")
    P, Pc = decode_weights(g)
    mask = R.make_mask_from_len(g["enc_len"])
    for name, kw in CASES.items():
        res = D.beam_search(g["enc"], mask, lambda e, m, t, tl: D.asr_decode_logits(e, m, t, tl, P, CFG), V, sos_eos=V - 1,
                            padding_idx=0, ctc_logits_fn=lambda e: R.token_postnet(e, Pc, ""), **kw)
        assert torch.equal(res["hypo_text"], g[f"{name}.hypo_text"]), (name, res["hypo_text"], g[f"{name}.hypo_text"])
        assert torch.equal(res["hypo_text_len"], g[f"{name}.hypo_text_len"]), name
        assert torch.allclose(res["hypo_text_confid"], g[f"{name}.hypo_text_confid"].float(), atol=1e-4), name
        assert torch.allclose(res["feat_token_len_ratio"], g[f"{name}.feat_token_len_ratio"].float(), atol=1e-5), name

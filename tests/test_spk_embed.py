"""Speaker embeddings of the TTS decoder (speechain/module/prenet/spk_embed.py:7-230, decoder/nar_tts.py:42-52, 257-263): the oracle's
restatement against vectors captured from the imported reference (tests/golden/g14_spk.npz: four configurations -- pretrained vectors
concatenated as in the VCTK recipes, both sources added behind their pre-projections with the decoder input combined too, the lookup table
concatenated with a decoder projection of its own, addition without pre-projection), then the HIP drop-in against the same vectors."""
import pytest
import torch

from oracle import cpu_ref as R
from tests.conftest import sub

CASES = dict(pre_cat=dict(d_model=48, spk_emb_dim_pretrained=24, spk_emb_comb="concat"),
             both_add=dict(d_model=48, spk_emb_dim_lookup=16, spk_emb_dim_pretrained=24, spk_num=7, spk_emb_comb="add", dec_comb=True),
             look_cat=dict(d_model=48, spk_emb_dim_lookup=16, spk_num=7, spk_emb_comb="concat", dec_comb=True, encdec_same_proj=False),
             pre_add_same=dict(d_model=48, spk_emb_dim_pretrained=48, spk_emb_comb="add"))


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_speaker_embedding_matches_reference(golden, tag):
    G = sub(golden("g14_spk"), tag + ".")
    conf = CASES[tag]
    P = {k: v.clone().requires_grad_() for k, v in sub(G, "sd.").items()}
    enc = G["enc"].clone().requires_grad_()
    dec = G["dec"].clone().requires_grad_() if "dec" in G else None
    look, pre, enc_out, dec_out = R.speaker_embed(P, "", enc, spk_ids=G.get("spk_ids"), spk_feat=G.get("spk_feat"), dec_input=dec,
                                                  spk_emb_comb=conf["spk_emb_comb"], dec_comb=conf.get("dec_comb", False),
                                                  encdec_same_proj=conf.get("encdec_same_proj", True))
    if "look" in G:
        assert torch.allclose(look, G["look"], atol=1e-6)
    if "pre" in G:
        assert torch.allclose(pre, G["pre"], atol=1e-6)
    assert torch.allclose(enc_out, G["enc_out"], atol=1e-5)
    loss = (enc_out * G["w1"]).sum()
    if dec is not None:
        assert torch.allclose(dec_out, G["dec_out"], atol=1e-5)
        loss = loss + (dec_out * G["w2"]).sum()
    loss.backward()
    for k, v in P.items():
        assert _rel(v.grad, G["grad." + k]) < 1e-5, k
    assert _rel(enc.grad, G["grad_enc"]) < 1e-5


def test_drop_in_has_the_reference_state_dict(golden):
    from speechain_amd.module.prenet.spk_embed import SpeakerEmbedPrenet

    for tag, conf in CASES.items():
        G = sub(sub(golden("g14_spk"), tag + "."), "sd.")
        m = SpeakerEmbedPrenet(**conf)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in G.items()}, tag
        m.load_state_dict(G, strict=True)
    with pytest.raises(AssertionError):
        SpeakerEmbedPrenet(d_model=8)
    with pytest.raises(AssertionError):
        SpeakerEmbedPrenet(d_model=8, spk_emb_dim_pretrained=4, spk_emb_comb="mul")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_speaker_embedding_matches_reference(golden, tag, mode):
    from speechain_amd import functional as Fn
    from speechain_amd.module.prenet.spk_embed import SpeakerEmbedPrenet

    G = sub(golden("g14_spk"), tag + ".")
    Fn.set_compute_dtype(torch.float32 if mode == "f32" else torch.bfloat16)
    tol_o, tol_g = (2e-5, 2e-5) if mode == "f32" else (1e-2, 2e-2)
    try:
        m = SpeakerEmbedPrenet(**CASES[tag])
        m.load_state_dict(sub(G, "sd."), strict=True)
        m = m.cuda().train()
        enc = G["enc"].cuda().requires_grad_()
        dec = G["dec"].cuda().requires_grad_() if "dec" in G else None
        ids = G["spk_ids"].cuda() if "spk_ids" in G else None
        feat = G["spk_feat"].cuda() if "spk_feat" in G else None
        look, pre = m(spk_ids=ids, spk_feat=feat)
        enc_out, dec_out = m.combine_spk_feat(spk_feat=pre, spk_feat_lookup=look, enc_output=enc, dec_input=dec)
        if "look" in G:
            assert _rel(look.float().cpu(), G["look"]) < tol_o
        if "pre" in G:
            assert _rel(pre.float().cpu(), G["pre"]) < tol_o
        assert _rel(enc_out.float().cpu(), G["enc_out"]) < tol_o, _rel(enc_out.float().cpu(), G["enc_out"])
        loss = (enc_out.float() * G["w1"].cuda()).sum()
        if dec is not None:
            assert _rel(dec_out.float().cpu(), G["dec_out"]) < tol_o
            loss = loss + (dec_out.float() * G["w2"].cuda()).sum()
        loss.backward()
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            assert _rel(p.grad.float().cpu(), G["grad." + k]) < tol_g, (k, _rel(p.grad.float().cpu(), G["grad." + k]))
        assert _rel(enc.grad.float().cpu(), G["grad_enc"]) < tol_g
        if dec is not None:
            assert _rel(dec.grad.float().cpu(), G["grad_dec"]) < tol_g
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()


@pytest.mark.gpu
def test_fastspeech2_decoder_applies_the_speaker_vectors_first():
    """decoder/nar_tts.py:257-263: with ``spk_emb`` the decoder equals the decoder without it fed the encoded text that the
    (fixture-pinned) SpeakerEmbedPrenet has already combined with the speaker vectors."""
    import copy
    import importlib.util
    import os

    from speechain_amd import functional as Fn
    from speechain_amd.module.decoder.nar_tts import FastSpeech2Decoder

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_conf", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    conf = bench.tts_recipe_conf(0.0, 0.0)
    kw = dict(decoder=conf["decoder"], postnet=conf["dec_postnet"], pitch_predictor=conf["pitch_predictor"], energy_predictor=conf["energy_predictor"],
              duration_predictor=conf["duration_predictor"], feat_frontend=conf["feat_frontend"])
    Fn.set_compute_dtype(torch.float32)
    try:
        torch.manual_seed(0)
        with_spk = FastSpeech2Decoder(input_size=384, spk_emb=dict(spk_emb_dim_pretrained=192, spk_emb_comb="concat"), **copy.deepcopy(kw)).cuda().eval()
        plain = FastSpeech2Decoder(input_size=384, **copy.deepcopy(kw)).cuda().eval()
        plain.load_state_dict({k: v for k, v in with_spk.state_dict().items() if not k.startswith("spk_emb.")}, strict=False)
        g = torch.Generator().manual_seed(3)
        enc = torch.randn(2, 12, 384, generator=g).cuda()
        mask = torch.ones(2, 1, 12, dtype=torch.bool).cuda()
        spk = torch.randn(2, 192, generator=g).cuda()
        with torch.no_grad():
            a = with_spk(enc_text=enc, enc_text_mask=mask, spk_feat=spk, min_frame_num=0)
            look, pre = with_spk.spk_emb(spk_feat=spk)
            enc2, _ = with_spk.spk_emb.combine_spk_feat(spk_feat=pre, spk_feat_lookup=look, enc_output=enc)
            b = plain(enc_text=enc2, enc_text_mask=mask, min_frame_num=0)
        assert torch.equal(a[2], b[2]) and torch.allclose(a[0].float(), b[0].float(), atol=1e-5) and torch.allclose(a[1].float(), b[1].float(), atol=1e-5)
        c = plain(enc_text=enc, enc_text_mask=mask, min_frame_num=0)[0].float()  # (the speaker vectors matter: other durations or other frames)
        assert c.shape != a[0].shape or not torch.allclose(a[0].float(), c, atol=1e-3)
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()

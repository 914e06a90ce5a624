"""CPU: the oracle (oracle/cpu_ref.py) against the golden vectors produced by the imported reference."""
import math

import pytest
import torch

from conftest import sub
from oracle import cpu_ref as R


def close(a, b, atol=1e-5, rtol=1e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), float((a - b).abs().max())


ASR = dict(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)
TTS = dict(sr=22050, hop_length=256, win_length=1024, n_fft=1024, n_mels=80, fmin=0, fmax=8000, mag_spec=True,
           return_energy=True, clamp=1e-5, log_base=None)
ODD = dict(sr=16000, hop_length=128, win_length=400, n_fft=512, n_mels=40, mel_scale="htk", mel_norm=False, log_base=None)


def test_frontend_asr(golden):
    g = golden("g1_frontend")
    feat, flen = R.speech2mel(g["wave"], g["wave_len"], **ASR)
    assert torch.equal(flen, g["asr_len"])
    close(feat, g["asr_logmel"], atol=2e-4)  # fft-order rounding, amplified by log10 near the floor
    close(R.mel_matrix(400, 80), g["asr_melw"], atol=0, rtol=0)
    # explicit framing/rfft definition agrees as well (independent of torch.stft)
    feat2, _ = R.speech2mel(g["wave"], g["wave_len"], explicit=True, **ASR)
    close(feat2, g["asr_logmel"], atol=2e-4)
    lin, _ = R.speech2linear(g["wave"], g["wave_len"], sr=16000, preemph=0.97, hop_length=0.010, win_length=0.025)
    close(lin[:2, :40], g["asr_power"], atol=1e-6, rtol=1e-4)


def test_frontend_tts_and_odd(golden):
    g = golden("g1_frontend")
    feat, flen, en, _ = R.speech2mel(g["wave"][:3], g["wave_len"][:3], **TTS)
    assert torch.equal(flen, g["tts_len"])
    close(feat, g["tts_logmel"], atol=2e-4)
    close(en, g["tts_energy"], atol=1e-5, rtol=1e-4)
    feat, flen = R.speech2mel(g["wave"][:2], g["wave_len"][:2], **ODD)
    assert torch.equal(flen, g["odd_len"])
    close(feat, g["odd_logmel"], atol=2e-4)
    close(R.mel_matrix(512, 40, mel_scale="htk", mel_norm=False), g["odd_melw"], atol=0, rtol=0)


def test_wave_len_not_mutated(golden):
    g = golden("g1_frontend")
    wl = g["wave_len"].clone()
    R.speech2mel(g["wave"], wl, **ASR)
    assert torch.equal(wl, g["wave_len"])


SPECAUG_CASES = dict(
    recipe=dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5),
    default=dict(),
    nowarp_mean=dict(time_warp=False, feat_norm=False, freq_mask_width=[2, 9], time_mask_width=[3, 12]),
    warponly=dict(freq_mask=False, time_mask=False),
    bilinear=dict(time_warp_mode="bilinear", freq_mask_width=15, freq_mask_num=2, time_mask_width=0.05, time_mask_num=2),
)


@pytest.mark.parametrize("name", list(SPECAUG_CASES))
def test_specaug_draws_bit_exact_and_output(golden, name):
    g = golden("g2_specaug")
    conf = dict(SPECAUG_CASES[name])
    feat_norm = conf.pop("feat_norm", True)
    mode = conf.pop("time_warp_mode", "bicubic")
    torch.manual_seed(int(g[f"{name}_seed"]))
    d = R.specaug_draw(g["feat_len"], 3, 80, **conf)
    drawn = [x for x in (d["warp_center"], d["warp_pos"], d["freq_len"], d["freq_pos"], d["time_len"], d["time_pos"])
             if x is not None]
    assert len(drawn) == int(g[f"{name}_ndraw"])
    for j, x in enumerate(drawn):
        ref = g[f"{name}_draw{j}"]
        assert torch.equal(torch.as_tensor(x).reshape(ref.shape), ref), (name, j)
    for explicit in ((False, True) if mode == "bicubic" else (False,)):  # (the explicit resampler restates the cubic kernel only)
        out = R.specaug_apply(g["feat"], g["feat_len"], d, time_warp_mode=mode, feat_norm=feat_norm, explicit_warp=explicit)
        close(out, g[f"{name}_out"], atol=1e-4 if explicit else 1e-6)


def test_posenc(golden):
    g = golden("g3_posenc")
    rows = [0, 1, 2, 17, 150, 299]
    for typ in ("mix", "sep"):
        close(R.posenc_table(300, 64, typ)[0, rows], g[f"{typ}_rows"], atol=0, rtol=0)
    P = {"p." + k: v for k, v in sub(g, "full_sd.").items()}
    P["p.posenc"] = R.posenc_table(300, 64)
    close(R.posenc_forward(g["x"], P, "p.", 64, emb_scale=True), g["full_out"], atol=1e-5)
    close(R.posenc_forward(g["x"], {"p.posenc": P["p.posenc"]}, "p.", 64), g["plain_out"], atol=1e-6)


def _weights(x):
    return torch.linspace(-1, 1, x.numel()).reshape(x.shape)


def test_mha_fwd_bwd(golden):
    g = golden("g4_layers")
    for tag, by_head in (("mha_dmodel", False), ("mha_byhead", True)):
        q, kv = g[f"{tag}.q"].requires_grad_(), g[f"{tag}.kv"].requires_grad_()
        P = {k: v.clone().requires_grad_() for k, v in sub(g, f"{tag}.sd.").items()}
        y, p = R.mha(kv, kv, q, g[f"{tag}.mask"], P, "", 4, by_head)
        close(y, g[f"{tag}.out"], atol=1e-5)
        close(p, g[f"{tag}.probs"], atol=1e-6)
        gq, gkv, gw = torch.autograd.grad((y * _weights(y)).sum(), [q, kv, P["q_layer.weight"]])
        close(gq, g[f"{tag}.gq"], atol=1e-4)
        close(gkv, g[f"{tag}.gkv"], atol=1e-4)
        close(gw, g[f"{tag}.gw_q"], atol=1e-3, rtol=1e-4)
    x = g["mha192.x"]
    y, p = R.mha(x, x, x, g["mha192.mask"], sub(g, "mha192.sd."), "", 2)
    close(y, g["mha192.out"], atol=1e-5)
    close(p, g["mha192.probs"], atol=1e-6)


@pytest.mark.parametrize("tag,act,typ", [("ffn_relu", "ReLU", "linear"), ("ffn_gelu", "GELU", "linear"),
                                         ("ffn_conv", "ReLU", "conv")])
def test_ffn(golden, tag, act, typ):
    g = golden("g4_layers")
    x = g[f"{tag}.x"].requires_grad_()
    P = {k: v.clone().requires_grad_() for k, v in sub(g, f"{tag}.sd.").items()}
    y = R.ffn(x, P, "", act, typ)
    close(y, g[f"{tag}.out"], atol=1e-5)
    w = P["in_layer.weight"] if typ == "linear" else P["in_layer.conv_lyr.weight"]
    gx, gw = torch.autograd.grad((y * _weights(y)).sum(), [x, w])
    close(gx, g[f"{tag}.gx"], atol=1e-4)
    close(gw, g[f"{tag}.gw_in"], atol=1e-3, rtol=1e-4)


def test_encoder_decoder_layers(golden):
    g = golden("g4_layers")
    for tag, first in (("enc_pre", True), ("enc_post", False)):
        x = g[f"{tag}.x"].requires_grad_()
        P = {k: v.clone().requires_grad_() for k, v in sub(g, f"{tag}.sd.").items()}
        y, p = R.encoder_layer(x, g[f"{tag}.mask"], P, "", 4, activation="GELU", layernorm_first=first)
        close(y, g[f"{tag}.out"], atol=2e-5)
        close(p, g[f"{tag}.probs"], atol=1e-6)
        gx, gln, gb = torch.autograd.grad((y * _weights(y)).sum(),
                                          [x, P["att_layernorm.weight"], P["multihead_att.output_layer.bias"]])
        close(gx, g[f"{tag}.gx"], atol=2e-4)
        close(gln, g[f"{tag}.g_attln_w"], atol=1e-3, rtol=1e-4)
        close(gb, g[f"{tag}.g_outb"], atol=1e-3, rtol=1e-4)
    tgt, src = g["dec.tgt"].requires_grad_(), g["dec.src"].requires_grad_()
    y, sp, ep = R.decoder_layer(tgt, src, g["dec.tmask"], g["dec.smask"], sub(g, "dec.sd."), "", 4,
                                scale_dp_by_head=True, activation="GELU")
    close(y, g["dec.out"], atol=2e-5)
    close(sp, g["dec.self_probs"], atol=1e-6)
    close(ep, g["dec.encdec_probs"], atol=1e-6)
    gt, gs = torch.autograd.grad((y * _weights(y)).sum(), [tgt, src])
    close(gt, g["dec.gtgt"], atol=2e-4)
    close(gs, g["dec.gsrc"], atol=2e-4)


def test_stacks(golden):
    g = golden("g5_stacks")
    P = sub(g, "enc.sd.")
    P["posenc.posenc"] = R.posenc_table(200, 64)
    y, m, att, hid = R.transformer_encoder(g["enc.x"], g["enc.mask"], P, "", num_layers=2, num_heads=4, d_model=64,
                                           activation="GELU")
    close(y, g["enc.out"], atol=5e-5)
    close(att[0], g["enc.att0"], atol=1e-6)
    close(att[1], g["enc.att1"], atol=1e-6)
    close(hid[0], g["enc.hid0"], atol=5e-5)
    close(hid[1], g["enc.hid1"], atol=5e-5)
    P2 = sub(g, "enc2.sd.")
    P2["posenc.posenc"] = R.posenc_table(200, 48)
    y2, m2, att2, _ = R.transformer_encoder(g["enc2.x"], g["enc2.mask"], P2, "", num_layers=2, num_heads=2, d_model=48,
                                            fdfwd_type="conv", scale_dp_by_head=True, uni_direction=True,
                                            layernorm_first=False)
    close(y2, g["enc2.out"], atol=5e-5)
    assert torch.equal(m2, g["enc2.mask_out"])
    close(att2[1], g["enc2.att1"], atol=1e-6)
    Pd = sub(g, "dec.sd.")
    Pd["posenc.posenc"] = R.posenc_table(200, 64)
    yd, sa, ea, hd = R.transformer_decoder(g["dec.tgt"], g["enc.out"], g["dec.tmask"], g["enc.mask"], Pd, "",
                                           num_layers=2, num_heads=4, d_model=64, emb_scale=False, activation="GELU")
    close(yd, g["dec.out"], atol=5e-5)
    close(sa[1], g["dec.self_att1"], atol=1e-6)
    close(ea[1], g["dec.encdec_att1"], atol=1e-6)
    close(hd[0], g["dec.hid0"], atol=5e-5)


def test_losses(golden):
    g = golden("g6_losses")
    for tag, conf in (("ce0", dict()), ("ce02", dict(label_smoothing=0.2)),
                      ("ce02n", dict(label_smoothing=0.2, length_normalized=True))):
        lg = g["ce_logits"].clone().requires_grad_()
        loss = R.cross_entropy(lg, g["text"], g["text_len"], **conf)
        close(loss, g[f"{tag}.loss"], atol=1e-5)
        close(torch.autograd.grad(loss, lg)[0], g[f"{tag}.grad"], atol=1e-6)
    lg = g["ctc_logits"].clone().requires_grad_()
    loss = R.ctc_loss(lg, g["enc_len"], g["ctc_text"], g["text_len"])
    close(loss, g["ctc_loss"], atol=1e-5)
    close(torch.autograd.grad(loss, lg)[0], g["ctc_grad"], atol=1e-6)
    # independent alpha recursion pins the per-utterance values (infeasible alignment -> inf -> zeroed)
    lp = g["ctc_logits"].double().log_softmax(-1)
    for i in range(3):
        tgt = g["ctc_text"][i, 1: int(g["text_len"][i]) - 1].tolist()
        nll = R.ctc_nll_explicit(lp[i, : int(g["enc_len"][i])], tgt)
        ref = float(g["ctc_per_utt"][i])
        assert (math.isinf(nll) and ref == 0.0) or abs(nll - ref) < 1e-4 * max(1.0, abs(ref)), (i, nll, ref)


def test_harness(golden):
    g = golden("g7_harness")
    P = sub(g, "sd.")
    P["encoder.posenc.posenc"] = R.posenc_table(200, 64)
    P["decoder.posenc.posenc"] = R.posenc_table(200, 64)
    for k in ("encoder.trfm_layers.0.multihead_att.q_layer.weight", "embed.weight",
              "decoder.trfm_layers.1.encdec_att.k_layer.bias", "ctc_layer.linear.weight"):
        P[k] = P[k].clone().requires_grad_()
    cfg = dict(enc_layers=2, dec_layers=2, num_heads=4, d_model=64, activation="GELU", label_smoothing=0.2, ctc_weight=0.3)
    o = R.asr_forward_loss(g["feat"], g["feat_len"], g["text"], g["text_len"], P, cfg)
    close(o["enc"], g["enc"], atol=5e-5)
    close(o["logits"], g["logits"], atol=1e-4)
    close(o["ctc_logits"], g["ctc_logits"], atol=1e-4)
    for k in ("ce", "ctc", "loss"):
        close(o[k], g[k], atol=1e-4)
    gs = torch.autograd.grad(o["loss"], [P["encoder.trfm_layers.0.multihead_att.q_layer.weight"], P["embed.weight"],
                                         P["decoder.trfm_layers.1.encdec_att.k_layer.bias"], P["ctc_layer.linear.weight"]])
    for a, k in zip(gs, ("g_enc0_q", "g_embed", "g_dec1_encdec_kb", "g_ctc_w")):
        close(a, g[k], atol=1e-4, rtol=1e-3)


# ---- g8: FeatureNormalization (row f1) -------------------------------------------------------------------------------------
def _fn_batches(g):
    return [(g[f"feat{j}"].clone(), g[f"len{j}"].clone()) for j in range(5)]


def _fn_close(a, b, what):
    assert a.shape == b.shape, what
    assert torch.allclose(a, b, atol=1e-6, rtol=1e-5), (what, float((a - b).abs().max()))


def _fn_bufs(g, tag, m):
    for k, v in m.buffers.items():
        ref = g[f"{tag}.buf.{k}"]
        if v.dtype == torch.int64:
            assert torch.equal(v, ref.to(torch.int64)), (tag, k)
        else:
            _fn_close(v, ref, f"{tag}.{k}")
    assert {k for k in g.keys() if k.startswith(tag + ".buf.")} == {f"{tag}.buf.{k}" for k in m.buffers}


def test_feature_normalization_oracle_vs_reference(golden):
    g = golden("g8_featnorm")
    bt = _fn_batches(g)
    m = R.FeatureNormalization("utterance")
    _fn_close(m(*bt[0])[0], g["utt.y"], "utterance")
    m = R.FeatureNormalization("batch")
    _fn_close(m(*bt[0])[0], g["batch.train.y"], "batch train")
    m.training = False
    _fn_close(m(*bt[1])[0], g["batch.eval.y"], "batch eval")
    m = R.FeatureNormalization()  # global: the recipes' `normalize: True`
    for step, (j, ep) in enumerate([(0, 1), (1, 1), (2, 2), (3, 5)]):
        _fn_close(m(*bt[j], epoch=ep)[0], g[f"global.s{step}.y"], f"global step {step}")
        _fn_bufs(g, f"global.s{step}", m)
    m.training = False
    y = m(*bt[4])[0]
    _fn_close(y, g["global.eval.y"], "global eval")
    _fn_close(m.recover(y), g["global.recover"], "recover")
    assert sorted(m.buffers) == sorted(str(k) for k in g["global.sd_keys"])
    m = R.FeatureNormalization()
    for step in range(2):
        _fn_close(m(*bt[step])[0], g[f"global_noepoch.s{step}.y"], f"no-epoch step {step}")
        _fn_bufs(g, f"global_noepoch.s{step}", m)
    _fn_close(R.FeatureNormalization(std_norm=False)(*bt[0], epoch=1)[0], g["meanonly.y"], "mean only")
    _fn_close(R.FeatureNormalization("utterance", mean_norm=False, clamp=0.5)(*bt[0])[0], g["stdonly.y"], "std only")
    _fn_close(R.FeatureNormalization("utterance")(bt[0][0][:, :, 0], bt[0][1])[0], g["twod.y"], "2-D input")
    m = R.FeatureNormalization("group")
    _fn_close(m(*bt[0], group_ids=7, epoch=1)[0], g["group_int.s0.y"], "group int 0")
    _fn_close(m(*bt[1], group_ids=7, epoch=1)[0], g["group_int.s1.y"], "group int 1")
    _fn_bufs(g, "group_int.s1", m)
    m = R.FeatureNormalization("group")
    _fn_close(m(*bt[0], group_ids=g["group_ids0"], epoch=1)[0], g["group_t.s0.y"], "group tensor 0")
    _fn_close(m(*bt[1], group_ids=g["group_ids1"], epoch=1)[0], g["group_t.s1.y"], "group tensor 1")
    _fn_bufs(g, "group_t.s1", m)
    m.training = False
    _fn_close(m(*bt[2], group_ids=g["group_ids2"])[0], g["group_t.eval.y"], "group tensor eval (unseen id -> average)")
    with pytest.raises(ValueError):
        R.FeatureNormalization("global", mean_norm=False)


# ---- g9: Conv2dPrenet (row f2) ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["bn", "relu", "asym", "zc", "narrow"])
def test_conv2d_prenet_oracle_vs_reference(golden, tag):
    from prenet_cases import CASES, state, weighted_sum
    g = golden("g9_prenet")
    P = state(g, tag)
    x, lens = g[f"{tag}.x"], g[f"{tag}.len"]
    y, ylen, stats = R.conv2d_prenet(x, lens.clone(), P, "", **CASES[tag], training=True)
    assert y.shape == g[f"{tag}.out"].shape and y.size(-1) == int(g[f"{tag}.output_size"])
    assert torch.allclose(y, g[f"{tag}.out"], atol=1e-5, rtol=1e-5), float((y - g[f"{tag}.out"]).abs().max())
    assert torch.equal(ylen, g[f"{tag}.out_len"])
    names = [k for k, v in P.items() if v.requires_grad]
    grads = torch.autograd.grad(weighted_sum(y), [P[k] for k in names])
    assert len(names) == len([k for k in g if k.startswith(f"{tag}.grad.")])
    for k, a in zip(names, grads):
        b = g[f"{tag}.grad.{k}"]
        assert torch.allclose(a, b, atol=1e-4 * max(1.0, float(b.abs().max())), rtol=1e-4), (k, float((a - b).abs().max()))
    for k, v in stats.items():
        assert torch.allclose(v, g[f"{tag}.after.{k}"], atol=1e-6, rtol=1e-5), k
    P2 = {k: v.detach() for k, v in P.items()}
    P2.update(stats)
    ye, _, _ = R.conv2d_prenet(x, lens.clone(), P2, "", **CASES[tag], training=False)
    assert torch.allclose(ye, g[f"{tag}.out_eval"], atol=1e-5, rtol=1e-5)


def test_conv2d_prenet_oracle_length_check():
    """conv_padding > 0 with stride 2 trips the reference's own length check (conv2d.py:266-271; observed on the imported module)."""
    P = {"conv.0.weight": torch.zeros(4, 1, 3, 3), "conv.0.bias": torch.zeros(4)}
    with pytest.raises(RuntimeError, match="feature lengths"):
        R.conv2d_prenet(torch.zeros(2, 37, 20), torch.tensor([37, 30]), P, "", conv_dims=[4], conv_padding=1, lnr_dims=None)


# ---- G10 FastSpeech2 (BASELINE configs[4]) ----------------------------------------------------------------------

TTS_SMALL_CFG = dict(d=32, heads=2, layers=2, dur_layers=2, pitch_layers=3, energy_layers=2, post_layers=3)


def tts_state(g, prefix="sd0."):
    P = {k[len(prefix):]: v.clone() for k, v in g.items() if k.startswith(prefix)}
    for key, d in (("encoder.encoder.posenc.posenc", 32), ("decoder.decoder.posenc.posenc", 32)):
        P[key] = R.posenc_table(400, d)
    return P


def test_fastspeech2_oracle_matches_reference(golden):
    """oracle.fastspeech2_forward_loss against the imported reference (TTSEncoder + FastSpeech2Decoder + LeastError terms, two training
    steps with a weight update in between): every output, the five loss terms, fifteen gradients spread over all sub-modules, and the
    buffers after the second step (BatchNorm running statistics, global feature / pitch / energy statistics)."""
    g = golden("g10_tts")
    P = tts_state(g)
    norms = {k: R.FeatureNormalization() for k in ("feat", "pitch", "energy")}
    sel = {"g_emb": "encoder.embedding.embed.weight", "g_enc0_ffn_in": "encoder.encoder.trfm_layers.0.feed_forward.in_layer.conv_lyr.weight",
           "g_dur_conv0": "decoder.duration_predictor.conv.0.conv_lyr.weight", "g_dur_ln0": "decoder.duration_predictor.conv.2.weight",
           "g_dur_lin": "decoder.duration_predictor.linear.weight", "g_pitch_emb_w": "decoder.pitch_predictor.conv_emb.0.conv_lyr.weight",
           "g_pitch_emb_b": "decoder.pitch_predictor.conv_emb.0.conv_lyr.bias", "g_energy_conv1": "decoder.energy_predictor.conv.4.conv_lyr.weight",
           "g_dec1_q": "decoder.decoder.trfm_layers.1.multihead_att.q_layer.weight", "g_dec_alpha": "decoder.decoder.posenc.alpha",
           "g_feat_pred": "decoder.feat_pred.weight", "g_post0": "decoder.postnet.conv.0.conv_lyr.weight",
           "g_post_bn1_w": "decoder.postnet.conv.1.weight", "g_post_last": "decoder.postnet.conv.8.conv_lyr.weight",
           "g_post_bn_last_b": "decoder.postnet.conv.9.bias"}
    for k in sel.values():
        P[k] = P[k].clone().requires_grad_()
    for step in range(2):
        t = f"s{step}."
        out = R.fastspeech2_forward_loss(g[t + "text"], g["text_len"], g[t + "dur"], g[t + "wave"], g["wave_len"], g[t + "pitch"], g["pitch_len"],
                                         P, TTS_SMALL_CFG, norms, epoch=1)
        assert torch.equal(out["tgt_feat_len"], g[t + "tgt_feat_len"]) and torch.equal(out["used_dur"], g[t + "used_dur"])
        for k in ("enc_text", "before", "after", "tgt_feat", "pred_pitch", "tgt_pitch", "pred_energy", "tgt_energy", "pred_dur"):
            assert torch.allclose(out[k], g[t + k], atol=2e-4, rtol=1e-4), (step, k, float((out[k] - g[t + k]).abs().max()))
        for k in ("feat_before", "feat_after", "pitch", "energy", "duration"):
            assert abs(float(out["loss_" + k]) - float(g[t + "loss_" + k])) < 1e-5 * max(1.0, abs(float(g[t + "loss_" + k]))), (step, k)
        grads = torch.autograd.grad(out["loss"], [P[k] for k in sel.values()])
        for (name, key), gr in zip(sel.items(), grads):
            ref = g[t + name]
            assert torch.allclose(gr, ref, atol=2e-5 + 1e-4 * float(ref.abs().max()), rtol=1e-3), (step, name, float((gr - ref).abs().max()))
        for k, v in out["bn_stats"].items():  # running statistics carry over to the next step
            P[k] = v
        if step == 0:
            with torch.no_grad():
                for key, gr in zip(sel.values(), grads):
                    P[key] -= 0.05 * gr
    for k, v in out["bn_stats"].items():
        assert torch.allclose(v, g["buf1." + k], atol=1e-5, rtol=1e-4), k
    for tag, name in (("feat", "decoder.feat_normalize"), ("pitch", "decoder.pitch_normalize"), ("energy", "decoder.energy_normalize")):
        for bk, bv in norms[tag].buffers.items():
            ref = g[f"buf1.{name}.{bk}"]
            assert torch.allclose(bv.float().reshape(-1), ref.float().reshape(-1), atol=1e-5, rtol=1e-4), (name, bk)


def test_average_scalar_by_duration_oracle(golden):
    g = golden("g10_tts")
    assert torch.allclose(R.average_scalar_by_duration(g["avg_frame"], g["avg_dur"]), g["avg_out"], atol=1e-6)


def test_recover_oracle_matches_reference(golden):
    """Row f4, recover(): the mel -> linear least-squares inversion and the post-processing around Griffin-Lim against the imported
    reference (g12_recover; the iteration itself lives in the absent torchaudio and is restated from its published algorithm)."""
    g = golden("g12_recover")
    close(R.linear2mel_recover(g["mel"], g["mel_len"], g["melw"]), g["lin"], atol=1e-4, rtol=1e-3)
    close(R.linear2mel_recover(g["mel"], g["mel_len"], g["melw"], log_base=None), g["lin_ln"], atol=1e-4, rtol=1e-3)
    wav, wav_len = R.speech2linear_recover_post(g["gl_wav"], g["feat_len"], 16, 0.97, 20)
    assert torch.equal(wav_len, g["wav_len"])
    close(wav, g["wav"], atol=1e-6)


def test_griffin_lim_oracle_reduces_the_spectral_error():
    """The restated iteration (torchaudio's published algorithm) does what Griffin-Lim must: the magnitude spectrogram of its output
    approaches the target with the iterations (a property; the third-party code itself cannot be run here)."""
    gen = torch.Generator().manual_seed(2)
    n_fft, hop = 64, 16
    win = torch.hann_window(n_fft)
    wave = torch.randn(2, 16 * 30, generator=gen).cumsum(-1) * 0.02
    mag = torch.stft(wave, n_fft, hop, n_fft, win, return_complex=True).abs().transpose(1, 2)
    init = torch.rand(2, mag.size(1), 33, generator=gen) + 1j * torch.rand(2, mag.size(1), 33, generator=gen)
    errs = []
    for n in (0, 4, 32):
        y = R.griffin_lim(mag, win, n_fft, hop, n_fft, n, 0.99, init)
        m2 = torch.stft(y, n_fft, hop, n_fft, win, return_complex=True).abs().transpose(1, 2)
        errs.append(float((m2 - mag).norm() / mag.norm()))
    assert errs[2] < errs[1] < errs[0] and errs[2] < 0.25, errs


def test_cross_entropy_options_oracle(golden):
    """Token re-weighting and confidence thresholding of the oracle's cross entropy against the imported reference (g13_ce_options)."""
    g = golden("g13_ce_options")
    cases = dict(w=dict(label_smoothing=0.1, token_weights=g["weights"]), tok=dict(confid_threshold=0.05, confid_level="token", length_normalized=True),
                 sent=dict(confid_threshold=0.02, confid_level="sentence"), tok_ls=dict(confid_threshold=0.08, confid_level="token", label_smoothing=0.2))
    for tag, conf in cases.items():
        lg = g["logits"].clone().requires_grad_()
        loss = R.cross_entropy(lg, g["text"], g["text_len"], **conf)
        close(loss.detach(), g[f"{tag}.loss"], atol=1e-5)
        close(torch.autograd.grad(loss, lg)[0], g[f"{tag}.grad"], atol=1e-6, rtol=1e-4)

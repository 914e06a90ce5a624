"""BASELINE.json configs[0]: the mini-LibriSpeech "transformer-small" plumbing case -- d = 144, 4 heads (head size 36), FFN 1024,
12 encoder / 6 decoder layers, bpe1k, batch 4 (sizes of recipes/asr/librispeech/train-clean-5/exp_cfg/5-bpe1k_conformer-small_
lr2e-3a.yaml:41-46 with the transformer encoder).

* CPU (`-m "not gpu"`): the oracle runs the whole chain from waveforms -- frontend, global mean/std normalisation, SpecAugment, conv
  prenet, encoder, decoder, heads, losses, backward -- and gives finite, reproducible values.
* GPU: the HIP path against that oracle on the same seeded inputs, in the exact-f32 parity mode and in bf16.  Head size 36 takes the
  unfused attention path and odd contraction lengths through the GEMM kernels, which no other fixture does."""
import pytest
import torch

from oracle import cpu_ref as R

CFG = dict(d_model=144, num_heads=4, enc_layers=12, dec_layers=6, fdfwd_dim=1024, vocab_size=1000)
FRONTEND = dict(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)
SPECAUG = dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5)


def _batch():
    gen = torch.Generator().manual_seed(144)
    B, L, V = 4, 12, CFG["vocab_size"]
    wave = torch.randn(B, 40000, generator=gen) * 0.1
    lens = torch.tensor([40000, 33000, 27500, 20000])
    for i in range(B):
        wave[i, lens[i]:] = 0
    text = torch.randint(1, V - 2, (B, L), generator=gen)
    text_len = torch.tensor([12, 10, 9, 7])
    for i in range(B):
        text[i, 0] = V - 1
        text[i, text_len[i] - 1] = V - 1
        text[i, text_len[i]:] = 0
    return wave, lens, text, text_len


PRENET = dict(conv_dims=[32, 32], conv_kernel=3, conv_stride=2, conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=CFG["d_model"])


def _model_and_prenet():
    from speechain_amd.harness import ASRTransformer
    from speechain_amd.module.prenet.conv2d import Conv2dPrenet

    torch.manual_seed(5)
    prenet = Conv2dPrenet(input_size=80, **PRENET)  # on the CPU it only holds the parameters: the oracle does the arithmetic
    model = ASRTransformer(**CFG, dropout=0.0, activation="GELU", posenc_maxlen=400, label_smoothing=0.2, ctc_weight=0.3)
    for m in (prenet, model):
        for p in m.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)
    return prenet, model


def _oracle_params(model):
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("posenc")) for k, v in model.state_dict().items()}
    P["embed.weight"] = P.pop("embed.embed.weight")
    return P


def _oracle_step(prenet, model, wave, lens, text, text_len, draws=None):
    """frontend -> normalisation -> SpecAugment -> prenet -> named path -> backward, all on the CPU oracle."""
    feat, flen = R.speech2mel(wave, lens, **FRONTEND)
    feat, flen = R.FeatureNormalization()(feat, flen, epoch=1)
    if draws is None:
        draws = R.specaug_draw(flen, feat.size(0), 80, **SPECAUG)
    feat = R.specaug_apply(feat, flen, draws)
    PP = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in prenet.state_dict().items()}
    x, xlen, _ = R.conv2d_prenet(feat, flen, PP, "", **PRENET, training=True)
    P = _oracle_params(model)
    P.update({"prenet." + k: v for k, v in PP.items()})
    cfg = dict(enc_layers=CFG["enc_layers"], dec_layers=CFG["dec_layers"], num_heads=CFG["num_heads"], d_model=CFG["d_model"],
               activation="GELU", label_smoothing=0.2, ctc_weight=0.3)
    out = R.asr_forward_loss(x, xlen, text, text_len, P, cfg)
    out["loss"].backward()
    return out, P, (feat, flen, x.detach(), xlen, draws)


def test_transformer_small_plumbing_on_the_cpu_oracle():
    wave, lens, text, text_len = _batch()
    prenet, model = _model_and_prenet()
    torch.manual_seed(1)
    out, P, (feat, flen, x, xlen, draws) = _oracle_step(prenet, model, wave, lens, text, text_len)
    want_flen = [int(n) // 160 + 1 for n in lens]  # centred STFT, hop 160 (speech2linear.py:201-212)
    want_xlen = [((n - 3) // 2 + 1 - 3) // 2 + 1 for n in want_flen]  # two k3 / s2 convolutions (prenet/conv2d.py:257-266)
    assert feat.shape == (4, want_flen[0], 80) and flen.tolist() == want_flen
    assert x.shape == (4, want_xlen[0], 144) and xlen.tolist() == want_xlen
    assert out["logits"].shape == (4, 11, 1000) and out["ctc_logits"].shape == (4, want_xlen[0], 1000)
    for k in ("ce", "ctc", "loss"):
        assert torch.isfinite(out[k]) and float(out[k]) > 0
    grads = [v.grad for k, v in P.items() if v.requires_grad]
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads)
    # the gradient reaches the conv prenet through the encoder input
    assert all(v.grad is not None for k, v in P.items() if k.startswith("prenet.") and v.requires_grad)
    # reproducible: the same draws give the same loss
    prenet2, model2 = _model_and_prenet()
    out2, _, _ = _oracle_step(prenet2, model2, wave, lens, text, text_len, draws=draws)
    assert float(out2["loss"]) == float(out["loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_transformer_small_hip_path_vs_oracle(mode):
    from speechain_amd import functional as Fn
    from speechain_amd.module.augment.specaug import SpecAugment
    from speechain_amd.module.frontend.speech2mel import Speech2MelSpec
    from speechain_amd.module.norm.feat_norm import FeatureNormalization

    wave, lens, text, text_len = _batch()
    prenet, model = _model_and_prenet()
    torch.manual_seed(1)
    ref, P, (rfeat, rflen, rx, rxlen, draws) = _oracle_step(prenet, model, wave, lens, text, text_len)
    Fn.set_compute_dtype(mode)
    try:
        # frontend -> normalisation -> SpecAugment (the oracle's draws) on the GPU
        fe = Speech2MelSpec(**FRONTEND).cuda()
        feat, flen = fe(wave.cuda(), lens.cuda())
        feat, flen = FeatureNormalization(input_size=80).cuda().train()(feat, flen, epoch=1)
        sa = SpecAugment(input_size=80, **SPECAUG).cuda().train()
        feat = sa.apply(feat, {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in draws.items()})
        assert torch.equal(flen.cpu(), rflen)
        assert float((feat.cpu() - rfeat).abs().max()) < 2e-3  # log-mel 1e-3, normalised by a std of ~0.5-1
        # Conv2dPrenet (row f2, training-mode BatchNorm) on the oracle's augmented features, then the named path on ITS output
        f32 = mode == torch.float32
        gp = prenet.cuda().train()
        x, xlen = gp(rfeat.cuda(), rflen.cuda())
        assert torch.equal(xlen.cpu(), rxlen)
        xe = float((x.float().cpu() - rx).norm() / rx.norm())
        assert xe < (1e-5 if f32 else 2e-2), xe
        gm = model.cuda().train()
        out = gm(x, xlen, text.cuda(), text_len.cuda())
        relf = lambda a, b, floor=0.0: float((a.double() - b.double()).norm() / (b.double().norm() + floor + 1e-300))
        for k in ("ce", "ctc", "loss"):  # f32: 1e-3 relative (north_star); bf16: 5e-3
            a, b = float(out[k]), float(ref[k])
            assert abs(a - b) < (1e-3 if f32 else 5e-3) * abs(b) + (2e-4 if f32 else 0.0), (k, a, b)
        for k in ("logits", "ctc_logits"):  # relative Frobenius: f32 1e-4, bf16 2e-2 (12 + 6 layers of bf16 products, head size 36: unfused attention)
            e = relf(out[k].detach().float().cpu(), ref[k].detach())
            assert e < (1e-4 if f32 else 2e-2), (k, e)
        out["loss"].backward()
        pairs = [(gm.encoder.trfm_layers[0].multihead_att.q_layer.weight, "encoder.trfm_layers.0.multihead_att.q_layer.weight"),
                 (gm.encoder.trfm_layers[11].feed_forward.out_layer.weight, "encoder.trfm_layers.11.feed_forward.out_layer.weight"),
                 (gm.decoder.trfm_layers[5].encdec_att.k_layer.bias, "decoder.trfm_layers.5.encdec_att.k_layer.bias"),
                 (gm.decoder.trfm_layers[0].self_att_ln.weight, "decoder.trfm_layers.0.self_att_ln.weight"),
                 (gm.embed.embed.weight, "embed.weight"), (gm.ctc_layer.linear.weight, "ctc_layer.linear.weight")]
        total = sum(float(v.grad.double().pow(2).sum()) for v in P.values() if v.grad is not None) ** 0.5
        for prm, key in pairs:  # relative Frobenius per parameter (floor: 3e-4 of the global gradient norm, for the key bias whose true gradient is zero)
            e = relf(prm.grad.detach().float().cpu(), P[key].grad, floor=3.3e-4 * total)
            assert e < (2e-3 if f32 else 4e-2), (key, e)
        for name, prm in gp.named_parameters():  # the prenet's gradients came back through 12 encoder layers
            e = relf(prm.grad.detach().float().cpu(), P["prenet." + name].grad, floor=3.3e-4 * total)
            assert e < (5e-4 if f32 else 6e-2), (name, e)
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()

"""BASELINE configs[2] at FULL size in the exact-f32 parity mode: the d512 / 8-head / ffn2048 / 12 + 6-layer / V5000 ASR transformer on a
32 x 249 x 512 encoder input and 32 x 64 token ids -- HIP forward + backward against the CPU oracle (a few seconds on the box's host
cores).  Tolerances: losses 1e-4 relative (north_star: logits 1e-3 rel); encoder output and logits 2e-4 abs / 1e-3 rel; gradients of
parameters spread over the depth of both stacks 2e-3 relative Frobenius error (sums over 7968 positions through 18 layers in float32)."""
import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu


def test_transformer_base_full_size_step_vs_oracle():
    from speechain_amd import functional as Fn
    from speechain_amd.harness import ASRTransformer

    torch.set_num_threads(16)
    B, T, d, L, V = 32, 249, 512, 64, 5000
    gen = torch.Generator().manual_seed(2024)
    feat = torch.randn(B, T, d, generator=gen) * 0.7
    flen = torch.randint(120, T + 1, (B,), generator=gen)
    flen[0] = T
    text = torch.randint(1, V - 2, (B, L), generator=gen)
    text_len = torch.randint(20, L + 1, (B,), generator=gen)
    text_len[0] = L
    text[:, 0] = V - 1
    for i in range(B):
        text[i, text_len[i] - 1] = V - 1
        text[i, text_len[i]:] = 0
    torch.manual_seed(7)
    cfg = dict(d_model=d, num_heads=8, enc_layers=12, dec_layers=6, fdfwd_dim=2048, vocab_size=V)
    model = ASRTransformer(**cfg, dropout=0.0, activation="GELU", posenc_maxlen=300, label_smoothing=0.2, ctc_weight=0.3)
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("posenc")) for k, v in model.state_dict().items()}
    P["embed.weight"] = P.pop("embed.embed.weight")
    ocfg = dict(enc_layers=12, dec_layers=6, num_heads=8, d_model=d, activation="GELU", label_smoothing=0.2, ctc_weight=0.3)
    ref = R.asr_forward_loss(feat, flen, text, text_len, P, ocfg)
    ref["loss"].backward()
    Fn.set_compute_dtype(torch.float32)
    try:
        gm = model.cuda().train()
        out = gm(feat.cuda(), flen.cuda(), text.cuda(), text_len.cuda())
        for k in ("ce", "ctc", "loss"):
            a, b = float(out[k]), float(ref[k])
            assert abs(a - b) < 1e-4 * abs(b), (k, a, b)
        assert torch.allclose(out["enc"].detach().float().cpu(), ref["enc"].detach(), atol=2e-4, rtol=1e-3)
        assert torch.allclose(out["logits"].detach().float().cpu(), ref["logits"].detach(), atol=2e-4, rtol=1e-3)
        out["loss"].backward()
        pairs = [(gm.encoder.trfm_layers[0].multihead_att.q_layer.weight, "encoder.trfm_layers.0.multihead_att.q_layer.weight"),
                 (gm.encoder.trfm_layers[0].att_layernorm.weight, "encoder.trfm_layers.0.att_layernorm.weight"),
                 (gm.encoder.trfm_layers[6].feed_forward.in_layer.weight, "encoder.trfm_layers.6.feed_forward.in_layer.weight"),
                 (gm.encoder.trfm_layers[11].feed_forward.out_layer.bias, "encoder.trfm_layers.11.feed_forward.out_layer.bias"),
                 (gm.decoder.trfm_layers[0].self_att.v_layer.weight, "decoder.trfm_layers.0.self_att.v_layer.weight"),
                 (gm.decoder.trfm_layers[5].encdec_att.k_layer.weight, "decoder.trfm_layers.5.encdec_att.k_layer.weight"),
                 (gm.decoder.layernorm.bias, "decoder.layernorm.bias"), (gm.embed.embed.weight, "embed.weight"),
                 (gm.postnet.linear.weight, "postnet.linear.weight"), (gm.ctc_layer.linear.bias, "ctc_layer.linear.bias")]
        for prm, key in pairs:
            g, r = prm.grad.detach().float().cpu(), P[key].grad
            e = float((g - r).norm() / (r.norm() + 1e-30))
            assert e < 2e-3, (key, e)
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()

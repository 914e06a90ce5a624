"""GPU parity of the transformer path (HIP kernels through the C ABI) against the golden vectors generated from the
reference modules: forward outputs, returned attention maps, and input/parameter gradients.

float32 mode is the parity mode (exact-f32 matrix cores): tolerances 2e-4 abs on activations, 1e-3 rel on gradients.
bfloat16 mode (training) is checked against the same fixtures with bf16-rounding tolerances."""
import math

import pytest
import torch

from conftest import sub
from oracle import cpu_ref as R
from speechain_amd import functional as Fn

pytestmark = pytest.mark.gpu

F32 = dict(atol=2e-4, rtol=1e-3)
# bf16 training mode: every comparison is a RELATIVE FROBENIUS error ||hip - ref|| / ||ref|| against the float32 fixture (an absolute
# tolerance says nothing without the tensor's scale).  One bf16 rounding is 2^-9 = 2e-3 relative per element; a layer output carries a few
# of them, a gradient a few more.  `floor` (added to the denominator) is for references that are zero in exact arithmetic.
BF16 = dict(rel=1.5e-2)
BF16_GRAD = dict(rel=3e-2)
BF16_PROB = dict(rel=2e-2)
MEASURED = []  # (what, relative error) of every bf16 comparison, printed at the end of the module with -s


@pytest.fixture(params=[torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def mode(request):
    Fn.set_compute_dtype(request.param)
    yield request.param
    Fn.set_compute_dtype(None)


def tol(mode):
    return F32 if mode == torch.float32 else BF16


def close(a, b, t, what=""):
    a = a.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if "rel" in t:
        e = float((a.double() - b.double()).norm() / (b.double().norm() + t.get("floor", 0.0) + 1e-300))
        MEASURED.append((what, e))
        assert e < t["rel"], (what, e, t)
        return
    err = float((a - b).abs().max())
    assert torch.allclose(a, b, **t), (what, err)


def teardown_module(module):
    if MEASURED:
        print("\nbf16 relative Frobenius errors:", ", ".join(f"{w} {e:.2e}" for w, e in MEASURED))


def _w(x):
    return torch.linspace(-1, 1, x.numel(), device=x.device).reshape(x.shape)


def _load(mod, sd):
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("posenc.posenc") or k.endswith("posenc") for k in missing), missing
    return mod.cuda().eval()


def test_posenc(golden, mode):
    from speechain_amd.module.transformer.pos_enc import PositionalEncoding

    g = golden("g3_posenc")
    for typ in ("mix", "sep"):
        pe = PositionalEncoding(posenc_type=typ, d_model=64, max_len=300)
        assert torch.equal(pe.posenc[0, [0, 1, 2, 17, 150, 299]], g[f"{typ}_rows"])
    pe = _load(PositionalEncoding(d_model=64, emb_scale=True, emb_layernorm=True, posenc_scale=True, max_len=300), sub(g, "full_sd."))
    close(pe(g["x"].cuda()), g["full_out"], tol(mode), "posenc full")
    pe2 = PositionalEncoding(d_model=64, max_len=300).cuda().eval()
    close(pe2(g["x"].cuda()), g["plain_out"], tol(mode), "posenc plain")


def test_mha(golden, mode):
    from speechain_amd.module.transformer.attention import MultiHeadedAttention

    g = golden("g4_layers")
    t = tol(mode)
    for tag, by_head in (("mha_dmodel", False), ("mha_byhead", True)):
        m = _load(MultiHeadedAttention(num_heads=4, d_model=64, dropout=0.0, scale_dp_by_head=by_head), sub(g, f"{tag}.sd."))
        q, kv = g[f"{tag}.q"].cuda().requires_grad_(), g[f"{tag}.kv"].cuda().requires_grad_()
        y, p = m(kv, kv, q, g[f"{tag}.mask"].cuda())
        close(y, g[f"{tag}.out"], t, tag + " out")
        close(p, g[f"{tag}.probs"], dict(atol=1e-5, rtol=1e-4) if mode == torch.float32 else BF16_PROB, tag + " probs")
        gq, gkv, gw = torch.autograd.grad((y.float() * _w(y)).sum(), [q, kv, m.q_layer.weight])
        close(gq, g[f"{tag}.gq"], t, tag + " gq")
        close(gkv, g[f"{tag}.gkv"], t, tag + " gkv")
        close(gw, g[f"{tag}.gw_q"], dict(atol=2e-3, rtol=1e-3) if mode == torch.float32 else BF16_GRAD, tag + " gw")
    # dh = 192, dense (B, T, T) causal & padding mask given explicitly
    m = _load(MultiHeadedAttention(num_heads=2, d_model=384, dropout=0.0), sub(g, "mha192.sd."))
    x = g["mha192.x"].cuda().requires_grad_()
    y, p = m(x, x, x, g["mha192.mask"].cuda())
    close(y, g["mha192.out"], t, "mha192 out")
    close(p, g["mha192.probs"], dict(atol=1e-5, rtol=1e-4) if mode == torch.float32 else BF16_PROB, "mha192 probs")
    (gx,) = torch.autograd.grad((y.float() * _w(y)).sum(), [x])
    close(gx, g["mha192.gx"], t if mode == torch.float32 else BF16_GRAD, "mha192 gx")
    # same result when the causal structure is given as a flag + (B,1,T) padding mask (what the decoder stack does)
    pad = g["mha192.mask"][:, -1:, :].cuda()
    y2, p2 = m(x, x, x, pad, _causal=True)
    close(y2, g["mha192.out"], t, "mha192 causal-flag out")


@pytest.mark.parametrize("tag,act,typ", [("ffn_relu", "ReLU", "linear"), ("ffn_gelu", "GELU", "linear"), ("ffn_conv", "ReLU", "conv")])
def test_ffn(golden, mode, tag, act, typ):
    from speechain_amd.module.transformer.feed_forward import PositionwiseFeedForward

    g = golden("g4_layers")
    f = _load(PositionwiseFeedForward(d_model=64, fdfwd_dim=160, dropout=0.0, fdfwd_activation=act, fdfwd_type=typ), sub(g, f"{tag}.sd."))
    x = g[f"{tag}.x"].cuda().requires_grad_()
    y = f(x)
    t = tol(mode)
    close(y, g[f"{tag}.out"], t, tag)
    w = f.in_layer.weight if typ == "linear" else f.in_layer.conv_lyr.weight
    gx, gw = torch.autograd.grad((y.float() * _w(y)).sum(), [x, w])
    close(gx, g[f"{tag}.gx"], t, tag + " gx")
    close(gw, g[f"{tag}.gw_in"], dict(atol=2e-3, rtol=1e-3) if mode == torch.float32 else BF16_GRAD, tag + " gw")


def test_encoder_decoder_layers(golden, mode):
    from speechain_amd.module.transformer.decoder import TransformerDecoderLayer
    from speechain_amd.module.transformer.encoder import TransformerEncoderLayer

    g = golden("g4_layers")
    t = tol(mode)
    tg = dict(atol=2e-3, rtol=2e-3) if mode == torch.float32 else BF16_GRAD
    for tag, first in (("enc_pre", True), ("enc_post", False)):
        e = _load(TransformerEncoderLayer(d_model=64, num_heads=4, fdfwd_dim=128, att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0,
                                          fdfwd_activation="GELU", layernorm_first=first), sub(g, f"{tag}.sd."))
        x = g[f"{tag}.x"].cuda().requires_grad_()
        y, p = e(x, g[f"{tag}.mask"].cuda())
        close(y, g[f"{tag}.out"], t, tag)
        gx, gln, gb = torch.autograd.grad((y.float() * _w(y)).sum(), [x, e.att_layernorm.weight, e.multihead_att.output_layer.bias])
        close(gx, g[f"{tag}.gx"], tg, tag + " gx")
        close(gln, g[f"{tag}.g_attln_w"], tg, tag + " gln")
        close(gb, g[f"{tag}.g_outb"], tg, tag + " gb")
    d = _load(TransformerDecoderLayer(d_model=64, num_heads=4, fdfwd_dim=128, att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0,
                                      fdfwd_activation="GELU", scale_dp_by_head=True), sub(g, "dec.sd."))
    tgt, src = g["dec.tgt"].cuda().requires_grad_(), g["dec.src"].cuda().requires_grad_()
    y, sp, ep = d(tgt, src, g["dec.tmask"].cuda(), g["dec.smask"].cuda())
    close(y, g["dec.out"], t, "dec layer")
    pt = dict(atol=1e-5, rtol=1e-4) if mode == torch.float32 else BF16_PROB
    close(sp, g["dec.self_probs"], pt, "dec self probs")
    close(ep, g["dec.encdec_probs"], pt, "dec encdec probs")
    gt, gs = torch.autograd.grad((y.float() * _w(y)).sum(), [tgt, src])
    close(gt, g["dec.gtgt"], tg, "dec gtgt")
    close(gs, g["dec.gsrc"], tg, "dec gsrc")


def test_stacks(golden, mode):
    from speechain_amd.module.transformer.decoder import TransformerDecoder
    from speechain_amd.module.transformer.encoder import TransformerEncoder

    g = golden("g5_stacks")
    t = tol(mode) if mode == torch.float32 else dict(rel=2e-2)  # (two layers + final LayerNorm)
    pt = dict(atol=1e-5, rtol=1e-4) if mode == torch.float32 else dict(rel=3e-2)
    enc = _load(TransformerEncoder(d_model=64, num_heads=4, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0,
                                   fdfwd_dropout=0.0, res_dropout=0.0, fdfwd_activation="GELU", posenc_maxlen=200), sub(g, "enc.sd."))
    y, m, att, hid = enc(g["enc.x"].cuda(), g["enc.mask"].cuda())
    close(y, g["enc.out"], t, "enc out")
    assert torch.equal(m.cpu(), g["enc.mask"])
    close(att[0], g["enc.att0"], pt, "att0")
    close(att[1], g["enc.att1"], pt, "att1")
    close(hid[0], g["enc.hid0"], t, "hid0")
    close(hid[1], g["enc.hid1"], t, "hid1")
    enc2 = _load(TransformerEncoder(d_model=48, num_heads=2, num_layers=2, fdfwd_dim=96, fdfwd_type="conv", posenc_dropout=0.0,
                                    att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0, posenc_scale=True, emb_layernorm=True,
                                    scale_dp_by_head=True, uni_direction=True, layernorm_first=False, posenc_maxlen=200), sub(g, "enc2.sd."))
    y2, m2, att2, _ = enc2(g["enc2.x"].cuda(), g["enc2.mask"].cuda())
    close(y2, g["enc2.out"], t, "enc2 out")
    assert torch.equal(m2.cpu(), g["enc2.mask_out"])
    close(att2[1], g["enc2.att1"], pt, "enc2 att1")
    dec = _load(TransformerDecoder(d_model=64, num_heads=4, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0,
                                   fdfwd_dropout=0.0, res_dropout=0.0, fdfwd_activation="GELU", emb_scale=False, posenc_maxlen=200),
                sub(g, "dec.sd."))
    yd, sa, ea, hd = dec(g["dec.tgt"].cuda(), g["enc.out"].cuda(), g["dec.tmask"].cuda(), g["enc.mask"].cuda())
    close(yd, g["dec.out"], t, "dec out")
    close(sa[1], g["dec.self_att1"], pt, "dec self att")
    close(ea[1], g["dec.encdec_att1"], pt, "dec encdec att")
    close(hd[0], g["dec.hid0"], t, "dec hid0")


@pytest.mark.parametrize("R,D", [(7968, 512), (2016, 512), (37, 64), (301, 520), (129, 1536), (65, 2048), (50, 100)])
def test_layernorm_bf16_vs_torch_fp32(R, D):
    """nn.LayerNorm(eps=1e-6) forward and backward in bf16 through the 16-byte-access kernels (D % 8 == 0) and the
    element-per-lane ones (D = 100) against torch fp32 on the same bf16-rounded inputs.  Outputs are bf16: one rounding
    (2^-8 relative) on top of fp32 arithmetic; parameter gradients are f32 sums over R rows of bf16-rounded dy."""
    from speechain_amd import functional as Fn

    gen = torch.Generator().manual_seed(R + D)
    x = (torch.randn(R, D, generator=gen) * 2 + 0.5).bfloat16()
    w = torch.randn(R, D, generator=gen).bfloat16()
    gamma = (1 + 0.1 * torch.randn(D, generator=gen))
    beta = 0.1 * torch.randn(D, generator=gen)
    xr, gr, br = x.float().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    gxr, ggr, gbr = torch.autograd.grad((yr * w.float()).sum(), [xr, gr, br])
    xc, gc, bc = x.cuda().requires_grad_(), gamma.cuda().requires_grad_(), beta.cuda().requires_grad_()
    y = Fn.layer_norm(xc, gc, bc, 1e-6)
    assert y.dtype == torch.bfloat16
    gx, gg, gb = torch.autograd.grad((y.float() * w.cuda().float()).sum(), [xc, gc, bc])
    assert torch.allclose(y.float().cpu(), yr, atol=2e-2, rtol=1e-2), float((y.float().cpu() - yr).abs().max())
    assert torch.allclose(gx.float().cpu(), gxr, atol=3e-2, rtol=2e-2), float((gx.float().cpu() - gxr).abs().max())
    scale = math.sqrt(R)
    assert float((gg.cpu() - ggr).abs().max()) < 2e-2 * scale and float((gb.cpu() - gbr).abs().max()) < 2e-2 * scale


def test_losses(golden, mode):
    from speechain_amd.criterion.cross_entropy import CrossEntropy
    from speechain_amd.criterion.ctc import CTCLoss

    g = golden("g6_losses")
    # (bf16 mode feeds the criteria bf16-ROUNDED logits: the loss moves by the rounding of its input, not by the kernels' arithmetic)
    lt = dict(atol=1e-4, rtol=1e-4) if mode == torch.float32 else dict(rel=5e-3)
    gt = dict(atol=1e-6, rtol=1e-3) if mode == torch.float32 else dict(rel=2e-2)
    for tag, conf in (("ce0", dict()), ("ce02", dict(label_smoothing=0.2)), ("ce02n", dict(label_smoothing=0.2, length_normalized=True))):
        lg = g["ce_logits"].to(mode).cuda().requires_grad_()
        loss = CrossEntropy(**conf)(logits=lg, text=g["text"].cuda(), text_len=g["text_len"].cuda())
        close(loss, g[f"{tag}.loss"], lt, tag)
        close(torch.autograd.grad(loss, lg)[0], g[f"{tag}.grad"], gt, tag + " grad")
    lg = g["ctc_logits"].to(mode).cuda().requires_grad_()
    loss = CTCLoss()(lg, g["enc_len"].cuda(), g["ctc_text"].cuda(), g["text_len"].cuda())
    close(loss, g["ctc_loss"], lt, "ctc")
    close(torch.autograd.grad(loss, lg)[0], g["ctc_grad"], gt, "ctc grad")


def test_cross_entropy_options(golden, mode, tmp_path):
    """Token re-weighting and confidence thresholding (cross_entropy.py:81-95,148-199) against the imported reference (g13_ce_options):
    losses and gradients; the weights also through the vocabulary-file constructor path."""
    from speechain_amd.criterion.cross_entropy import CrossEntropy

    g = golden("g13_ce_options")
    # (bf16 mode feeds the criteria bf16-ROUNDED logits: the loss moves by the rounding of its input, not by the kernels' arithmetic)
    lt = dict(atol=1e-4, rtol=1e-4) if mode == torch.float32 else dict(rel=5e-3)
    gt = dict(atol=1e-6, rtol=1e-3) if mode == torch.float32 else dict(rel=2e-2)
    vocab = tmp_path / "vocab"
    vocab.write_text("\n".join(f"tok{i}" for i in range(29)) + "\n")
    cases = dict(w=dict(label_smoothing=0.1, token_vocab=str(vocab), new_weights={"tok3": 2.5, "tok11": 0.25, "tok28": 0.0}),
                 tok=dict(confid_threshold=0.05, confid_level="token", length_normalized=True),
                 sent=dict(confid_threshold=0.02, confid_level="sentence"),
                 tok_ls=dict(confid_threshold=0.08, confid_level="token", label_smoothing=0.2))
    for tag, conf in cases.items():
        lg = g["logits"].to(mode).cuda().requires_grad_()
        loss = CrossEntropy(**conf)(logits=lg, text=g["text"].cuda(), text_len=g["text_len"].cuda())
        if mode == torch.float32 or tag == "w":  # (bf16 logits can move a token across a confidence threshold: compared in f32 only)
            close(loss, g[f"{tag}.loss"], lt, tag)
            close(torch.autograd.grad(loss, lg)[0], g[f"{tag}.grad"], gt, tag + " grad")


def test_ctc_edge_cases_vs_oracle():
    """Infeasible alignment (more labels + forced blanks than frames: zero_infinity path), an utterance shorter than the
    padded batch (zero gradient beyond its length), a label repeated three times, a single-label target."""
    from speechain_amd.criterion.ctc import CTCLoss

    gen = torch.Generator().manual_seed(5)
    B, T, V = 4, 9, 23
    logits = torch.randn(B, T, V, generator=gen)
    E = V - 1  # <sos/eos>, at both ends of every sentence as the criterion's interface requires (criterion/ctc.py:48-52)
    text = torch.tensor([[E, 3, 4, 5, 6, E, 0, 0, 0, 0],      # feasible
                         [E, 7, 7, 7, 7, 7, 7, E, 0, 0],      # 6 equal labels need 11 frames > 9: infinite loss -> zeroed
                         [E, 9, 2, 9, 8, 9, E, 0, 0, 0],      # the same label three times, not adjacent
                         [E, 11, E, 0, 0, 0, 0, 0, 0, 0]])    # one label
    text_len = torch.tensor([6, 8, 7, 3])
    enc_len = torch.tensor([9, 9, 7, 3])
    lg_ref = logits.clone().requires_grad_()
    ref = R.ctc_loss(lg_ref, enc_len, text, text_len)
    gref = torch.autograd.grad(ref, lg_ref)[0]
    lg = logits.cuda().requires_grad_()
    loss = CTCLoss()(lg, enc_len.cuda(), text.cuda(), text_len.cuda())
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    g = torch.autograd.grad(loss, lg)[0].cpu()
    assert float((g - gref).abs().max()) < 1e-6 + 1e-4 * float(gref.abs().max())
    assert float(g[1].abs().max()) == 0.0 and float(g[2, 7:].abs().max()) == 0.0 and float(g[3, 3:].abs().max()) == 0.0


def test_ctc_full_size_vs_oracle():
    """BASELINE shapes: (32, 249, 5000) logits, 62 labels incl. repeats; value and gradient vs torch's CPU CTC."""
    from speechain_amd.criterion.ctc import CTCLoss

    gen = torch.Generator().manual_seed(3)
    B, T, V, L = 32, 249, 5000, 64
    logits = torch.randn(B, T, V, generator=gen)
    text = torch.randint(1, V - 2, (B, L), generator=gen)
    text[:, 10] = text[:, 9]  # a repeated label
    text_len = torch.randint(20, L + 1, (B,), generator=gen)
    text_len[0] = L
    enc_len = torch.randint(150, T + 1, (B,), generator=gen)
    enc_len[0] = T
    for i in range(B):
        text[i, 0] = V - 1
        text[i, text_len[i] - 1] = V - 1
        text[i, text_len[i]:] = 0
    lg_ref = logits.clone().requires_grad_()
    ref = R.ctc_loss(lg_ref, enc_len, text, text_len)
    gref = torch.autograd.grad(ref, lg_ref)[0]
    lg = logits.cuda().requires_grad_()
    loss = CTCLoss()(lg, enc_len.cuda(), text.cuda(), text_len.cuda())
    assert abs(float(loss) - float(ref)) < 1e-3 * abs(float(ref))
    g = torch.autograd.grad(loss, lg)[0].cpu()
    assert float((g - gref).abs().max()) < 1e-6 + 1e-3 * float(gref.abs().max())


def test_harness_step(golden, mode):
    """a-harness: encoder -> decoder -> heads -> 0.7 CE + 0.3 CTC with the reference's sequencing, values and gradients."""
    from speechain_amd.harness import ASRTransformer

    g = golden("g7_harness")
    model = ASRTransformer(d_model=64, num_heads=4, enc_layers=2, dec_layers=2, fdfwd_dim=128, vocab_size=53, dropout=0.0,
                           posenc_maxlen=200, label_smoothing=0.2, ctc_weight=0.3)
    sd = sub(g, "sd.")
    sd["embed.embed.weight"] = sd.pop("embed.weight")
    _load(model, sd)
    out = model(g["feat"].cuda(), g["feat_len"].cuda(), g["text"].cuda(), g["text_len"].cuda())
    t = tol(mode) if mode == torch.float32 else dict(rel=2e-2)
    close(out["enc"], g["enc"], t, "enc")
    close(out["logits"], g["logits"], t, "logits")
    close(out["ctc_logits"], g["ctc_logits"], t, "ctc_logits")
    lt = dict(atol=2e-4, rtol=2e-4) if mode == torch.float32 else dict(rel=5e-3)
    for k in ("ce", "ctc", "loss"):
        close(out[k], g[k], lt, k)
    out["loss"].backward()
    gt = dict(atol=2e-4, rtol=2e-3) if mode == torch.float32 else dict(rel=4e-2)
    close(model.encoder.trfm_layers[0].multihead_att.q_layer.weight.grad, g["g_enc0_q"], gt, "g enc0 q")
    close(model.embed.embed.weight.grad, g["g_embed"], gt, "g embed")
    # (a key projection's bias has NO gradient in exact arithmetic -- softmax ignores a shift all keys share --: the fixture holds the
    #  reference's rounding residue; measured against the scale of the query weight's gradient instead of its own)
    kb_floor = float(g["g_enc0_q"].norm())
    close(model.decoder.trfm_layers[1].encdec_att.k_layer.bias.grad, g["g_dec1_encdec_kb"],
          gt if mode == torch.float32 else dict(rel=1e-3, floor=kb_floor), "g kb")
    close(model.ctc_layer.linear.weight.grad, g["g_ctc_w"], gt, "g ctc w")


def test_hipgraph_replay_matches_eager_steps():
    """Two identical small models: one trained eagerly, one with the step (fwd + bwd + SGD) captured in a hipGraph and
    replayed; dropout off so both see identical arithmetic -> identical losses for several steps."""
    from speechain_amd.harness import ASRTransformer

    Fn.set_compute_dtype(torch.bfloat16)
    try:
        V = 29

        def make():
            torch.manual_seed(7)
            return ASRTransformer(d_model=64, num_heads=4, enc_layers=2, dec_layers=1, fdfwd_dim=128, vocab_size=V, dropout=0.0,
                                  posenc_maxlen=100).cuda().train()

        gen = torch.Generator().manual_seed(8)
        x = torch.randn(4, 30, 64, generator=gen).cuda()
        xlen = torch.tensor([30, 25, 22, 16]).cuda()
        text = torch.randint(1, V - 2, (4, 8), generator=gen)
        tl = torch.tensor([8, 7, 6, 5])
        for i, l in enumerate(tl):
            text[i, 0] = V - 1
            text[i, l - 1] = V - 1
            text[i, l:] = 0
        text, tl = text.cuda(), tl.cuda()

        def step(model, opt):
            opt.zero_grad(set_to_none=False)
            out = model(x, xlen, text, tl)
            out["loss"].backward()
            opt.step()
            return out["loss"].detach()

        m1, m2 = make(), make()
        o1 = torch.optim.SGD(m1.parameters(), lr=0.05)
        o2 = torch.optim.SGD(m2.parameters(), lr=0.05)
        eager = [float(step(m1, o1)) for _ in range(6)]
        warm = [float(step(m2, o2)) for _ in range(2)]  # also creates the .grad tensors the graph will reuse
        Fn.weights.invalidate()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = step(m2, o2)
        replayed = []
        for _ in range(3):
            g.replay()
            replayed.append(float(loss))
        torch.cuda.synchronize()
        # capture itself does not execute: eager steps 0,1 then replays = steps 2,3,4
        assert warm == eager[:2]
        for a, b in zip(replayed, eager[2:5]):
            assert abs(a - b) <= 2e-3 * abs(b), (replayed, eager)
        assert eager[5] < eager[0]  # and it actually learns
    finally:
        Fn.set_compute_dtype(None)


def test_gradient_sinks_match_plain_autograd():
    """With GradBucketReducer's sinks the kernels write parameter gradients straight into the flat buckets; the values must
    equal those of plain autograd accumulation (no reducer), including the fused q/k/v slots."""
    from speechain_amd.ddp import GradBucketReducer, attention_fusion_groups
    from speechain_amd.harness import ASRTransformer

    Fn.set_compute_dtype(torch.float32)
    try:
        V = 23

        def make():
            torch.manual_seed(11)
            return ASRTransformer(d_model=64, num_heads=4, enc_layers=2, dec_layers=2, fdfwd_dim=128, vocab_size=V, dropout=0.0,
                                  posenc_maxlen=100).cuda().train()

        gen = torch.Generator().manual_seed(12)
        x = torch.randn(3, 20, 64, generator=gen).cuda()
        xlen = torch.tensor([20, 15, 11]).cuda()
        text = torch.randint(1, V - 2, (3, 7), generator=gen)
        tl = torch.tensor([7, 6, 4])
        for i, l in enumerate(tl):
            text[i, 0] = V - 1
            text[i, l - 1] = V - 1
            text[i, l:] = 0
        text, tl = text.cuda(), tl.cuda()
        ref = make()
        ref(x, xlen, text, tl)["loss"].backward()
        want = {n: p.grad.clone() for n, p in ref.named_parameters()}
        m = make()
        red = GradBucketReducer(m.parameters(), bucket_mb=0.25, adjacent=attention_fusion_groups(m))
        for _ in range(2):  # twice: the second step must not see stale slots or double counting
            red.zero_grad()
            m(x, xlen, text, tl)["loss"].backward()
            red.wait()
        flat_ranges = [(b["flat"].data_ptr(), b["flat"].data_ptr() + 4 * b["flat"].numel()) for b in red.buckets]
        for n, p in m.named_parameters():
            assert p.grad is not None, n
            assert any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_ranges), n
            assert torch.allclose(p.grad, want[n], atol=1e-5, rtol=1e-4), (n, float((p.grad - want[n]).abs().max()))
    finally:
        Fn.GradSinks.clear()
        Fn.set_compute_dtype(None)


@pytest.mark.parametrize("sinks", [False, True])
def test_side_stream_branches_match_one_stream(monkeypatch, sinks):
    """The CTC head + loss and the decoder's encoder-side key|value projections run on streams of their own, forward and backward
    (harness.py, decoder.py): losses and every parameter gradient must equal the one-stream run, eagerly and step after step (a missing
    join shows up as a stale or half-written gradient), with plain autograd accumulation and with the gradient sinks + grouped weight
    gradients of the training step."""
    from speechain_amd import harness
    from speechain_amd.ddp import GradBucketReducer, attention_fusion_groups
    from speechain_amd.module.transformer import decoder as dec_mod

    Fn.set_compute_dtype(torch.bfloat16 if sinks else torch.float32)
    try:
        V = 31

        def make():
            torch.manual_seed(21)
            return harness.ASRTransformer(d_model=128, num_heads=2, enc_layers=2, dec_layers=3, fdfwd_dim=256, vocab_size=V, dropout=0.0,
                                          posenc_maxlen=200).cuda().train()

        gen = torch.Generator().manual_seed(22)
        x = torch.randn(6, 90, 128, generator=gen).cuda()
        xlen = torch.tensor([90, 81, 77, 60, 45, 33]).cuda()
        text = torch.randint(1, V - 2, (6, 12), generator=gen)
        tl = torch.tensor([12, 11, 9, 8, 6, 5])
        for i, l in enumerate(tl):
            text[i, 0] = V - 1
            text[i, l - 1] = V - 1
            text[i, l:] = 0
        text, tl = text.cuda(), tl.cuda()

        def run(branches: bool):
            monkeypatch.setattr(harness, "_CTC_BRANCH", branches)
            monkeypatch.setattr(dec_mod, "_CROSSKV_BRANCH", branches)
            Fn.GradSinks.clear()
            m = make()
            red = GradBucketReducer(m.parameters(), bucket_mb=0.25, adjacent=attention_fusion_groups(m)) if sinks else None
            outs = []
            for _ in range(3):
                if red is not None:
                    red.zero_grad()
                else:
                    m.zero_grad(set_to_none=True)
                out = m(x, xlen, text, tl)
                out["loss"].backward()
                if red is not None:
                    red.wait()
                torch.cuda.synchronize()
                outs.append((out["loss"].detach().clone(), out["ctc"].detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}))
            return outs

        one, two = run(False), run(True)
        tol = dict(atol=2e-3, rtol=2e-2) if sinks else dict(atol=1e-5, rtol=1e-4)  # (bf16: the encoder-output gradient is summed in another order)
        for (l1, c1, g1), (l2, c2, g2) in zip(one, two):
            assert torch.allclose(l1, l2, atol=1e-6, rtol=1e-6) and torch.allclose(c1, c2, atol=1e-6, rtol=1e-6), (l1, l2, c1, c2)
            for n in g1:
                assert torch.allclose(g1[n].float(), g2[n].float(), **tol), (n, float((g1[n].float() - g2[n].float()).abs().max()))
    finally:
        Fn.GradSinks.clear()
        Fn.set_compute_dtype(None)


def test_batched_weight_refresh_matches_lazy_casts():
    """weights.refresh() re-casts every cached bf16 copy (plain, fused q|k|v, transposed) in one launch; after an in-place
    parameter update the refreshed copies must equal freshly computed ones."""
    torch.manual_seed(3)
    a = torch.nn.Parameter(torch.randn(96, 64, device="cuda"))
    b = torch.nn.Parameter(torch.randn(40, 64, device="cuda"))
    c = torch.nn.Parameter(torch.randn(70, 33, device="cuda"))
    Fn.weights.invalidate()
    f1 = Fn.weights.fused([a, b], torch.bfloat16)
    t1 = Fn.weights.fused_t([a, b], torch.bfloat16)
    c1 = Fn.weights.fused([c], torch.bfloat16)
    ct = Fn.weights.fused_t([c], torch.bfloat16)
    with torch.no_grad():
        for p in (a, b, c):
            p.mul_(1.5).add_(0.25)
    Fn.weights.refresh()
    want = torch.cat([a, b]).detach().bfloat16()
    assert Fn.weights.fused([a, b], torch.bfloat16).data_ptr() == f1.data_ptr()  # cache hit, same buffers
    assert torch.equal(f1, want) and torch.equal(t1, want.t())
    assert torch.equal(c1, c.detach().bfloat16()) and torch.equal(ct, c.detach().bfloat16().t())
    Fn.weights.invalidate()


@pytest.mark.parametrize("drop_p", [0.0, 0.2])
@pytest.mark.parametrize("case", ["self_pad", "self_causal", "cross", "self_long"])
def test_fused_attention_matches_unfused(case, drop_p):
    """bf16, head size 64: the fused kernel (no score tensor in HBM) against the GEMM + softmax + GEMM path, forward and
    gradients, with padding mask, causal flag and dropout (identical masks by construction)."""
    from speechain_amd.module.transformer.attention import MultiHeadedAttention

    Fn.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(5)
        d, H, B = 256, 4, 3
        Tq, Tk = (249, 249) if case != "cross" else (63, 249)
        if case == "self_long":  # a 15 s utterance: 375 frames, two key blocks in the backward sweep
            Tq = Tk = 375
        m = MultiHeadedAttention(num_heads=H, d_model=d, dropout=drop_p).cuda().train()
        q = torch.randn(B, Tq, d, device="cuda", requires_grad=True)
        kv = q if case != "cross" else torch.randn(B, Tk, d, device="cuda", requires_grad=True)
        lens = torch.tensor([Tk, 200, 131], device="cuda")
        mask = (torch.arange(Tk, device="cuda")[None, :] < lens[:, None])[:, None, :]
        w = torch.randn(B, Tq, d, device="cuda")
        outs = []
        for fused in (False, True):
            Fn.set_attention_probs(not fused)
            Fn.DropoutState.reset(1234)
            y, p = m(kv, kv, q, mask, _causal=(case == "self_causal"))
            assert (p is None) == fused
            g = torch.autograd.grad((y.float() * w).sum(), [q] + ([kv] if case == "cross" else []) + [m.q_layer.weight, m.v_layer.bias])
            outs.append((y.float(), [t.float() for t in g]))
        (y0, g0), (y1, g1) = outs
        assert torch.allclose(y0, y1, atol=2e-2, rtol=2e-2), float((y0 - y1).abs().max())
        for a, b in zip(g0, g1):
            assert torch.allclose(a, b, atol=3e-2 * float(a.abs().max()) + 1e-3, rtol=5e-2), float((a - b).abs().max())
    finally:
        Fn.set_attention_probs(True)
        Fn.set_compute_dtype(None)


@pytest.mark.parametrize("Tq,Tk,causal", [(249, 249, False), (63, 700, False), (130, 130, True), (1, 5, False), (300, 64, False)])
def test_streaming_attention_forward_through_the_c_abi(Tq, Tk, causal):
    """sc_attention_fwd (streaming kernel: key masks, any Tk) on packed (B, T, H*64) bf16 buffers against torch fp32 attention
    of the same bf16 values: output (bf16: 2^-8 relative on top of an f32 softmax over bf16-rounded P) and the saved row
    logsumexp; an utterance whose keys are all masked gives zeros and -inf."""
    from speechain_amd._lib import lib, stream

    gen = torch.Generator().manual_seed(Tq * 1000 + Tk)
    B, H, dh = 3, 2, 64
    q = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
    k = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
    v = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
    lens = torch.tensor([Tk, max(1, Tk // 2 + 1), 0]).cuda()  # full, ragged, nothing visible
    m8 = (torch.arange(Tk, device="cuda")[None, :] < lens[:, None]).to(torch.uint8)[:, None, :].contiguous()
    out = torch.empty(B, Tq, H * dh, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H, Tq, device="cuda")
    scale = 1.0 / math.sqrt(dh)
    lib.call("sc_attention_fwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(lse), B, H, Tq, Tk, q.stride(0), q.stride(1), k.stride(0),
             k.stride(1), out.stride(0), out.stride(1), Fn._p(m8), m8.stride(0), 0, 1 if causal else 0, scale, 0.0, None, 0, stream())
    qf, kf, vf = (t.float().view(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
    s_ = (qf @ kf.transpose(-1, -2)) * scale
    ok = m8.bool()[:, :, None, :].expand(B, H, Tq, Tk).clone()
    if causal:
        ok &= (torch.arange(Tk, device="cuda")[None, :] <= torch.arange(Tq, device="cuda")[:, None])[None, None]
    s_ = s_.masked_fill(~ok, float("-inf"))
    ref_lse = torch.logsumexp(s_, -1)
    pr = torch.nan_to_num(torch.softmax(s_, -1), nan=0.0)
    ref = (pr @ vf).transpose(1, 2).reshape(B, Tq, H * dh)
    assert torch.allclose(out.float(), ref, atol=2e-2, rtol=2e-2), float((out.float() - ref).abs().max())
    got = lse.view(B, H, Tq)
    fin = torch.isfinite(ref_lse)
    assert torch.equal(torch.isfinite(got), fin)
    assert float((got[fin] - ref_lse[fin]).abs().max()) < 2e-3 if bool(fin.any()) else True
    assert float(out[2].float().abs().max()) == 0.0


@pytest.mark.parametrize("Tq,Tk,causal", [(249, 249, False), (63, 249, False), (63, 63, True), (130, 130, True), (1, 5, False), (300, 64, False),
                                          (70, 128, False), (200, 129, False), (64, 256, False), (33, 200, False),
                                          (300, 300, False), (375, 375, True), (70, 600, False), (500, 257, False)])
def test_attention_backward_through_the_c_abi(Tq, Tk, causal):
    """sc_attention_bwd (key-on-the-lane kernel; 4 or 8 waves of 32 keys per launch, one launch per block of 256 keys) on packed (B, T, H*64) bf16 buffers against torch
    autograd in fp32 on the same bf16 values: dQ, dK, dV (bf16 outputs of sums over bf16-rounded P and dS: 2 % of the largest
    element); ragged key masks, an utterance without visible keys (zero gradients), query counts that are not multiples of 64."""
    from speechain_amd._lib import lib, stream

    gen = torch.Generator().manual_seed(Tq * 1000 + Tk + 7)
    B, H, dh = 3, 2, 64
    q = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
    k = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
    v = torch.randn(B, Tk, H * dh, generator=gen).bfloat16().cuda()
    do = torch.randn(B, Tq, H * dh, generator=gen).bfloat16().cuda()
    lens = torch.tensor([Tk, max(1, Tk // 2 + 1), 0]).cuda()
    m8 = (torch.arange(Tk, device="cuda")[None, :] < lens[:, None]).to(torch.uint8)[:, None, :].contiguous()
    out = torch.empty(B, Tq, H * dh, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H, Tq, device="cuda")
    scale = 1.0 / math.sqrt(dh)
    lib.call("sc_attention_fwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(lse), B, H, Tq, Tk, q.stride(0), q.stride(1), k.stride(0),
             k.stride(1), out.stride(0), out.stride(1), Fn._p(m8), m8.stride(0), 0, 1 if causal else 0, scale, 0.0, None, 0, stream())
    dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
    ws_bytes = lib.value("sc_attention_bwd_workspace", B, H, Tq, Tk)  # more than 256 keys: dQ is summed over the key blocks through it
    assert (ws_bytes > 0) == (Tk > 256)
    ws = torch.full((max(ws_bytes // 4, 1),), float("nan"), device="cuda")
    lib.call("sc_attention_bwd", Fn._p(q), Fn._p(k), Fn._p(v), Fn._p(out), Fn._p(do), Fn._p(lse), Fn._p(dq), Fn._p(dk), Fn._p(dv), B, H, Tq, Tk,
             q.stride(0), q.stride(1), k.stride(0), k.stride(1), out.stride(0), out.stride(1), dq.stride(0), dq.stride(1), dk.stride(0),
             dk.stride(1), Fn._p(m8), m8.stride(0), 0, 1 if causal else 0, scale, 0.0, None, 0, Fn._p(ws) if ws_bytes else None, ws_bytes, stream())
    qf, kf, vf = (t.float().view(B, -1, H, dh).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
    s_ = (qf @ kf.transpose(-1, -2)) * scale
    ok = m8.bool()[:, :, None, :].expand(B, H, Tq, Tk).clone()
    if causal:
        ok &= (torch.arange(Tk, device="cuda")[None, :] <= torch.arange(Tq, device="cuda")[:, None])[None, None]
    pr = torch.nan_to_num(torch.softmax(s_.masked_fill(~ok, float("-inf")), -1), nan=0.0)
    ref = (pr @ vf).transpose(1, 2).reshape(B, Tq, H * dh)
    gq, gk, gv = torch.autograd.grad((ref * do.float()).sum(), [qf, kf, vf])
    for name, got, want in (("dq", dq, gq), ("dk", dk, gk), ("dv", dv, gv)):
        want = want.transpose(1, 2).reshape(got.shape)
        assert bool(torch.isfinite(got.float()).all()), name
        tol = 2e-2 * float(want.abs().max()) + 1e-3
        assert float((got.float() - want).abs().max()) <= tol, (name, float((got.float() - want).abs().max()), tol)
    assert float(dq[2].float().abs().max()) == 0.0 and float(dk[2].float().abs().max()) == 0.0 and float(dv[2].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_embedding_table_gradient_vs_torch(dtype):
    """sc_embedding_bwd (one wave per position, the first holder of an id sums its rows in position order) against nn.Embedding's
    autograd: repeated ids, the padding id, ids that never occur, D not a multiple of the 256-column pass, an id held by a whole row of
    positions (the owner lists its rows in LDS and loads them eight at a time)."""
    from speechain_amd import functional as Fn
    gen = torch.Generator().manual_seed(5)
    for V, D, B, L in ((5000, 512, 32, 63), (37, 100, 3, 70), (9, 8, 2, 5), (50, 24, 4, 2000)):  # (the last: N > 7680, ids read from global memory)
        table = torch.nn.Parameter(torch.randn(V, D, generator=gen).cuda())
        ids = torch.randint(0, V, (B, L), generator=gen)
        ids[0, :4] = 0           # padding id
        ids[1, :] = ids[1, 0]    # one id many times
        ids = ids.cuda()
        w = torch.randn(B, L, D, generator=gen).cuda().to(dtype)
        out = Fn.embedding(table, ids, 0)
        assert torch.equal(out, torch.nn.functional.embedding(ids, table.detach()))
        (out.to(dtype) * w).sum().backward()
        got, table.grad = table.grad.clone(), None
        ref = torch.nn.functional.embedding(ids, table, padding_idx=0)
        (ref.to(dtype) * w).sum().backward()
        assert float(got[0].abs().max()) == 0.0  # the padding row
        scale = float(table.grad.abs().max())
        assert torch.allclose(got, table.grad, atol=(1e-5 if dtype == torch.float32 else 2e-2) * scale, rtol=1e-5), float((got - table.grad).abs().max())


def test_masked_gradient_from_layernorm_backward_equals_the_separate_dropout_launch():
    """In a pre-LN stack the LayerNorm backward kernel also writes the masked gradient of the dropout site in front of it
    (sc_layernorm_bwd_dz, functional._MaskedGrad) instead of the site re-reading dx in a launch of its own.  Same generator, same
    offsets, dz formed from the rounded dx: every gradient of a 3-layer encoder must come out BIT FOR BIT as with the hand-over off."""
    from speechain_amd.module.transformer.encoder import TransformerEncoder

    def run(fused: bool):
        torch.manual_seed(0)
        Fn.set_compute_dtype(torch.bfloat16)
        Fn.DropoutState.reset(1234)
        Fn.weights.invalidate()
        Fn._MaskedGrad.reset()
        Fn._MaskedGrad.enabled = fused
        hits0 = Fn._MaskedGrad.hits
        enc = TransformerEncoder(input_size=64, d_model=64, num_heads=4, num_layers=3, fdfwd_dim=128, posenc_dropout=0.2, att_dropout=0.2,
                                 fdfwd_dropout=0.2, res_dropout=0.2, layernorm_first=True).cuda().train()
        for p in enc.parameters():
            if p.dim() == 1:
                torch.nn.init.normal_(p, std=0.3)
        x = torch.randn(3, 40, 64, device="cuda", requires_grad=True)
        mask = torch.ones(3, 1, 40, dtype=torch.bool, device="cuda")
        mask[1, :, 31:] = False
        out = enc(x, mask)[0]
        (out.float() * torch.linspace(-1, 1, 64, device="cuda")).sum().backward()
        grads = {k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None}
        grads["input"] = x.grad.detach().clone()
        return grads, Fn._MaskedGrad.hits - hits0

    try:
        g_off, hits_off = run(False)
        g_on, hits_on = run(True)
    finally:
        Fn.set_compute_dtype(None)
        Fn._MaskedGrad.enabled = True
        Fn.weights.invalidate()
    assert hits_off == 0
    assert hits_on >= 5, hits_on  # 3 layers x 2 sites, all but the last feed-forward's... every site whose output feeds a LayerNorm
    assert g_on.keys() == g_off.keys()
    for k in g_on:
        assert torch.equal(g_on[k], g_off[k]), (k, float((g_on[k].float() - g_off[k].float()).abs().max()))

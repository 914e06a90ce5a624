"""Generates the golden fixtures in this directory by IMPORTING AND RUNNING the reference modules.

Run in the build container only (needs /root/reference, never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is stubbed and why:
  * ``torchaudio`` is absent from the image; the reference only needs
    ``torchaudio.functional.melscale_fbanks`` on the forward path (linear2mel.py:135).  A stub module
    provides it from ``oracle/melscale.py`` (restatement of torchaudio's published algorithm) -> the
    mel matrix values are "parity unpinned"; everything downstream of them is the reference's own code.
  * ``GPUtil`` (only imported by utilbox/import_util.py) is stubbed with an empty module.

  * g8 (FeatureNormalization): feat_norm.py:474 creates its batch counter with ``.cuda(device=...)`` even for CPU inputs; this
    container has no GPU, so while g8 runs ``torch.Tensor.cuda`` returns the tensor itself.  Arithmetic is untouched.  The
    module's DDP branch (all_gather over ranks) cannot run here; tests check the 2-rank result against this single-process
    behaviour on the concatenated batch, which is what the reference's gather formulas reduce to.

Outputs are data only (inputs, seeds' products, expected outputs) as ``.npz``; no reference text.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import melscale  # noqa: E402

# ---- stubs -------------------------------------------------------------------------------------
ta = types.ModuleType("torchaudio")
ta.functional = types.ModuleType("torchaudio.functional")
ta.transforms = types.ModuleType("torchaudio.transforms")


def _melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    return torch.from_numpy(melscale.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm, mel_scale))


ta.functional.melscale_fbanks = _melscale_fbanks
sys.modules["torchaudio"] = ta
sys.modules["torchaudio.functional"] = ta.functional
sys.modules["torchaudio.transforms"] = ta.transforms
gp = types.ModuleType("GPUtil")
gp.GPU = object
gp.getGPUs = lambda: []
sys.modules["GPUtil"] = gp

from speechain.module.frontend.speech2mel import Speech2MelSpec  # noqa: E402
from speechain.module.frontend.speech2linear import Speech2LinearSpec  # noqa: E402
from speechain.module.augment.specaug import SpecAugment  # noqa: E402
from speechain.module.transformer.pos_enc import PositionalEncoding  # noqa: E402
from speechain.module.transformer.attention import MultiHeadedAttention  # noqa: E402
from speechain.module.transformer.feed_forward import PositionwiseFeedForward  # noqa: E402
from speechain.module.transformer.encoder import TransformerEncoder, TransformerEncoderLayer  # noqa: E402
from speechain.module.transformer.decoder import TransformerDecoder, TransformerDecoderLayer  # noqa: E402
from speechain.module.postnet.token import TokenPostnet  # noqa: E402
from speechain.module.prenet.embed import EmbedPrenet  # noqa: E402
from speechain.criterion.cross_entropy import CrossEntropy  # noqa: E402
from speechain.criterion.ctc import CTCLoss  # noqa: E402
from speechain.utilbox.train_util import make_mask_from_len  # noqa: E402

torch.set_num_threads(4)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _compact(a):
    """float32 arrays that are exactly representable in float16 (the seeded weights) are stored as float16."""
    if a.dtype == np.float32 and a.size > 64:
        h = a.astype(np.float16)
        if np.array_equal(h.astype(np.float32), a):
            return h
    return a


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: _compact(npy(v)) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  keys={len(arrays)}")


def seeded_state(module, seed, scale=None):
    """Deterministic re-init of every parameter (buffers untouched): N(0, std) with xavier-like std."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() > 1:
                std = (2.0 / (p.size(0) + p.size(1))) ** 0.5 if scale is None else scale
                p.copy_((torch.randn(p.shape, generator=g) * std).half().float())  # fp16-exact: compact fixture
            elif n.endswith("alpha"):
                p.copy_(torch.tensor(1.25))
            elif "layernorm" in n or n.endswith("_ln.weight") or n.endswith("_ln.bias"):
                base = 1.0 if n.endswith("weight") else 0.0
                p.copy_(base + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return {k: v.clone() for k, v in module.state_dict().items()}


# ---- G1 frontend ---------------------------------------------------------------------------------

def g1_frontend():
    g = torch.Generator().manual_seed(101)
    lens = torch.tensor([24000, 19999, 16000, 8001])
    wave = torch.randn(4, 24000, 1, generator=g) * 0.1
    # a decaying-spectrum utterance (tests accuracy on low-power bins) and a silent tail
    t = torch.arange(24000) / 16000.0
    wave[2, :, 0] = 0.3 * torch.sin(2 * np.pi * 220 * t) + 0.01 * torch.sin(2 * np.pi * 3000 * t) + 1e-4 * wave[2, :, 0]
    for i, l in enumerate(lens):
        wave[i, l:] = 0.0
    out = dict(wave=wave, wave_len=lens)
    # ASR preset (config/feat/log_mel/conventional_asr.yaml + recipe frontend conf)
    asr = Speech2MelSpec(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)
    feat, feat_len = asr(wave.clone(), lens.clone())
    out.update(asr_logmel=feat, asr_len=feat_len, asr_melw=asr.linear2mel.mel_fbanks.weight)
    lin = Speech2LinearSpec(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025)
    lfeat, _ = lin(wave.clone(), lens.clone())
    out.update(asr_power=lfeat[:2, :40])  # a slice keeps the fixture small
    # TTS preset (recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:206-217)
    tts = Speech2MelSpec(sr=22050, hop_length=256, win_length=1024, n_fft=1024, n_mels=80, fmin=0, fmax=8000,
                         mag_spec=True, return_energy=True, clamp=1e-5, log_base=None)
    f2, l2, e2, el2 = tts(wave[:3].clone(), lens[:3].clone())
    out.update(tts_logmel=f2, tts_len=l2, tts_energy=e2, tts_melw=tts.linear2mel.mel_fbanks.weight)
    # no-preemphasis, non-default n_fft > win (central window padding), htk / no norm, log base e
    odd = Speech2MelSpec(sr=16000, hop_length=128, win_length=400, n_fft=512, n_mels=40, mel_scale="htk",
                         mel_norm=False, log_base=None)
    f3, l3 = odd(wave[:2].clone(), lens[:2].clone())
    out.update(odd_logmel=f3, odd_len=l3, odd_melw=odd.linear2mel.mel_fbanks.weight)
    save("g1_frontend", **out)


# ---- G2 SpecAugment ------------------------------------------------------------------------------

def _capture_randint():
    """Wraps torch.randint to record every draw the reference makes, in order."""
    rec = []
    orig = torch.randint

    def wrapped(*a, **k):
        r = orig(*a, **k)
        rec.append(r.clone())
        return r

    return rec, orig, wrapped


def g2_specaug():
    g = torch.Generator().manual_seed(202)
    feat = torch.randn(3, 120, 80, generator=g)
    feat_len = torch.tensor([120, 97, 64])
    for i, l in enumerate(feat_len):
        feat[i, l:] = 0.0
    out = dict(feat=feat, feat_len=feat_len)
    cases = dict(
        recipe=dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5),
        default=dict(),
        nowarp_mean=dict(time_warp=False, feat_norm=False, freq_mask_width=[2, 9], time_mask_width=[3, 12]),
        warponly=dict(freq_mask=False, time_mask=False),
        bilinear=dict(time_warp_mode="bilinear", freq_mask_width=15, freq_mask_num=2, time_mask_width=0.05, time_mask_num=2),
    )
    for ci, (name, conf) in enumerate(cases.items()):
        mod = SpecAugment(input_size=80, **conf)
        mod.train()
        torch.manual_seed(1000 + ci)
        rec, orig, wrapped = _capture_randint()
        torch.randint = wrapped
        try:
            y, _ = mod(feat.clone(), feat_len.clone())
        finally:
            torch.randint = orig
        out[f"{name}_out"] = y
        out[f"{name}_seed"] = np.int64(1000 + ci)
        out[f"{name}_ndraw"] = np.int64(len(rec))
        for j, r in enumerate(rec):
            out[f"{name}_draw{j}"] = r
    save("g2_specaug", **out)


# ---- G3 posenc -----------------------------------------------------------------------------------

def g3_posenc():
    out = {}
    for typ in ("mix", "sep"):
        pe = PositionalEncoding(posenc_type=typ, d_model=64, max_len=300)
        out[f"{typ}_rows"] = pe.posenc[0, [0, 1, 2, 17, 150, 299]]
    g = torch.Generator().manual_seed(303)
    x = torch.randn(2, 50, 64, generator=g)
    pe = PositionalEncoding(d_model=64, emb_scale=True, emb_layernorm=True, posenc_scale=True, max_len=300)
    sd = seeded_state(pe, 3)
    pe.eval()
    out.update(x=x, full_out=pe(x.clone()), **{"full_sd." + k: v for k, v in sd.items() if k != "posenc"})
    pe2 = PositionalEncoding(d_model=64, max_len=300)
    pe2.eval()
    out.update(plain_out=pe2(x.clone()))
    save("g3_posenc", **out)


# ---- G4 layers: fwd + grads ----------------------------------------------------------------------

def _grads(outputs, wrt):
    loss = sum((o * torch.linspace(-1, 1, o.numel()).reshape(o.shape)).sum() for o in outputs)
    return torch.autograd.grad(loss, wrt, allow_unused=True)


def g4_layers():
    out = {}
    g = torch.Generator().manual_seed(404)
    # MHA: d=64 h=4 (dh 16), both scale modes, padding mask and full (causal & padding) mask
    for tag, by_head in (("mha_dmodel", False), ("mha_byhead", True)):
        m = MultiHeadedAttention(num_heads=4, d_model=64, dropout=0.0, scale_dp_by_head=by_head)
        sd = seeded_state(m, 41)
        m.eval()
        q = torch.randn(2, 9, 64, generator=g, requires_grad=True)
        kv = torch.randn(2, 13, 64, generator=g, requires_grad=True)
        mask = make_mask_from_len(torch.tensor([13, 8]))
        y, p = m(kv, kv, q, mask)
        gq, gkv, gw = _grads([y], [q, kv, m.q_layer.weight])
        out.update({f"{tag}.q": q, f"{tag}.kv": kv, f"{tag}.mask": mask, f"{tag}.out": y, f"{tag}.probs": p,
                    f"{tag}.gq": gq, f"{tag}.gkv": gkv, f"{tag}.gw_q": gw})
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
    # dh = 192 (FastSpeech2 shape: d=384, h=2), causal self-attention mask (B,T,T)
    m = MultiHeadedAttention(num_heads=2, d_model=384, dropout=0.0)
    sd = seeded_state(m, 42)
    m.eval()
    x = torch.randn(2, 11, 384, generator=g, requires_grad=True)
    pad = make_mask_from_len(torch.tensor([11, 7]))
    full = torch.logical_and(pad.repeat(1, 11, 1), TransformerDecoder.subsequent_mask(2, 11))
    y, p = m(x, x, x, full)
    (gx,) = _grads([y], [x])
    out.update({"mha192.x": x, "mha192.mask": full, "mha192.out": y, "mha192.probs": p, "mha192.gx": gx})
    out.update({f"mha192.sd.{k}": v for k, v in sd.items()})
    # FFN linear (ReLU, GELU) and conv k=3 (ReLU)
    for tag, conf in (("ffn_relu", dict()), ("ffn_gelu", dict(fdfwd_activation="GELU")),
                      ("ffn_conv", dict(fdfwd_type="conv"))):
        f = PositionwiseFeedForward(d_model=64, fdfwd_dim=160, dropout=0.0, **conf)
        sd = seeded_state(f, 43)
        f.eval()
        x = torch.randn(2, 10, 64, generator=g, requires_grad=True)
        y = f(x)
        w_in = f.in_layer.weight if conf.get("fdfwd_type") != "conv" else f.in_layer.conv_lyr.weight
        gx, gw = _grads([y], [x, w_in])
        out.update({f"{tag}.x": x, f"{tag}.out": y, f"{tag}.gx": gx, f"{tag}.gw_in": gw})
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
    # encoder layer (pre-LN and post-LN) with grads wrt input and LN weight
    for tag, first in (("enc_pre", True), ("enc_post", False)):
        e = TransformerEncoderLayer(d_model=64, num_heads=4, fdfwd_dim=128, att_dropout=0.0, fdfwd_dropout=0.0,
                                    res_dropout=0.0, fdfwd_activation="GELU", layernorm_first=first)
        sd = seeded_state(e, 44)
        e.eval()
        x = torch.randn(2, 12, 64, generator=g, requires_grad=True)
        mask = make_mask_from_len(torch.tensor([12, 5]))
        y, p = e(x, mask)
        gx, gln, gb = _grads([y], [x, e.att_layernorm.weight, e.multihead_att.output_layer.bias])
        out.update({f"{tag}.x": x, f"{tag}.mask": mask, f"{tag}.out": y, f"{tag}.probs": p, f"{tag}.gx": gx,
                    f"{tag}.g_attln_w": gln, f"{tag}.g_outb": gb})
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
    # decoder layer
    d = TransformerDecoderLayer(d_model=64, num_heads=4, fdfwd_dim=128, att_dropout=0.0, fdfwd_dropout=0.0,
                                res_dropout=0.0, fdfwd_activation="GELU", scale_dp_by_head=True)
    sd = seeded_state(d, 45)
    d.eval()
    tgt = torch.randn(2, 7, 64, generator=g, requires_grad=True)
    src = torch.randn(2, 12, 64, generator=g, requires_grad=True)
    tpad = make_mask_from_len(torch.tensor([7, 4]))
    tmask = torch.logical_and(tpad.repeat(1, 7, 1), TransformerDecoder.subsequent_mask(2, 7))
    smask = make_mask_from_len(torch.tensor([12, 9]))
    y, sp, ep = d(tgt, src, tmask, smask)
    gt, gs = _grads([y], [tgt, src])
    out.update({"dec.tgt": tgt, "dec.src": src, "dec.tmask": tmask, "dec.smask": smask, "dec.out": y,
                "dec.self_probs": sp, "dec.encdec_probs": ep, "dec.gtgt": gt, "dec.gsrc": gs})
    out.update({f"dec.sd.{k}": v for k, v in sd.items()})
    save("g4_layers", **out)


# ---- G5 stacks -----------------------------------------------------------------------------------

def g5_stacks():
    out = {}
    g = torch.Generator().manual_seed(505)
    enc = TransformerEncoder(d_model=64, num_heads=4, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0,
                             fdfwd_dropout=0.0, res_dropout=0.0, fdfwd_activation="GELU", posenc_maxlen=200)
    sd = seeded_state(enc, 51)
    enc.eval()
    x = torch.randn(2, 20, 64, generator=g)
    mask = make_mask_from_len(torch.tensor([20, 13]))
    y, m2, att, hid = enc(x.clone(), mask)
    out.update({"enc.x": x, "enc.mask": mask, "enc.out": y, "enc.att0": att[0], "enc.att1": att[1],
                "enc.hid0": hid[0], "enc.hid1": hid[1]})
    out.update({f"enc.sd.{k}": v for k, v in sd.items() if not k.endswith("posenc.posenc")})
    # TTS-flavoured encoder: conv FFN, posenc alpha, emb LN, uni-directional
    enc2 = TransformerEncoder(d_model=48, num_heads=2, num_layers=2, fdfwd_dim=96, fdfwd_type="conv", posenc_dropout=0.0,
                              att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0, posenc_scale=True, emb_layernorm=True,
                              scale_dp_by_head=True, uni_direction=True, layernorm_first=False, posenc_maxlen=200)
    sd = seeded_state(enc2, 52)
    enc2.eval()
    x2 = torch.randn(2, 15, 48, generator=g)
    mask2 = make_mask_from_len(torch.tensor([15, 9]))
    y2, m22, att2, hid2 = enc2(x2.clone(), mask2)
    out.update({"enc2.x": x2, "enc2.mask": mask2, "enc2.out": y2, "enc2.mask_out": m22, "enc2.att1": att2[1]})
    out.update({f"enc2.sd.{k}": v for k, v in sd.items() if not k.endswith("posenc.posenc")})
    dec = TransformerDecoder(d_model=64, num_heads=4, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0,
                             fdfwd_dropout=0.0, res_dropout=0.0, fdfwd_activation="GELU", emb_scale=False,
                             posenc_maxlen=200)
    sd = seeded_state(dec, 53)
    dec.eval()
    tgt = torch.randn(2, 9, 64, generator=g)
    tmask = make_mask_from_len(torch.tensor([9, 6]))
    yd, sa, ea, hd = dec(tgt.clone(), y.detach(), tmask, mask)
    out.update({"dec.tgt": tgt, "dec.tmask": tmask, "dec.out": yd, "dec.self_att1": sa[1], "dec.encdec_att1": ea[1],
                "dec.hid0": hd[0]})
    out.update({f"dec.sd.{k}": v for k, v in sd.items() if not k.endswith("posenc.posenc")})
    save("g5_stacks", **out)


# ---- G6 criteria ---------------------------------------------------------------------------------

def g6_losses():
    out = {}
    g = torch.Generator().manual_seed(606)
    V = 37
    logits = (torch.randn(3, 8, V, generator=g) * 2).requires_grad_(True)
    text = torch.randint(1, V - 2, (3, 9), generator=g)
    text_len = torch.tensor([9, 6, 4])
    for i, l in enumerate(text_len):
        text[i, 0] = V - 1
        text[i, l - 1] = V - 1
        text[i, l:] = 0
    out.update(ce_logits=logits, text=text, text_len=text_len)
    for tag, conf in (("ce0", dict()), ("ce02", dict(label_smoothing=0.2)),
                      ("ce02n", dict(label_smoothing=0.2, length_normalized=True))):
        loss = CrossEntropy(**conf)(logits=logits, text=text, text_len=text_len)
        (gl,) = torch.autograd.grad(loss, logits)
        out.update({f"{tag}.loss": loss, f"{tag}.grad": gl})
    ctc_logits = (torch.randn(3, 21, V, generator=g) * 2).requires_grad_(True)
    enc_len = torch.tensor([21, 15, 3])  # last one is infeasible (3 frames < 2 labels w/ repeat?) -> may be inf
    text2 = text.clone()
    text2[1, 1:5] = torch.tensor([5, 5, 7, 7])  # repeated labels exercise the skip rule
    loss = CTCLoss()(ctc_logits, enc_len, text2.clone(), text_len.clone())
    (gl,) = torch.autograd.grad(loss, ctc_logits)
    per_utt = torch.nn.functional.ctc_loss(
        ctc_logits.transpose(0, 1).log_softmax(-1),
        torch.cat([text2[i, 1: text_len[i] - 1] for i in range(3)]), enc_len, text_len - 2, blank=0, reduction="none",
        zero_infinity=True)
    out.update(ctc_logits=ctc_logits, enc_len=enc_len, ctc_text=text2, ctc_loss=loss, ctc_grad=gl, ctc_per_utt=per_utt)
    save("g6_losses", **out)


# ---- G7 harness ----------------------------------------------------------------------------------

def g7_harness():
    g = torch.Generator().manual_seed(707)
    d, H, V = 64, 4, 53
    enc = TransformerEncoder(d_model=d, num_heads=H, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0,
                             fdfwd_dropout=0.0, res_dropout=0.0, fdfwd_activation="GELU", posenc_maxlen=200)
    emb = EmbedPrenet(embedding_dim=d, vocab_size=V)
    dec = TransformerDecoder(d_model=d, num_heads=H, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0,
                             fdfwd_dropout=0.0, res_dropout=0.0, fdfwd_activation="GELU", emb_scale=False,
                             posenc_maxlen=200)
    post, ctc_layer = TokenPostnet(input_size=d, vocab_size=V), TokenPostnet(input_size=d, vocab_size=V)
    P = {}
    for name, mod, seed in (("encoder", enc, 71), ("embed_mod", emb, 72), ("decoder", dec, 73), ("postnet", post, 74),
                            ("ctc_layer", ctc_layer, 75)):
        sd = seeded_state(mod, seed)
        mod.eval()
        for k, v in sd.items():
            P[f"{name}.{k}"] = v
    P["embed.weight"] = P.pop("embed_mod.embed.weight")
    feat = torch.randn(3, 30, d, generator=g)
    feat_len = torch.tensor([30, 22, 17])
    text = torch.randint(1, V - 2, (3, 8), generator=g)
    text_len = torch.tensor([8, 6, 5])
    for i, l in enumerate(text_len):
        text[i, 0] = V - 1
        text[i, l - 1] = V - 1
        text[i, l:] = 0
    # sequencing of model/ar_asr.py:386-388 (eos strip), encoder/asr.py:126-131, decoder/ar_asr.py:49-78, :453-458, :550
    dtext, dlen = text.clone(), text_len - 1
    for i in range(3):
        dtext[i, text_len[i] - 1] = 0
    dtext = dtext[:, :-1]
    fmask = make_mask_from_len(feat_len)
    e, fmask, _, _ = enc(feat.clone(), fmask)
    tmask = make_mask_from_len(dlen)
    dd, _, _, _ = dec(src=e, src_mask=fmask, tgt=emb(dtext), tgt_mask=tmask)
    logits, ctc_logits = post(dd), ctc_layer(e)
    ce = CrossEntropy(label_smoothing=0.2)(logits=logits, text=text, text_len=text_len)
    ctc = CTCLoss()(ctc_logits, feat_len.clone(), text.clone(), text_len.clone())
    loss = 0.7 * ce + 0.3 * ctc
    gsel = torch.autograd.grad(loss, [enc.trfm_layers[0].multihead_att.q_layer.weight, emb.embed.weight,
                                      dec.trfm_layers[1].encdec_att.k_layer.bias, ctc_layer.linear.weight])
    out = dict(feat=feat, feat_len=feat_len, text=text, text_len=text_len, enc=e, logits=logits, ctc_logits=ctc_logits,
               ce=ce, ctc=ctc, loss=loss, g_enc0_q=gsel[0], g_embed=gsel[1], g_dec1_encdec_kb=gsel[2], g_ctc_w=gsel[3])
    out.update({f"sd.{k}": v for k, v in P.items() if not k.endswith("posenc.posenc")})
    save("g7_harness", **out)


def g8_featnorm():
    """FeatureNormalization (module/norm/feat_norm.py:78-338, 452-509): every norm_type, training sequences with the running
    statistics (epoch gating, epoch=None), evaluation, mean-only / std-only, 2-D input, tensor / int group ids."""
    from speechain.module.norm.feat_norm import FeatureNormalization

    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        gen = torch.Generator().manual_seed(80)
        B, T, F = 4, 13, 5
        out = {}
        batches = []
        for j in range(5):
            feat = torch.randn(B, T, F, generator=gen) * (1.0 + 0.5 * j) + 0.3 * j
            flen = torch.tensor([13, 9, 2, 6]) if j % 2 == 0 else torch.tensor([7, 13, 11, 3])
            for i in range(B):
                feat[i, flen[i]:] = 0
            batches.append((feat, flen))
            out[f"feat{j}"], out[f"len{j}"] = feat, flen

        def bufs(m, tag):
            for k, v in m.named_buffers():
                out[f"{tag}.buf.{k}"] = v.clone().float() if v.dtype != torch.int64 else v.clone()

        # utterance / batch
        m = FeatureNormalization(input_size=F, norm_type="utterance").train()
        out["utt.y"] = m(batches[0][0].clone(), batches[0][1].clone())[0]
        m = FeatureNormalization(input_size=F, norm_type="batch").train()
        out["batch.train.y"] = m(batches[0][0].clone(), batches[0][1].clone())[0]
        out["batch.eval.y"] = m.eval()(batches[1][0].clone(), batches[1][1].clone())[0]
        # global (the recipes' `normalize: True`): epochs 1,1,2 accumulate, epoch 5 is frozen (max_epoch_num = 4), then eval
        m = FeatureNormalization(input_size=F).train()
        for step, (j, ep) in enumerate([(0, 1), (1, 1), (2, 2), (3, 5)]):
            out[f"global.s{step}.y"] = m(batches[j][0].clone(), batches[j][1].clone(), epoch=ep)[0]
            bufs(m, f"global.s{step}")
        out["global.eval.y"] = m.eval()(batches[4][0].clone(), batches[4][1].clone())[0]
        out["global.recover"] = m.recover(out["global.eval.y"].clone())
        sd = m.state_dict()
        out["global.sd_keys"] = np.array(sorted(sd.keys()))
        # epoch=None: the counter restarts at every call
        m = FeatureNormalization(input_size=F).train()
        for step in range(2):
            out[f"global_noepoch.s{step}.y"] = m(batches[step][0].clone(), batches[step][1].clone())[0]
            bufs(m, f"global_noepoch.s{step}")
        # mean only / std only
        m = FeatureNormalization(input_size=F, std_norm=False).train()
        out["meanonly.y"] = m(batches[0][0].clone(), batches[0][1].clone(), epoch=1)[0]
        # (mean_norm=False with the global / group types raises in the reference itself: feat_norm.py:475 reads
        #  curr_aver_mean.device of None -- so std-only is recorded for the utterance type, where it works)
        m = FeatureNormalization(input_size=F, norm_type="utterance", mean_norm=False, clamp=0.5).train()
        out["stdonly.y"] = m(batches[0][0].clone(), batches[0][1].clone())[0]
        # 2-D input
        m = FeatureNormalization(input_size=1, norm_type="utterance").train()
        out["twod.y"] = m(batches[0][0][:, :, 0].clone(), batches[0][1].clone())[0]
        # group: one id for the whole batch (int), then per-utterance ids, then an unseen id at evaluation
        m = FeatureNormalization(input_size=F, norm_type="group").train()
        out["group_int.s0.y"] = m(batches[0][0].clone(), batches[0][1].clone(), group_ids=7, epoch=1)[0]
        out["group_int.s1.y"] = m(batches[1][0].clone(), batches[1][1].clone(), group_ids=7, epoch=1)[0]
        bufs(m, "group_int.s1")
        m = FeatureNormalization(input_size=F, norm_type="group").train()
        ids0, ids1 = torch.tensor([3, 5, 3, 5]), torch.tensor([5, 5, 3, 9])
        out["group_ids0"], out["group_ids1"] = ids0, ids1
        out["group_t.s0.y"] = m(batches[0][0].clone(), batches[0][1].clone(), group_ids=ids0, epoch=1)[0]
        out["group_t.s1.y"] = m(batches[1][0].clone(), batches[1][1].clone(), group_ids=ids1, epoch=1)[0]
        bufs(m, "group_t.s1")
        ids2 = torch.tensor([3, 11, 9, 5])  # 11 was never seen: average statistics
        out["group_ids2"] = ids2
        out["group_t.eval.y"] = m.eval()(batches[2][0].clone(), batches[2][1].clone(), group_ids=ids2)[0]
    finally:
        torch.Tensor.cuda = real_cuda
    save("g8_featnorm", **out)


# ---- G9 Conv2dPrenet (row f2) --------------------------------------------------------------------------------------------

def g9_prenet():
    """speechain/module/prenet/conv2d.py + linear.py run as imported: training-mode output, lengths, gradients of every parameter,
    the BatchNorm running statistics after the step, then the eval-mode output with those statistics."""
    from speechain.module.prenet.conv2d import Conv2dPrenet
    cases = dict(
        bn=(dict(conv_dims=[8, 16], conv_kernel=3, conv_stride=2, conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=32), (3, 37, 20)),
        relu=(dict(conv_dims=[8, 8], lnr_dims=[24, 16], lnr_activation="ReLU"), (3, 37, 20)),
        asym=(dict(conv_dims=[8], conv_kernel=[3, 5], conv_stride=[2, 1], conv_batchnorm=True, lnr_dims=None), (2, 21, 20)),
        zc=(dict(conv_dims=[8, 8], conv_activation="ReLU", lnr_dims=None, zero_centered=True), (2, 21, 20)),
        narrow=(dict(conv_dims=[4, 4], conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=8), (2, 5, 20)),
    )
    out = {}
    g = torch.Generator().manual_seed(909)
    for tag, (conf, (B, T, Fd)) in cases.items():
        m = Conv2dPrenet(input_size=Fd, **conf)
        seeded_state(m, 91)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.copy_(1 + 0.1 * torch.randn(mod.weight.shape, generator=g))
                    mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                    mod.running_mean.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                    mod.running_var.copy_(1 + 0.2 * torch.rand(mod.bias.shape, generator=g))
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.train()
        x = torch.randn(B, T, Fd, generator=g) * 1.5 - 0.7
        lens = torch.tensor([T] + [max(T - 5 * (i + 1), 1) for i in range(B - 1)])
        y, ylen = m(x.clone(), lens.clone())
        names, params = zip(*m.named_parameters())
        grads = _grads([y], list(params))
        out.update({f"{tag}.x": x, f"{tag}.len": lens, f"{tag}.out": y, f"{tag}.out_len": ylen, f"{tag}.output_size": m.output_size})
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
        out.update({f"{tag}.grad.{n}": gr for n, gr in zip(names, grads)})
        out.update({f"{tag}.after.{k}": v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
        m.eval()
        ye, _ = m(x.clone(), lens.clone())
        out[f"{tag}.out_eval"] = ye
    save("g9_prenet", **out)

# ---- G10 FastSpeech2 (BASELINE configs[4]) ---------------------------------------------------------------

TTS_SMALL = dict(d=32, heads=2, layers=2, ffn=64, pred=16, post=16, vocab=20)


def tts_confs(c=TTS_SMALL, dropout=0.0):
    trf = dict(posenc_dropout=dropout, posenc_scale=True, emb_scale=False, emb_layernorm=True, d_model=c["d"], num_heads=c["heads"],
               num_layers=c["layers"], att_dropout=dropout, fdfwd_dim=c["ffn"], fdfwd_type="conv", fdfwd_args=dict(kernel_size=3),
               fdfwd_dropout=dropout, res_dropout=dropout, layernorm_first=True, posenc_maxlen=400)
    enc = dict(vocab_size=c["vocab"], embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=c["d"])),
               encoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf)))
    dec = dict(
        decoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf)),
        postnet=dict(type="postnet.conv1d.Conv1dPostnet", conf=dict(conv_dims=[c["post"], -1, 0], conv_kernel=5, conv_batchnorm=True,
                                                                     conv_activation="Tanh", conv_dropout=dropout)),
        pitch_predictor=dict(type="prenet.var_pred.Conv1dVarPredictor", conf=dict(conv_dims=[c["pred"], -1, -1], conv_kernel=5, conv_dropout=dropout,
                                                                                 conv_emb_kernel=1, conv_emb_dropout=0.0)),
        energy_predictor=dict(type="prenet.var_pred.Conv1dVarPredictor", conf=dict(conv_dims=[c["pred"], -1], conv_kernel=3, conv_dropout=dropout,
                                                                                  conv_emb_kernel=1, conv_emb_dropout=0.0)),
        duration_predictor=dict(type="prenet.var_pred.Conv1dVarPredictor", conf=dict(conv_dims=[c["pred"], -1], conv_kernel=3, conv_dropout=dropout)),
        feat_frontend=dict(type="frontend.speech2mel.Speech2MelSpec", conf=dict(sr=22050, mag_spec=True, hop_length=256, win_length=1024, n_mels=80,
                                                                                fmin=0, fmax=8000, log_base=None, clamp=1e-5)),
        feat_normalize=True, pitch_normalize=True, energy_normalize=True)
    return enc, dec


def g10_tts():
    """FastSpeech2 (recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml at small sizes): TTSEncoder (module/encoder/tts.py) ->
    FastSpeech2Decoder (module/decoder/nar_tts.py: variance predictors, duration handling, average_scalar_by_duration, length
    regulation, decoder stack, feat_pred, Conv1dPostnet) -> the five LeastError terms of model/nar_tts.py:472-560.  Training mode, all
    dropout rates 0, two consecutive steps (BatchNorm running statistics, the global feature / pitch / energy normalisation buffers)."""
    import copy

    from speechain.criterion.least_error import LeastError
    from speechain.module.decoder.nar_tts import FastSpeech2Decoder
    from speechain.module.encoder.tts import TTSEncoder

    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        gen = torch.Generator().manual_seed(1010)
        ec, dc = tts_confs()
        enc = TTSEncoder(**copy.deepcopy(ec))
        dec = FastSpeech2Decoder(input_size=TTS_SMALL["d"], **copy.deepcopy(dc))
        P = {}
        for name, mod, seed in (("encoder", enc, 101), ("decoder", dec, 102)):
            sd = seeded_state(mod, seed)
            mod.train()
            for k, v in sd.items():
                P[f"{name}.{k}"] = v
        out = {f"sd0.{k}": v for k, v in P.items() if not k.endswith("posenc.posenc") and "mel_fbanks" not in k and "preemph" not in k}
        B, L = 3, 7
        text_len = torch.tensor([7, 6, 4])
        wave_len = torch.tensor([14000, 11000, 8200])
        for step in range(2):
            text = torch.randint(1, TTS_SMALL["vocab"], (B, L), generator=gen)
            dur = (torch.rand(B, L, generator=gen) * 0.2 + 0.02).half().float()
            for i in range(B):
                text[i, text_len[i]:] = 0
                dur[i, text_len[i]:] = 0
            dur[1, 2] = 0.0  # a token without frames
            wave = (torch.randn(B, int(wave_len.max()), 1, generator=gen) * 0.1).half().float()
            for i in range(B):
                wave[i, wave_len[i]:] = 0
            T = int(wave_len.max()) // 256 + 1
            pitch = (torch.rand(B, T, generator=gen) * 200 + 80).half().float()
            pitch_len = wave_len // 256 + 1
            for i in range(B):
                pitch[i, pitch_len[i]:] = 0
            e_text, e_mask, _, _ = enc(text=text.clone(), text_len=text_len.clone())
            r = dec(enc_text=e_text, enc_text_mask=e_mask, duration=dur.clone(), duration_len=text_len.clone(), pitch=pitch.clone(),
                    pitch_len=pitch_len.clone(), feat=wave.clone(), feat_len=wave_len.clone(), epoch=1, min_frame_num=0)  # (model/nar_tts.py:256: 0)
            (before, after, pred_len, tgt_feat, tgt_feat_len, pred_pitch, tgt_pitch, tgt_pitch_len, pred_energy, tgt_energy, tgt_energy_len,
             pred_dur, pred_gate, used_dur, used_dur_len, _, _) = r
            assert pred_gate is None
            l1, l2 = LeastError(loss_type="L1"), LeastError(loss_type="L2")
            terms = dict(feat_before=l1(pred=before, tgt=tgt_feat, tgt_len=tgt_feat_len), feat_after=l1(pred=after, tgt=tgt_feat, tgt_len=tgt_feat_len),
                         pitch=l2(pred=pred_pitch, tgt=tgt_pitch, tgt_len=tgt_pitch_len), energy=l2(pred=pred_energy, tgt=tgt_energy, tgt_len=tgt_energy_len),
                         duration=l2(pred=pred_dur, tgt=torch.log(used_dur.float() + 1), tgt_len=used_dur_len))
            loss = sum(terms.values())
            sel = {"g_emb": enc.embedding.embed.weight, "g_enc0_ffn_in": enc.encoder.trfm_layers[0].feed_forward.in_layer.conv_lyr.weight,
                   "g_dur_conv0": dec.duration_predictor.conv[0].conv_lyr.weight, "g_dur_ln0": dec.duration_predictor.conv[2].weight,
                   "g_dur_lin": dec.duration_predictor.linear.weight, "g_pitch_emb_w": dec.pitch_predictor.conv_emb[0].conv_lyr.weight,
                   "g_pitch_emb_b": dec.pitch_predictor.conv_emb[0].conv_lyr.bias, "g_energy_conv1": dec.energy_predictor.conv[4].conv_lyr.weight,
                   "g_dec1_q": dec.decoder.trfm_layers[1].multihead_att.q_layer.weight, "g_dec_alpha": dec.decoder.posenc.alpha,
                   "g_feat_pred": dec.feat_pred.weight, "g_post0": dec.postnet.conv[0].conv_lyr.weight, "g_post_bn1_w": dec.postnet.conv[1].weight,
                   "g_post_last": dec.postnet.conv[8].conv_lyr.weight, "g_post_bn_last_b": dec.postnet.conv[9].bias}
            grads = torch.autograd.grad(loss, list(sel.values()))
            tag = f"s{step}."
            out.update({tag + "text": text, tag + "dur": dur, tag + "wave": wave, tag + "pitch": pitch, tag + "enc_text": e_text,
                        tag + "before": before, tag + "after": after, tag + "pred_len": pred_len, tag + "tgt_feat": tgt_feat,
                        tag + "tgt_feat_len": tgt_feat_len, tag + "pred_pitch": pred_pitch, tag + "tgt_pitch": tgt_pitch,
                        tag + "pred_energy": pred_energy, tag + "tgt_energy": tgt_energy, tag + "pred_dur": pred_dur, tag + "used_dur": used_dur,
                        tag + "loss": loss})
            out.update({tag + "loss_" + k: v for k, v in terms.items()})
            out.update({tag + k: v for k, v in zip(sel.keys(), grads)})
            if step == 0:  # a plain SGD move between the two steps, so that the second step sees other weights
                with torch.no_grad():
                    for prm, gr in zip(sel.values(), grads):
                        prm -= 0.05 * gr
        out["text_len"], out["wave_len"], out["pitch_len"] = text_len, wave_len, pitch_len
        for name, mod in (("encoder", enc), ("decoder", dec)):
            for k, v in mod.named_buffers():
                if not k.endswith("posenc.posenc") and "mel_fbanks" not in k and "preemph" not in k:
                    out[f"buf1.{name}.{k}"] = v.clone().float() if v.dtype.is_floating_point else v.clone()
        # the two stand-alone helpers on their own data (tokens without frames, durations overshooting the frames)
        fs = torch.randn(2, 11, generator=gen)
        du = torch.tensor([[3.0, 0.0, 4.0, 2.0, 5.0], [1.0, 1.0, 0.0, 0.0, 0.0]])
        ts, _ = FastSpeech2Decoder.average_scalar_by_duration(fs, du, torch.tensor([5, 2]))
        out.update(avg_frame=fs, avg_dur=du, avg_out=ts)
        save("g10_tts", **out)
    finally:
        torch.Tensor.cuda = real_cuda


# ---- G11 inference callers: beam search, CTC prefix scorer, prefix decoding ------------------------------------

DECODE_SMALL = dict(d_model=128, num_heads=2, dec_layers=2, fdfwd_dim=128, vocab_size=23, activation="GELU")  # heads of 64: the cached kernels' size


def g11_decode():
    """beam_searching (infer_func/beam_search.py:106) driven by the reference's own ARASRDecoder (module/decoder/ar_asr.py) and a
    TokenPostnet CTC head, three argument sets; CTCPrefixScorer (infer_func/ctc_decoding.py) on its own data for three steps.
    ``speechain.model.lm`` (only a type annotation in beam_search.py:14,128) drags the whole model package in, whose dependencies
    (editdistance, ruamel.yaml ...) are absent: a stub module with an empty ``LM`` class stands in for the import."""
    lm = types.ModuleType("speechain.model.lm")
    lm.LM = type("LM", (), {})
    sys.modules.setdefault("speechain.model", types.ModuleType("speechain.model"))
    sys.modules["speechain.model.lm"] = lm
    from speechain.infer_func.beam_search import beam_searching
    from speechain.infer_func.ctc_decoding import CTCPrefixScorer
    from speechain.module.decoder.ar_asr import ARASRDecoder

    c = DECODE_SMALL
    d, V = c["d_model"], c["vocab_size"]
    g = torch.Generator().manual_seed(1111)
    dec = ARASRDecoder(vocab_size=V, embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=d)),
                       decoder=dict(type="transformer.decoder.TransformerDecoder",
                                    conf=dict(num_heads=c["num_heads"], num_layers=c["dec_layers"], fdfwd_dim=c["fdfwd_dim"],
                                              fdfwd_activation=c["activation"], posenc_dropout=0.0, att_dropout=0.0, fdfwd_dropout=0.0,
                                              res_dropout=0.0, emb_scale=False, posenc_maxlen=100)))
    ctc_layer = TokenPostnet(input_size=d, vocab_size=V)
    sd = seeded_state(dec, 111, scale=0.18)
    sd_ctc = seeded_state(ctc_layer, 112, scale=0.25)
    with torch.no_grad():
        dec.postnet.linear.bias[V - 1] += 1.5  # <sos/eos> likely enough for some hypotheses to end before the length limit
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    dec.eval()
    ctc_layer.eval()
    B, T = 3, 14
    enc = torch.randn(B, T, d, generator=g)
    enc_len = torch.tensor([14, 11, 8])
    mask = make_mask_from_len(enc_len)
    out = dict(enc=enc, enc_len=enc_len)
    out.update({f"sd.decoder.{k}": v for k, v in sd.items() if not k.endswith("posenc.posenc")})
    out.update({f"sd.ctc_layer.{k}": v for k, v in sd_ctc.items()})
    cases = dict(a=dict(beam_size=4, min_f2t_ratio=1.5),
                 b=dict(beam_size=4, min_f2t_ratio=-8, ctc_weight=0.3, temperature=1.3, length_penalty=1.2, ctc_temperature=1.1),
                 c=dict(beam_size=3, min_f2t_ratio=2.0, eos_filtering=True, eos_threshold=1.2),
                 d=dict(beam_size=2, min_f2t_ratio=-9, ctc_weight=0.5))
    with torch.no_grad():
        for name, kw in cases.items():
            res = beam_searching(enc_feat=enc.clone(), enc_feat_mask=mask.clone(), asr_decode_fn=dec, ctc_decode_fn=ctc_layer,
                                 vocab_size=V, sos_eos=V - 1, padding_idx=0, **kw)
            for k, v in res.items():
                out[f"{name}.{k}"] = v
        # prefix logits of one fixed batch of prefixes (what a cached decode step must reproduce position by position)
        text = torch.randint(1, V - 1, (B, 7), generator=g)
        text[:, 0] = V - 1
        text_len = torch.tensor([7, 5, 3])
        for i, l in enumerate(text_len):
            text[i, l:] = 0
        out.update(pref_text=text, pref_text_len=text_len, pref_logits=dec(enc, mask, text, text_len)[0])
        # CTCPrefixScorer alone: 2 sentences x 3 beams, 9 tokens, three steps with a fixed choice of parents and tokens
        Bc, beam, Tc, Vc = 2, 3, 7, 9
        x = torch.log_softmax(torch.randn(Bc, Tc, Vc, generator=g) * 1.5, dim=-1).repeat_interleave(beam, dim=0)
        lens = torch.tensor([7, 7, 7, 5, 5, 5])
        sc = CTCPrefixScorer(x=x.clone(), enc_lens=lens, batch_size=Bc, beam_size=beam, blank_index=0, eos_index=Vc - 1)
        out.update(ctc_x=x, ctc_lens=lens)
        gtext = torch.zeros(Bc * beam, 0, dtype=torch.long)
        state = None
        picks = [(torch.tensor([0, 0, 1, 3, 4, 4]), torch.tensor([3, 5, 2, 1, 4, 7])),
                 (torch.tensor([2, 0, 1, 5, 3, 3]), torch.tensor([5, 6, 2, 4, 4, 1])),
                 (torch.tensor([1, 1, 0, 4, 5, 3]), torch.tensor([1, 5, 7, 2, 6, 6]))]
        for step, (beam_idx, tokens) in enumerate(picks):
            psi, state = sc.forward_step(gtext, state)
            out[f"ctc_psi{step}"] = psi
            gtext = torch.cat([gtext[beam_idx], tokens[:, None]], dim=1)
            token_idx = tokens.view(Bc, beam) + (torch.arange(beam) * Vc)[None, :]
            state = sc.permute_mem(state, beam_idx, token_idx)
            out[f"ctc_r{step}"], out[f"ctc_psimem{step}"] = state[0], state[1][:, 0]
            out[f"ctc_beam_idx{step}"], out[f"ctc_tokens{step}"] = beam_idx, tokens
        psi, _ = sc.forward_step(gtext, state)
        out["ctc_psi3"] = psi
    save("g11_decode", **out)


# ---- G12 recover(): mel -> linear (reference, deterministic) and the post-processing around Griffin-Lim -----------------------------

def g12_recover():
    """LinearSpec2MelSpec.recover (linear2mel.py:191-224) as the reference computes it, and Speech2LinearSpec.recover
    (speech2linear.py:252-351) with its ``griffin_lim`` member -- torchaudio.transforms.GriffinLim, absent here -- replaced by a function
    that returns a fixed waveform: pins the length rule, the sliding-window inverse pre-emphasis and the silencing of the tails."""
    from speechain.module.frontend.linear2mel import LinearSpec2MelSpec

    g = torch.Generator().manual_seed(1212)
    l2m = LinearSpec2MelSpec(n_fft=64, n_mels=12, sr=16000)
    mel = torch.randn(3, 9, 12, generator=g) * 0.8 - 1.0
    mel_len = torch.tensor([9, 6, 9])
    lin = l2m.recover(mel.clone(), mel_len)
    l2m_ln = LinearSpec2MelSpec(n_fft=64, n_mels=12, sr=16000, log_base=None)
    lin_ln = l2m_ln.recover(mel.clone(), mel_len)
    s2l = Speech2LinearSpec(hop_length=16, win_length=64, sr=16000, preemphasis=0.97)
    T = 11
    fixed = torch.randn(2, 16 * (T - 1), generator=g) * 0.1
    s2l.griffin_lim = lambda spec: fixed.clone()
    flen = torch.tensor([T, 7])
    wav, wav_len = s2l.recover(torch.rand(2, T, 33, generator=g), flen, inv_preemph_winlen=20)
    save("g12_recover", mel=mel, mel_len=mel_len, melw=l2m.mel_fbanks.weight, lin=lin, lin_ln=lin_ln, gl_wav=fixed, feat_len=flen, wav=wav,
         wav_len=wav_len)


# ---- G13 CrossEntropy options: token re-weighting and confidence thresholding (cross_entropy.py:81-95,148-199) ----------------------

def g13_ce_options():
    out = {}
    g = torch.Generator().manual_seed(1313)
    V = 29
    logits = (torch.randn(4, 7, V, generator=g) * 1.5).requires_grad_(True)
    text = torch.randint(1, V - 2, (4, 8), generator=g)
    text_len = torch.tensor([8, 6, 8, 3])
    for i, l in enumerate(text_len):
        text[i, 0] = V - 1
        text[i, l - 1] = V - 1
        text[i, l:] = 0
    weights = torch.ones(V)
    weights[3], weights[11], weights[V - 1] = 2.5, 0.25, 0.0
    out.update(logits=logits, text=text, text_len=text_len, weights=weights)
    cases = dict(w=dict(label_smoothing=0.1), tok=dict(confid_threshold=0.05, confid_level="token", length_normalized=True),
                 sent=dict(confid_threshold=0.02, confid_level="sentence"), tok_ls=dict(confid_threshold=0.08, confid_level="token", label_smoothing=0.2))
    for tag, conf in cases.items():
        crit = CrossEntropy(**conf)
        if tag == "w":
            crit.token_weights = weights  # (criterion_init builds it from a vocabulary file and .cuda(): cross_entropy.py:85-95)
        loss = crit(logits=logits, text=text.clone(), text_len=text_len.clone())
        (gl,) = torch.autograd.grad(loss, logits)
        out.update({f"{tag}.loss": loss, f"{tag}.grad": gl})
    save("g13_ce_options", **out)


# ---- G14 speaker embeddings (module/prenet/spk_embed.py:7-230; FastSpeech2Decoder.forward step 1, decoder/nar_tts.py:257-263) -----------
SPK_CASES = dict(
    # the VCTK recipes (recipes/tts/vctk/exp_cfg/16khz_ecapa_mfa_fastspeech2_mic2.yaml:228-230): pretrained vectors, concatenated
    pre_cat=dict(d_model=48, spk_emb_dim_pretrained=24, spk_emb_comb="concat"),
    # both sources added after their pre-projections (dims differ from d_model), decoder input combined by the same projection
    both_add=dict(d_model=48, spk_emb_dim_lookup=16, spk_emb_dim_pretrained=24, spk_num=7, spk_emb_comb="add", dec_comb=True),
    # lookup table only, concatenated, a projection of its own for the decoder input
    look_cat=dict(d_model=48, spk_emb_dim_lookup=16, spk_num=7, spk_emb_comb="concat", dec_comb=True, encdec_same_proj=False),
    # added without pre-projection (dims equal d_model)
    pre_add_same=dict(d_model=48, spk_emb_dim_pretrained=48, spk_emb_comb="add"),
)


def g14_spk():
    from speechain.module.prenet.spk_embed import SpeakerEmbedPrenet

    g = torch.Generator().manual_seed(140)
    B, L, T = 3, 9, 13
    out = {}
    for i, (tag, conf) in enumerate(SPK_CASES.items()):
        m = SpeakerEmbedPrenet(**conf)
        sd = seeded_state(m, 141 + i)
        spk_ids = torch.tensor([3, 1, 6]) if m.use_lookup else None
        spk_feat = torch.randn(B, conf["spk_emb_dim_pretrained"], generator=g) * 2.0 if m.use_pretrain else None
        if tag == "both_add":
            spk_feat = spk_feat.unsqueeze(1)  # (B, 1, D): the shape model/nar_tts.py:286 documents
        enc = torch.randn(B, L, conf["d_model"], generator=g).requires_grad_()
        dec = torch.randn(B, T, conf["d_model"], generator=g).requires_grad_() if conf.get("dec_comb") else None
        look, pre = m(spk_ids=spk_ids, spk_feat=spk_feat)
        enc_out, dec_out = m.combine_spk_feat(spk_feat=pre, spk_feat_lookup=look, enc_output=enc, dec_input=dec)
        w1 = torch.randn(enc_out.shape, generator=g)
        loss = (enc_out * w1).sum()
        if dec_out is not None:
            w2 = torch.randn(dec_out.shape, generator=g)
            loss = loss + (dec_out * w2).sum()
            out[f"{tag}.w2"], out[f"{tag}.dec"], out[f"{tag}.dec_out"] = w2, dec, dec_out
        names = [n for n, _ in m.named_parameters()]
        grads = torch.autograd.grad(loss, [p for _, p in m.named_parameters()] + [enc] + ([dec] if dec is not None else []))
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
        out.update({f"{tag}.grad.{n}": gr for n, gr in zip(names, grads)})
        out[f"{tag}.grad_enc"] = grads[len(names)]
        if dec is not None:
            out[f"{tag}.grad_dec"] = grads[len(names) + 1]
        out.update({f"{tag}.enc": enc, f"{tag}.enc_out": enc_out, f"{tag}.w1": w1})
        if spk_ids is not None:
            out[f"{tag}.spk_ids"], out[f"{tag}.look"] = spk_ids, look
        if spk_feat is not None:
            out[f"{tag}.spk_feat"], out[f"{tag}.pre"] = spk_feat, pre
    save("g14_spk", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    g1_frontend()
    g2_specaug()
    g3_posenc()
    g4_layers()
    g5_stacks()
    g6_losses()
    g7_harness()
    g8_featnorm()
    g9_prenet()
    g10_tts()
    g11_decode()
    g12_recover()
    g13_ce_options()
    g14_spk()

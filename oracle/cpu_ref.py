"""TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of the SpeeChain ASR/TTS compute hot path.

Every function cites the reference lines (under /root/reference/) whose arithmetic it restates.
The functions are pure (explicit weights in, tensors out; nothing is mutated) and run on CPU in
fp32 with stock torch ops.  They are pinned against the imported reference itself by
``tests/golden/make_golden.py`` (run in the build container) and the committed fixtures under
``tests/golden/``; see ``oracle/__init__.py`` for the parity status.

Weights are passed as flat ``dict[str, Tensor]`` keyed exactly like the reference ``state_dict``
(``prefix + 'k_layer.weight'`` ...), so one seeded state dict drives reference, oracle and product.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import melscale

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------------------
# a14  make_mask_from_len          speechain/utilbox/train_util.py:31-75
# --------------------------------------------------------------------------------------------------


def make_mask_from_len(data_len: Tensor, max_len: Optional[int] = None, return_3d: bool = True) -> Tensor:
    """Boolean padding mask: True for positions < len (train_util.py:58-71)."""
    if max_len is None:
        max_len = int(data_len.max())
    mask = torch.arange(max_len)[None, :] < data_len.reshape(-1, 1).cpu()
    return mask[:, None, :] if return_3d else mask


# --------------------------------------------------------------------------------------------------
# a1-a4  frontend                  speechain/module/frontend/{speech2linear,linear2mel,speech2mel}.py
# --------------------------------------------------------------------------------------------------


def secs_to_samples(value, sr: int) -> int:
    """float seconds -> samples by truncation (speech2linear.py:85-90)."""
    return int(value * sr) if isinstance(value, float) else value


def preemphasis(wave: Tensor, wave_len: Tensor, coef: float) -> Tensor:
    """y[n] = x[n] - coef*x[n-1] (x[-1]=0), then y[n >= len] = 0 (speech2linear.py:167-173).

    wave: (B, T) f32.  The reference evaluates it as a 2-tap cross-correlation with taps [-coef, 1].
    """
    prev = F.pad(wave, (1, 0))[:, :-1]
    taps = torch.tensor([-coef, 1.0], dtype=wave.dtype)
    out = prev * taps[0] + wave * taps[1]
    idx = torch.arange(wave.size(1))[None, :]
    return out.masked_fill(idx >= wave_len.reshape(-1, 1), 0.0)


def stft_frames(wave: Tensor, n_fft: int, hop: int, win_length: int, window: Tensor, center: bool = True) -> Tensor:
    """Explicit framing restatement of torch.stft's semantics used at speech2linear.py:198.

    center=True reflect-pads the *batch tensor* by n_fft//2 on both sides (so a short utterance sees
    its own zero padding, not a reflection, at its right edge); the window is zero-padded centrally to
    n_fft when win_length < n_fft.  Returns windowed frames (B, T_frames, n_fft).
    """
    if center:
        wave = F.pad(wave[:, None, :], (n_fft // 2, n_fft // 2), mode="reflect")[:, 0]
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        window = F.pad(window, (left, n_fft - win_length - left))
    return wave.unfold(-1, n_fft, hop) * window


def stft_power_explicit(wave: Tensor, n_fft: int, hop: int, win_length: int, window: Tensor, center: bool = True,
                        dtype=torch.float64) -> Tensor:
    """|rFFT(frames)|^2 in ``dtype`` -- an independent definition of speech2linear.py:198,214."""
    frames = stft_frames(wave.to(dtype), n_fft, hop, win_length, window.to(dtype), center)
    spec = torch.fft.rfft(frames, n=n_fft, dim=-1)
    return spec.real ** 2 + spec.imag ** 2


def stft_power(wave: Tensor, n_fft: int, hop: int, win_length: int, window: Tensor, center: bool = True) -> Tensor:
    """(B, T_frames, n_fft//2+1) power via the stock op the reference calls (speech2linear.py:198,214)."""
    spec = torch.stft(wave, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window, center=center,
                      normalized=False, onesided=True, return_complex=True).transpose(1, 2)
    return spec.real ** 2 + spec.imag ** 2


def stft_feat_len(wave_len: Tensor, n_fft: int, hop: int, center: bool = True) -> Tensor:
    """(len + 2*(n_fft//2) - n_fft)//hop + 1 (speech2linear.py:201-212); does NOT mutate its input."""
    padded = wave_len + 2 * (n_fft // 2) if center else wave_len
    return torch.div(padded - n_fft, hop, rounding_mode="floor") + 1


def speech2linear(wave: Tensor, wave_len: Tensor, *, hop_length, win_length, sr: int = 16000, n_fft: int = None,
                  preemph: float = None, window: str = "hann", center: bool = True, mag_spec: bool = False,
                  return_energy: bool = False, clamp: float = 1e-10, logging: bool = False,
                  log_base: float = None, explicit: bool = False):
    """Speech2LinearSpec.forward (speech2linear.py:142-250), pre_stft_norm=None, normalized=False, onesided."""
    hop, win = secs_to_samples(hop_length, sr), secs_to_samples(win_length, sr)
    n_fft = win if n_fft is None else n_fft
    if wave.dim() == 3:
        wave = wave.squeeze(-1)
    if preemph is not None:
        wave = preemphasis(wave, wave_len, preemph)
    win_t = getattr(torch, f"{window}_window")(win, dtype=wave.dtype)
    if explicit:
        power = stft_power_explicit(wave, n_fft, hop, win, win_t, center, dtype=wave.dtype)
    else:
        power = stft_power(wave, n_fft, hop, win, win_t, center)
    feat_len = stft_feat_len(wave_len, n_fft, hop, center)
    frame_idx = torch.arange(power.size(1))[None, :]
    pad = frame_idx >= feat_len.reshape(-1, 1)
    energy = None
    if return_energy:
        energy = torch.sqrt(torch.clamp(power.sum(dim=-1), min=1e-10)).masked_fill(pad, 0.0)
    power = power.masked_fill(pad[:, :, None], 0.0)
    feat = torch.sqrt(power) if mag_spec else power
    if logging:
        feat = torch.clamp(feat, min=clamp).log()
        if log_base is not None:
            feat = feat / math.log(log_base)
    if return_energy:
        return feat, feat_len, energy, feat_len
    return feat, feat_len


def mel_matrix(n_fft: int, n_mels: int, sr: int = 16000, fmin: float = 0.0, fmax: float = None,
               mel_scale: str = "slaney", mel_norm: bool = True) -> Tensor:
    """(n_mels, n_fft//2+1) weight of the bias-free Linear at linear2mel.py:135-155."""
    fmax = fmax if fmax is not None else sr // 2
    fb = melscale.melscale_fbanks(n_fft // 2 + 1, fmin, fmax, n_mels, sr, norm="slaney" if mel_norm else None,
                                  mel_scale=mel_scale)
    return torch.from_numpy(fb.T.copy())


def linear2mel(feat: Tensor, mel_w: Tensor, clamp: float = 1e-10, logging: bool = True, log_base: float = 10.0) -> Tensor:
    """LinearSpec2MelSpec.forward (linear2mel.py:163-189)."""
    feat = F.linear(feat, mel_w)
    if logging:
        feat = torch.clamp(feat, min=clamp).log()
        if log_base is not None:
            feat = feat / math.log(log_base)
    return feat


def delta_features(feat: Tensor, delta_order: int, delta_N: int = 2) -> Tensor:
    """DeltaFeature.forward (delta_feat.py:57-81): regression deltas with replicate padding, concatenated."""
    kernel = torch.arange(-delta_N, delta_N + 1, dtype=feat.dtype) / (2 * sum(i * i for i in range(1, delta_N + 1)))
    outs, cur = [feat], feat
    for _ in range(delta_order):
        x = F.pad(cur[:, None], (0, 0, delta_N, delta_N), mode="replicate")
        cur = F.conv2d(x, kernel.reshape(1, 1, -1, 1))[:, 0]
        outs.append(cur)
    return torch.cat(outs, dim=-1)


def speech2mel(wave: Tensor, wave_len: Tensor, *, n_mels: int, hop_length, win_length, n_fft: int = None,
               sr: int = 16000, preemphasis: float = None, window: str = "hann", center: bool = True,
               mag_spec: bool = False, return_energy: bool = False, fmin: float = 0.0, fmax: float = None,
               clamp: float = 1e-10, logging: bool = True, log_base: float = 10.0, mel_scale: str = "slaney",
               mel_norm: bool = True, mel_w: Tensor = None, explicit: bool = False):
    """Speech2MelSpec.forward (speech2mel.py:158-189) = a2 -> a3 (kwargs named as the reference's)."""
    hop, win = secs_to_samples(hop_length, sr), secs_to_samples(win_length, sr)
    n_fft = win if n_fft is None else n_fft
    out = speech2linear(wave, wave_len, hop_length=hop, win_length=win, sr=sr, n_fft=n_fft, preemph=preemphasis,
                        window=window, center=center, mag_spec=mag_spec, return_energy=return_energy, logging=False,
                        explicit=explicit)
    if mel_w is None:
        mel_w = mel_matrix(n_fft, n_mels, sr, fmin, fmax, mel_scale, mel_norm)
    feat = linear2mel(out[0], mel_w.to(out[0].dtype), clamp, logging, log_base)
    if return_energy:
        return feat, out[1], out[2], out[3]
    return feat, out[1]


# --------------------------------------------------------------------------------------------------
# a5  SpecAugment                  speechain/module/augment/specaug.py:19-206
# --------------------------------------------------------------------------------------------------


def specaug_draw(feat_len: Tensor, batch: int, feat_dim: int, *, time_warp: bool = True, time_warp_window: int = 5,
                 freq_mask: bool = True, freq_mask_width=30, freq_mask_num: int = 2, time_mask: bool = True,
                 time_mask_width=0.05, time_mask_num: int = 2, device="cpu") -> Dict:
    """All random integers of one SpecAugment.forward call, in the reference's draw order.

    warp_center, warp_pos on the default CPU generator (specaug.py:98-108); freq mask_len then mask_pos
    (:145-158) and time mask_len then mask_pos (:178-191) on ``device``'s generator.
    """
    tmin = int(feat_len.min())
    d: Dict = dict(time_minlen=tmin, time_maxlen=int(feat_len.max()), warp_center=None, warp_pos=None,
                   freq_len=None, freq_pos=None, time_len=None, time_pos=None)
    W = time_warp_window
    if time_warp and tmin > 2 * W + 1:
        d["warp_center"] = int(torch.randint(low=W + 1, high=tmin - W, size=(1,))[0])
        d["warp_pos"] = int(torch.randint(low=d["warp_center"] - W, high=d["warp_center"] + W, size=(1,))[0])
    if freq_mask:
        fw = [0, freq_mask_width] if isinstance(freq_mask_width, int) else list(freq_mask_width)
        d["freq_len"] = torch.randint(fw[0], fw[1] + 1, size=(batch, freq_mask_num), device=device)
        d["freq_pos"] = torch.randint(0, max(1, feat_dim - int(d["freq_len"].max())), size=(batch, freq_mask_num),
                                      device=device)
    if time_mask:
        tw = [0, time_mask_width] if isinstance(time_mask_width, (int, float)) else list(time_mask_width)
        lo, hi = tw
        lo = int(lo * tmin) if isinstance(lo, float) else lo
        hi = int(hi * tmin) if isinstance(hi, float) else hi
        hi = min(hi, tmin)
        d["time_len"] = torch.randint(lo, hi + 1, size=(batch, time_mask_num), device=device)
        d["time_pos"] = torch.randint(0, max(1, tmin - int(d["time_len"].max())), size=(batch, time_mask_num),
                                      device=device)
    return d


def _keys_cubic(t: Tensor, A: float = -0.75) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """The four Keys cubic-convolution taps used by aten's bicubic upsampling."""
    def inner(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def outer(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return outer(t + 1), inner(t), inner(1 - t), outer(2 - t)


def time_resample_cubic(seg: Tensor, out_len: int) -> Tensor:
    """1-D restatement of F.interpolate(seg[:,None], size=(out_len, F), mode='bicubic', align_corners=True)
    (specaug.py:111-122) for seg (B, T_in, F): the F axis keeps its size so only T is resampled.
    src = dst * fp32((T_in-1)/(out_len-1)); 4 taps, border-clamped indices, A=-0.75."""
    t_in = seg.size(1)
    scale = torch.tensor((t_in - 1) / (out_len - 1) if out_len > 1 else 0.0, dtype=torch.float32)
    src = torch.arange(out_len, dtype=torch.float32) * scale
    i0 = torch.floor(src)
    lam = (src - i0).clamp(0.0, 1.0)
    i0 = i0.long()
    w = _keys_cubic(lam)
    out = torch.zeros(seg.size(0), out_len, seg.size(2), dtype=seg.dtype)
    for j in range(4):
        idx = (i0 - 1 + j).clamp(0, t_in - 1)
        out = out + seg[:, idx, :] * w[j][None, :, None]
    return out


def specaug_apply(feat: Tensor, feat_len: Tensor, draws: Dict, *, time_warp_mode: str = "bicubic",
                  feat_norm: bool = True, explicit_warp: bool = False) -> Tensor:
    """SpecAugment.forward given its random integers (specaug.py:86-204).  Returns a new tensor."""
    feat = feat.clone()
    B, T, Fd = feat.shape
    tmin = draws["time_minlen"]
    if draws["warp_center"] is not None:
        c, p = draws["warp_center"], draws["warp_pos"]
        if explicit_warp:
            left = time_resample_cubic(feat[:, :c], p)
            right = time_resample_cubic(feat[:, c:tmin], tmin - p)
        else:
            x = feat[:, None]
            left = F.interpolate(x[:, :, :c], size=(p, Fd), mode=time_warp_mode, align_corners=True)[:, 0]
            right = F.interpolate(x[:, :, c:tmin], size=(tmin - p, Fd), mode=time_warp_mode, align_corners=True)[:, 0]
        feat[:, :p] = left
        feat[:, p:tmin] = right
    mask = specaug_mask(draws, B, T, Fd)
    if mask is not None:
        fill = 0.0 if feat_norm else feat.mean()
        feat = feat.masked_fill(mask, fill)
    return feat


def specaug_mask(draws: Dict, B: int, T: int, Fd: int) -> Optional[Tensor]:
    """The (B, T, F) boolean union of frequency and time ranges (specaug.py:160-199)."""
    mask = None
    if draws["freq_len"] is not None:
        ln, ps = draws["freq_len"].cpu()[:, :, None], draws["freq_pos"].cpu()[:, :, None]
        ax = torch.arange(Fd)[None, None, :]
        mask = ((ps <= ax) & (ax < ps + ln)).any(dim=1, keepdim=True)
    if draws["time_len"] is not None:
        ln, ps = draws["time_len"].cpu()[:, None, :], draws["time_pos"].cpu()[:, None, :]
        ax = torch.arange(T)[None, :, None]
        tm = ((ps <= ax) & (ax < ps + ln)).any(dim=-1, keepdim=True)
        mask = tm if mask is None else torch.logical_or(mask, tm)
    if mask is not None:
        mask = mask.expand(B, T, Fd)
    return mask


# --------------------------------------------------------------------------------------------------
# a6  PositionalEncoding           speechain/module/transformer/pos_enc.py:24-188
# --------------------------------------------------------------------------------------------------


def posenc_table(max_len: int, d_model: int, posenc_type: str = "mix") -> Tensor:
    """(1, max_len, d) sinusoid table (pos_enc.py:115-146)."""
    pos = torch.arange(0, max_len, dtype=torch.float)[:, None]
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * (math.log(10000.0) / d_model))
    tab = torch.zeros(max_len, d_model)
    if posenc_type == "mix":
        tab[:, 0::2] = torch.sin(pos / div)
        tab[:, 1::2] = torch.cos(pos / div)
    else:
        div2 = torch.exp(torch.arange(d_model, 2 * d_model, 2, dtype=torch.float) * (math.log(10000.0) / d_model))
        tab[:, : d_model // 2] = torch.sin(pos / div)
        tab[:, d_model // 2:] = torch.cos(pos / div2)
    return tab[None]


def posenc_forward(x: Tensor, P: Dict[str, Tensor], prefix: str, d_model: int, emb_scale: bool = False) -> Tensor:
    """PositionalEncoding.forward in eval mode (pos_enc.py:148-182): [LN eps 1e-5] -> [*sqrt(d)] -> + table*[alpha]."""
    if prefix + "emb_layernorm.weight" in P:
        x = F.layer_norm(x, (d_model,), P[prefix + "emb_layernorm.weight"], P[prefix + "emb_layernorm.bias"], 1e-5)
    if emb_scale:
        x = x * math.sqrt(d_model)
    tab = P[prefix + "posenc"][:, : x.size(1)]
    if prefix + "alpha" in P:
        tab = tab * P[prefix + "alpha"]
    return x + tab


# --------------------------------------------------------------------------------------------------
# a7-a10  attention / FFN / layers / stacks      speechain/module/transformer/*.py
# --------------------------------------------------------------------------------------------------


def mha(k: Tensor, v: Tensor, q: Tensor, mask: Optional[Tensor], P: Dict[str, Tensor], prefix: str, num_heads: int,
        scale_dp_by_head: bool = False) -> Tuple[Tensor, Tensor]:
    """MultiHeadedAttention.forward, dropout off (attention.py:67-133).  Argument order (k, v, q, mask);
    mask is (B,1,Tk) or (B,Tq,Tk) bool with True = attend.  Returns (out (B,Tq,d), probs (B,H,Tq,Tk))."""
    B, d = k.size(0), q.size(-1)
    dh = d // num_heads
    def split(x, name):
        return F.linear(x, P[prefix + name + ".weight"], P[prefix + name + ".bias"]).view(B, -1, num_heads, dh).transpose(1, 2)
    kh, vh, qh = split(k, "k_layer"), split(v, "v_layer"), split(q, "q_layer")
    scale = 1 / math.sqrt(dh) if scale_dp_by_head else 1 / math.sqrt(d)
    scores = torch.matmul(qh, kh.transpose(2, 3)) * scale
    if mask is not None:
        scores = scores.masked_fill(~mask[:, None], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, vh).transpose(1, 2).contiguous().view(B, -1, d)
    return F.linear(ctx, P[prefix + "output_layer.weight"], P[prefix + "output_layer.bias"]), probs


def _activation(name: str):
    return getattr(torch.nn, name)()


def ffn(x: Tensor, P: Dict[str, Tensor], prefix: str, activation: str = "ReLU", fdfwd_type: str = "linear") -> Tensor:
    """PositionwiseFeedForward.forward, dropout off (feed_forward.py:77-106).  'conv' = Conv1d k odd, 'same'
    zero padding, stride 1 over time (prenet/conv1d.py:73-90), weights ``in_layer.conv_lyr.weight`` (f,d,k)."""
    act = _activation(activation)
    if fdfwd_type == "linear":
        h = act(F.linear(x, P[prefix + "in_layer.weight"], P[prefix + "in_layer.bias"]))
        return F.linear(h, P[prefix + "out_layer.weight"], P[prefix + "out_layer.bias"])
    w1, w2 = P[prefix + "in_layer.conv_lyr.weight"], P[prefix + "out_layer.conv_lyr.weight"]
    h = act(F.conv1d(x.transpose(1, 2), w1, P[prefix + "in_layer.conv_lyr.bias"], padding=(w1.size(-1) - 1) // 2))
    return F.conv1d(h, w2, P[prefix + "out_layer.conv_lyr.bias"], padding=(w2.size(-1) - 1) // 2).transpose(1, 2)


def _ln(x, P, name, eps=1e-6):
    return F.layer_norm(x, (x.size(-1),), P[name + ".weight"], P[name + ".bias"], eps)


def encoder_layer(src: Tensor, mask: Tensor, P, prefix: str, num_heads: int, *, scale_dp_by_head=False,
                  activation="ReLU", fdfwd_type="linear", layernorm_first=True) -> Tuple[Tensor, Tensor]:
    """TransformerEncoderLayer.forward, dropout off (encoder.py:100-143); LN eps 1e-6 (:97-98)."""
    x = _ln(src, P, prefix + "att_layernorm") if layernorm_first else src
    att, probs = mha(x, x, x, mask, P, prefix + "multihead_att.", num_heads, scale_dp_by_head)
    y = att + src
    if not layernorm_first:
        y = _ln(y, P, prefix + "att_layernorm")
    z = _ln(y, P, prefix + "fdfwd_layernorm") if layernorm_first else y
    out = ffn(z, P, prefix + "feed_forward.", activation, fdfwd_type) + y
    if not layernorm_first:
        out = _ln(out, P, prefix + "fdfwd_layernorm")
    return out, probs


def subsequent_mask(batch: int, n: int) -> Tensor:
    """Lower-triangular (incl. diagonal) True mask (encoder.py:294-308)."""
    return ~torch.triu(torch.ones(batch, n, n, dtype=torch.bool), diagonal=1)


def transformer_encoder(src: Tensor, mask: Tensor, P, prefix: str, *, num_layers: int, num_heads: int, d_model: int,
                        emb_scale=False, scale_dp_by_head=False, activation="ReLU", fdfwd_type="linear",
                        layernorm_first=True, uni_direction=False):
    """TransformerEncoder.forward, dropout off (encoder.py:310-349) -> (out, mask, attmat[L], hidden[L])."""
    x = posenc_forward(src, P, prefix + "posenc.", d_model, emb_scale)
    if uni_direction:
        B, _, T = mask.shape
        mask = torch.logical_and(mask.repeat(1, T, 1), subsequent_mask(B, T))
    attmat, hidden = [], []
    for l in range(num_layers):
        x, a = encoder_layer(x, mask, P, f"{prefix}trfm_layers.{l}.", num_heads, scale_dp_by_head=scale_dp_by_head,
                             activation=activation, fdfwd_type=fdfwd_type, layernorm_first=layernorm_first)
        attmat.append(a)
        hidden.append(x.clone())
    if layernorm_first:
        x = _ln(x, P, prefix + "layernorm")
    return x, mask, attmat, hidden


def decoder_layer(tgt: Tensor, src: Tensor, tgt_mask: Tensor, src_mask: Tensor, P, prefix: str, num_heads: int, *,
                  scale_dp_by_head=False, activation="ReLU", layernorm_first=True):
    """TransformerDecoderLayer.forward, dropout off (decoder.py:94-175).  enc-dec attention never scales by
    head (decoder.py:76-78)."""
    x = _ln(tgt, P, prefix + "self_att_ln") if layernorm_first else tgt
    sa, sp = mha(x, x, x, tgt_mask, P, prefix + "self_att.", num_heads, scale_dp_by_head)
    y = sa + tgt
    if not layernorm_first:
        y = _ln(y, P, prefix + "self_att_ln")
    yn = _ln(y, P, prefix + "encdec_att_ln") if layernorm_first else y
    ea, ep = mha(src, src, yn, src_mask, P, prefix + "encdec_att.", num_heads, False)
    z = ea + y
    if not layernorm_first:
        z = _ln(z, P, prefix + "encdec_att_ln")
    zn = _ln(z, P, prefix + "fdfwd_ln") if layernorm_first else z
    out = ffn(zn, P, prefix + "feed_forward.", activation, "linear") + z
    if not layernorm_first:
        out = _ln(out, P, prefix + "fdfwd_ln")
    return out, sp, ep


def transformer_decoder(tgt: Tensor, src: Tensor, tgt_mask: Tensor, src_mask: Tensor, P, prefix: str, *,
                        num_layers: int, num_heads: int, d_model: int, emb_scale=True, scale_dp_by_head=False,
                        activation="ReLU", layernorm_first=True):
    """TransformerDecoder.forward, dropout off (decoder.py:307-362).  Decoder *layers* are always built with
    the layer default layernorm_first=True (decoder.py:271-285); ``layernorm_first`` only gates the final LN."""
    x = posenc_forward(tgt, P, prefix + "posenc.", d_model, emb_scale)
    B, _, L = tgt_mask.shape
    m = torch.logical_and(tgt_mask.repeat(1, L, 1), subsequent_mask(B, L))
    self_att, encdec_att, hidden = [], [], []
    for l in range(num_layers):
        x, a, e = decoder_layer(x, src, m, src_mask, P, f"{prefix}trfm_layers.{l}.", num_heads,
                                scale_dp_by_head=scale_dp_by_head, activation=activation, layernorm_first=True)
        self_att.append(a)
        encdec_att.append(e)
        hidden.append(x.clone())
    if layernorm_first:
        x = _ln(x, P, prefix + "layernorm")
    return x, self_att, encdec_att, hidden


# --------------------------------------------------------------------------------------------------
# a11  TokenPostnet                speechain/module/postnet/token.py:40-49
# --------------------------------------------------------------------------------------------------


def token_postnet(x: Tensor, P, prefix: str) -> Tensor:
    return F.linear(x, P[prefix + "linear.weight"], P[prefix + "linear.bias"])


# --------------------------------------------------------------------------------------------------
# a12  CTCLoss                     speechain/criterion/ctc.py:25-66
# --------------------------------------------------------------------------------------------------


def ctc_loss(ctc_logits: Tensor, enc_len: Tensor, text: Tensor, text_len: Tensor, blank: int = 0,
             zero_infinity: bool = True) -> Tensor:
    """log_softmax over V, strip the leading/trailing <sos/eos> of ``text``, per-utterance CTC negative
    log-likelihood (NOT length-normalised), zero_infinity, mean over the batch."""
    lp = ctc_logits.transpose(0, 1).log_softmax(dim=-1)
    tgt, tgt_len = text[:, 1:-1], text_len - 2
    flat = torch.cat([tgt[i, : tgt_len[i]] for i in range(tgt.size(0))])
    loss = F.ctc_loss(lp, flat, enc_len, tgt_len, blank=blank, reduction="none", zero_infinity=zero_infinity)
    return loss.mean()


def ctc_nll_explicit(log_probs: Tensor, target: Sequence[int], blank: int = 0) -> float:
    """Independent alpha recursion for ONE utterance (log_probs (T, V) float64), Graves et al. 2006.
    Used to pin ``ctc_loss`` on small cases."""
    ext = [blank]
    for s in target:
        ext += [int(s), blank]
    S, T = len(ext), log_probs.size(0)
    neg = float("-inf")
    alpha = [neg] * S
    alpha[0] = float(log_probs[0, blank])
    if S > 1:
        alpha[1] = float(log_probs[0, ext[1]])
    def lse(*xs):
        m = max(xs)
        return m if m == neg else m + math.log(sum(math.exp(x - m) for x in xs))
    for t in range(1, T):
        new = [neg] * S
        for s in range(S):
            terms = [alpha[s]]
            if s >= 1:
                terms.append(alpha[s - 1])
            if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                terms.append(alpha[s - 2])
            new[s] = lse(*terms) + float(log_probs[t, ext[s]])
        alpha = new
    return -(lse(alpha[S - 1], alpha[S - 2]) if S > 1 else alpha[0])


# --------------------------------------------------------------------------------------------------
# a13  CrossEntropy                speechain/criterion/cross_entropy.py:97-212
# --------------------------------------------------------------------------------------------------


def cross_entropy(logits: Tensor, text: Tensor, text_len: Tensor, label_smoothing: float = 0.0,
                  length_normalized: bool = False, temperature: float = 1.0, token_weights: Optional[Tensor] = None,
                  confid_threshold: float = 0.0, confid_level: str = "sentence") -> Tensor:
    """Label-smoothed, length-masked CE with the optional token re-weighting (:148-149) and confidence thresholding (:160-199).  When
    ``text`` carries a leading <sos/eos> (one longer than logits) it is dropped and lengths shrink by one (:115-125)."""
    if logits.size(1) == text.size(1) - 1:
        text, text_len = text[:, 1:], text_len - 1
    B, L, V = logits.shape
    lp = torch.log_softmax(logits.reshape(B * L, V) / temperature, dim=-1)
    lp_t = lp.gather(1, text.reshape(-1, 1)).squeeze(-1)
    if label_smoothing > 0:
        tok = lp_t * (1 - label_smoothing) + (lp * (label_smoothing / V)).sum(dim=1)
    else:
        tok = lp_t
    if token_weights is not None:
        tok = tok * token_weights[text.reshape(-1)]
    valid = make_mask_from_len(text_len, max_len=L, return_3d=False)
    drop = ~valid.reshape(-1)
    valid_sent = None
    if confid_threshold > 0:
        if confid_level == "token":  # tokens the model is not confident about leave the loss and the sentence lengths (:161-170)
            low = lp_t <= math.log(confid_threshold)
            drop = drop | low
            text_len = (~low.reshape(B, L)).masked_fill(~valid, False).sum(dim=-1)
            valid_sent = text_len > 0
        else:  # whole sentences whose summed log-probability is below len * log(threshold) (:171-184)
            sent_lp = lp_t.reshape(B, L).masked_fill(~valid, 0.0).sum(dim=-1, keepdim=True)
            low = sent_lp <= (text_len * math.log(confid_threshold)).unsqueeze(-1)
            drop = drop | low.expand(-1, L).reshape(-1)
            valid_sent = ~low.squeeze(-1)
    sent = tok.masked_fill(drop, 0.0).reshape(B, L).sum(dim=-1)
    if length_normalized:
        sent = sent / (text_len + 1e-10)
    return -sent.mean() if confid_threshold == 0 else -sent.sum() / (valid_sent.sum() + 1e-10)


# --------------------------------------------------------------------------------------------------
# a-harness  the callers' sequencing, restated (module/encoder/asr.py:103-131,
#            module/decoder/ar_asr.py:49-78, model/ar_asr.py:386-388,453-458,550)
# --------------------------------------------------------------------------------------------------


def asr_forward_loss(feat: Tensor, feat_len: Tensor, text: Tensor, text_len: Tensor, P: Dict[str, Tensor], cfg: Dict):
    """Log-mel (already subsampled to the encoder rate) -> encoder -> decoder -> heads -> 0.7 CE + 0.3 CTC.

    ``feat`` (B, T', d_model) is the encoder input (the Conv2dPrenet is outside the named path);
    ``text`` (B, L) starts and ends with <sos/eos>; the decoder consumes text[:, :-1] (model/ar_asr.py:386-388).
    Weight prefixes: 'encoder.', 'embed.weight', 'decoder.', 'postnet.', 'ctc_layer.'.
    """
    enc_mask = make_mask_from_len(feat_len, max_len=feat.size(1))
    enc, _, _, _ = transformer_encoder(feat, enc_mask, P, "encoder.", num_layers=cfg["enc_layers"],
                                       num_heads=cfg["num_heads"], d_model=cfg["d_model"],
                                       activation=cfg["activation"], emb_scale=False)
    dec_in, dec_len = text.clone(), text_len - 1
    dec_in[torch.arange(text.size(0)), text_len - 1] = 0  # each sentence's trailing <sos/eos> -> padding idx
    dec_in = dec_in[:, :-1]
    emb = F.embedding(dec_in, P["embed.weight"], padding_idx=0)
    tmask = make_mask_from_len(dec_len, max_len=dec_in.size(1))
    dec, _, _, _ = transformer_decoder(emb, enc, tmask, enc_mask, P, "decoder.", num_layers=cfg["dec_layers"],
                                       num_heads=cfg["num_heads"], d_model=cfg["d_model"],
                                       activation=cfg["activation"], emb_scale=False)
    logits = token_postnet(dec, P, "postnet.")
    ctc_logits = token_postnet(enc, P, "ctc_layer.")
    ce = cross_entropy(logits, text, text_len, label_smoothing=cfg.get("label_smoothing", 0.0))
    ctc = ctc_loss(ctc_logits, feat_len, text, text_len)
    w = cfg.get("ctc_weight", 0.3)
    return dict(enc=enc, logits=logits, ctc_logits=ctc_logits, ce=ce, ctc=ctc, loss=(1 - w) * ce + w * ctc)


# ---- feature normalisation (row f1) --------------------------------------------------------------------------------------
class FeatureNormalization:
    """Stateful restatement of speechain/module/norm/feat_norm.py (FeatureNormalization.forward :78-338,
    register_mean_std_batch :452-509, update_aver_mean_std :511-531, recover :533-604), single process.

    Mirrored on purpose: statistics over the valid frames ``feat[i][:feat_len[i]]`` with torch's unbiased std, clamped at
    ``clamp`` (:115-133); the normalisation is applied to the WHOLE padded tensor (:137-139, :322-331); the running
    statistics are re-created on every update and weighted 1 / (number of batches seen), the counter restarting at 1 whenever
    ``epoch`` is None (:472-476) and frozen once ``epoch > max_epoch_num`` (:477-480, :483); the average over groups takes
    every buffer whose name ends in ``_mean`` / ``_std`` -- including the previous average itself (:519-528).
    ``mean_norm=False`` with the global / group types raises in the reference (:475 reads ``curr_aver_mean.device`` of None);
    here it raises a ValueError."""

    def __init__(self, norm_type: str = "global", mean_norm: bool = True, std_norm: bool = True, clamp: float = 1e-10,
                 max_epoch_num: int = 4):
        if norm_type not in ("utterance", "batch", "group", "global"):
            raise ValueError(f"norm_type can only be one of 'utterance', 'batch', 'group', 'global', but got {norm_type}")
        if not mean_norm and norm_type in ("global", "group"):
            raise ValueError("mean_norm=False is broken in the reference for the global / group types (feat_norm.py:475)")
        self.norm_type, self.mean_norm, self.std_norm = norm_type, mean_norm, std_norm
        self.clamp, self.max_epoch_num = clamp, max_epoch_num
        self.training = True
        self.buffers = {}  # insertion-ordered, like nn.Module._buffers

    def _register(self, mean, std, prefix, epoch):
        b = self.buffers
        if epoch is None or f"{prefix}_batch" not in b:
            b[f"{prefix}_batch"] = torch.tensor([1], dtype=torch.int64)
        elif epoch <= self.max_epoch_num:
            b[f"{prefix}_batch"] = b[f"{prefix}_batch"] + 1
        if epoch is None or epoch <= self.max_epoch_num:
            w = 1 / b[f"{prefix}_batch"]
            if self.mean_norm:
                b[f"{prefix}_mean"] = mean if f"{prefix}_mean" not in b else w * mean + (1 - w) * b[f"{prefix}_mean"]
            if self.std_norm:
                b[f"{prefix}_std"] = std if f"{prefix}_std" not in b else w * std + (1 - w) * b[f"{prefix}_std"]

    def _update_aver(self, epoch):
        if epoch is None or epoch <= self.max_epoch_num:
            means = [v for k, v in self.buffers.items() if k.endswith("_mean")]
            stds = [v for k, v in self.buffers.items() if k.endswith("_std")]
            am, as_ = None, None
            for v in means:
                am = v.clone() if am is None else am + v
            for v in stds:
                as_ = v.clone() if as_ is None else as_ + v
            self.buffers["aver_mean"] = am / len(means)
            self.buffers["aver_std"] = as_ / len(stds)

    def __call__(self, feat: Tensor, feat_len: Tensor, group_ids=None, epoch: int = None):
        squeeze = feat.dim() == 2
        if squeeze:
            feat = feat.unsqueeze(-1)
        B = feat.size(0)
        means = torch.stack([feat[i][: feat_len[i]].mean(dim=0) for i in range(B)]) if self.mean_norm else None
        stds = torch.clamp(torch.stack([feat[i][: feat_len[i]].std(dim=0) for i in range(B)]), min=self.clamp) if self.std_norm else None
        if self.norm_type == "utterance":
            feat = feat - means.unsqueeze(1) if means is not None else feat
            feat = feat / stds.unsqueeze(1) if stds is not None else feat
        elif self.norm_type == "group" and isinstance(group_ids, Tensor):
            if self.training:
                order = []
                for g in group_ids.tolist():
                    if g not in order:
                        order.append(g)
                for g in order:
                    sel = group_ids == g
                    self._register(means[sel].mean(dim=0) if means is not None else None,
                                   stds[sel].mean(dim=0) if stds is not None else None, g, epoch)
                self._update_aver(epoch)
            feat = feat.clone()
            for i in range(B):
                g = int(group_ids[i])
                if self.mean_norm:
                    feat[i] -= self.buffers.get(f"{g}_mean", self.buffers["aver_mean"])
                if self.std_norm:
                    feat[i] /= self.buffers.get(f"{g}_std", self.buffers["aver_std"])
        else:
            bm = means.mean(dim=0) if (self.training and means is not None) else None
            bs = stds.mean(dim=0) if (self.training and stds is not None) else None
            if self.norm_type == "batch":
                if self.training:
                    feat = feat - bm if bm is not None else feat
                    feat = feat / bs if bs is not None else feat
                else:
                    feat = feat - means.unsqueeze(1) if means is not None else feat
                    feat = feat / stds.unsqueeze(1) if stds is not None else feat
            else:
                if self.norm_type == "group" and not isinstance(group_ids, (str, int)):
                    raise AssertionError("group-level normalisation of a whole batch needs group_ids as a string or integer")
                prefix = "global" if self.norm_type == "global" else group_ids
                if self.training:
                    self._register(bm, bs, prefix, epoch)
                feat = feat - self.buffers[f"{prefix}_mean"] if means is not None else feat
                feat = feat / self.buffers[f"{prefix}_std"] if stds is not None else feat
        return (feat.squeeze(-1) if squeeze else feat), feat_len

    def recover(self, feat: Tensor, group_ids=None) -> Tensor:
        """feat_norm.py:533-604 for the global type and whole-batch groups."""
        assert self.norm_type not in ("utterance", "batch")
        prefix = "global" if self.norm_type == "global" else str(group_ids)
        feat = feat * self.buffers[f"{prefix}_std"] if self.std_norm else feat
        return feat + self.buffers[f"{prefix}_mean"] if self.mean_norm else feat


# ---- Conv2d prenet (row f2) ---------------------------------------------------------------------------------------------
def conv2d_prenet_layout(conv_dims, conv_batchnorm: bool, conv_activation: Optional[str], conv_dropout, lnr_dims,
                         zero_centered: bool = False) -> List[Dict[str, Optional[int]]]:
    """Positions of each block's members inside the reference's ``nn.Sequential`` ``conv`` (conv2d.py:168-205): Conv2d, then
    BatchNorm2d if ``conv_batchnorm``, then the activation unless (last block, no linear part, zero_centered and a ReLU-series
    activation) (:186-190), then Dropout if ``conv_dropout``.  Parameter names follow: ``conv.{pos}.weight`` ..."""
    out, pos = [], 0
    for i in range(len(conv_dims)):
        blk = dict(conv=pos, bn=None, act=False)
        pos += 1
        if conv_batchnorm:
            blk["bn"] = pos
            pos += 1
        if conv_activation is not None:
            if not (i == len(conv_dims) - 1 and lnr_dims is None) or not (zero_centered and "ReLU" in conv_activation):
                blk["act"] = True
                pos += 1
        if conv_dropout is not None:
            pos += 1
        out.append(blk)
    return out


def conv2d_prenet(feat: Tensor, feat_len: Tensor, P: Dict[str, Tensor], prefix: str = "", *, conv_dims=(64, 64), conv_kernel=3,
                  conv_stride=2, conv_padding=0, conv_batchnorm: bool = False, conv_activation: Optional[str] = "ReLU",
                  lnr_dims=512, lnr_activation: Optional[str] = None, zero_centered: bool = False, training: bool = True,
                  momentum: float = 0.1, bn_eps: float = 1e-5):
    """Conv2dPrenet.forward (conv2d.py:227-280) + LinearPrenet.forward (linear.py:112-128), dropout off.
    (B, T, F) -> unsqueeze(1) -> [Conv2d(k, s, p; bias iff no batchnorm) -> BatchNorm2d -> activation] x len(conv_dims) ->
    (B, C, T', F') -> transpose(1, 2).view(B, T', C * F') (channel-major flattening, :251) -> Linear blocks.
    Lengths: floor((len - k) / s) + 1 per block (:254-263; the padding is not counted there, as in the reference).
    BatchNorm2d in training mode normalises with the biased batch variance and returns the updated running statistics
    (momentum 0.1, unbiased variance) as ``stats[f'conv.{pos}.running_mean']`` ...; in eval mode it uses the running ones.
    Inputs narrower than the receptive field are zero-padded first (:236-247).  Returns (out, out_len, stats)."""
    conv_dims = [conv_dims] if isinstance(conv_dims, int) else list(conv_dims)
    pair = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v, v)
    kernel, stride, padding = pair(conv_kernel), pair(conv_stride), pair(conv_padding)
    min_h = min_w = 1
    for _ in conv_dims:  # (:141-152: both use index [0] of kernel / stride / padding)
        min_h = (min_h - 1) * stride[0] + kernel[0] - 2 * padding[0]
        min_w = (min_w - 1) * stride[0] + kernel[0] - 2 * padding[0]
    if feat.size(1) < min_h:
        d = min_h - feat.size(1)
        feat = F.pad(feat, (0, 0, int(d / 2), d - int(d / 2)))
        feat_len = feat_len + d
    if feat.size(2) < min_w:
        d = min_w - feat.size(2)
        feat = F.pad(feat, (int(d / 2), d - int(d / 2)))
    stats = {}
    x = feat.unsqueeze(1)
    for blk in conv2d_prenet_layout(conv_dims, conv_batchnorm, conv_activation, None, lnr_dims, zero_centered):
        name = f"{prefix}conv.{blk['conv']}."
        x = F.conv2d(x, P[name + "weight"], P.get(name + "bias"), stride=stride, padding=padding)
        if blk["bn"] is not None:
            bn = f"{prefix}conv.{blk['bn']}."
            rm, rv = P[bn + "running_mean"].detach().clone(), P[bn + "running_var"].detach().clone()
            x = F.batch_norm(x, rm, rv, P[bn + "weight"], P[bn + "bias"], training, momentum, bn_eps)
            stats[bn + "running_mean"], stats[bn + "running_var"] = rm, rv
        if blk["act"]:
            x = _activation(conv_activation)(x)
    B, C, T, Fd = x.shape
    x = x.transpose(1, 2).contiguous().view(B, T, C * Fd)
    for _ in conv_dims:
        feat_len = torch.div(feat_len - kernel[0], stride[0], rounding_mode="floor") + 1
    if int(feat_len.max()) != x.size(1):  # (:266-271) e.g. any conv_padding > 0 with stride 2, or a batch that is not tightly padded
        raise RuntimeError("There is a bug in the Conv2dPrenet.The calculation of the feature lengths has something wrong.")
    if lnr_dims is not None:
        dims = [lnr_dims] if isinstance(lnr_dims, int) else list(lnr_dims)
        pos = 0
        for i in range(len(dims)):
            x = F.linear(x, P[f"{prefix}linear.linear.{pos}.weight"], P[f"{prefix}linear.linear.{pos}.bias"])
            pos += 1
            if lnr_activation is not None and (i != len(dims) - 1 or not (zero_centered and "ReLU" in lnr_activation)):
                x = _activation(lnr_activation)(x)
                pos += 1
    return x, feat_len, stats


# --------------------------------------------------------------------------------------------------
# f4  FastSpeech2 glue (BASELINE configs[4])
#     prenet/conv1d.py:20-128 (Conv1dEv), prenet/var_pred.py:42-233, postnet/conv1d.py:16-166, prenet/embed.py:13-66,
#     encoder/tts.py:20-86, decoder/nar_tts.py:21-502, criterion/least_error.py:15-129, model/nar_tts.py:472-560
# --------------------------------------------------------------------------------------------------

def conv1d_same(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """Conv1dEv(padding_mode='same', stride 1, odd kernel) on a (B, T, C) tensor (prenet/conv1d.py:79-86, 120-128)."""
    assert w.size(-1) % 2 == 1
    return F.conv1d(x.transpose(1, 2), w, b, padding=(w.size(-1) - 1) // 2).transpose(1, 2)


def var_predictor(x: Tensor, P: Dict[str, Tensor], prefix: str, num_layers: int) -> Tensor:
    """Conv1dVarPredictor.forward, dropout off (var_pred.py:185-215): [Conv1dEv -> ReLU -> LayerNorm over channels, eps 1e-12
    (:9-39, :141-160) -> Dropout] x num_layers, then Linear(C, 1) squeezed.  The Sequential holds 4 entries per block."""
    for i in range(num_layers):
        x = torch.relu(conv1d_same(x, P[f"{prefix}conv.{4 * i}.conv_lyr.weight"], P[f"{prefix}conv.{4 * i}.conv_lyr.bias"]))
        x = F.layer_norm(x, (x.size(-1),), P[f"{prefix}conv.{4 * i + 2}.weight"], P[f"{prefix}conv.{4 * i + 2}.bias"], 1e-12)
    return F.linear(x, P[prefix + "linear.weight"], P[prefix + "linear.bias"]).squeeze(-1)


def var_embed(scalar: Tensor, P: Dict[str, Tensor], prefix: str) -> Tensor:
    """Conv1dVarPredictor.emb_pred_scalar (var_pred.py:217-233): (B, T) -> (B, T, d) through Conv1dEv(1, d, conv_emb_kernel)."""
    return conv1d_same(scalar.unsqueeze(-1), P[prefix + "conv_emb.0.conv_lyr.weight"], P[prefix + "conv_emb.0.conv_lyr.bias"])


def conv1d_postnet(x: Tensor, P: Dict[str, Tensor], prefix: str, num_layers: int, training: bool, activation: str = "Tanh",
                   momentum: float = 0.1, eps: float = 1e-5) -> Tuple[Tensor, Dict[str, Tensor]]:
    """Conv1dPostnet.forward, dropout off (postnet/conv1d.py:146-166): [Conv1dEv without bias -> BatchNorm1d -> activation -> Dropout] per
    block over the WHOLE padded (B, T) grid (no length masking in the reference); 4 Sequential entries per block.  Returns the
    output and the running statistics after this call (nn.BatchNorm1d: momentum 0.1, unbiased running variance)."""
    stats = {}
    act = _activation(activation)
    for i in range(num_layers):
        x = conv1d_same(x, P[f"{prefix}conv.{4 * i}.conv_lyr.weight"], None)
        rm, rv = P[f"{prefix}conv.{4 * i + 1}.running_mean"].clone(), P[f"{prefix}conv.{4 * i + 1}.running_var"].clone()
        x = F.batch_norm(x.transpose(1, 2), rm, rv, P[f"{prefix}conv.{4 * i + 1}.weight"], P[f"{prefix}conv.{4 * i + 1}.bias"], training,
                         momentum, eps).transpose(1, 2)
        stats[f"{prefix}conv.{4 * i + 1}.running_mean"], stats[f"{prefix}conv.{4 * i + 1}.running_var"] = rm, rv
        x = act(x)
    return x, stats


def average_scalar_by_duration(frame_scalar: Tensor, duration: Tensor) -> Tensor:
    """FastSpeech2Decoder.average_scalar_by_duration (decoder/nar_tts.py:150-204): token k averages the frames [sum(d[:k]), sum(d[:k+1]))
    that exist (frames beyond the tensor are not counted); no frames -> 0 (the 1e-10 in the denominator)."""
    B, T = frame_scalar.shape
    end = torch.cumsum(duration, dim=1)
    start = end - duration
    t = torch.arange(T)[None, None, :]
    m = (t >= start.unsqueeze(-1)) & (t < end.unsqueeze(-1))
    return (frame_scalar.unsqueeze(1) * m).sum(-1) / (m.sum(-1) + 1e-10)


def proc_duration(duration: Tensor, min_frame_num: int = 0, max_frame_num: Optional[int] = None, reduction_factor: int = 1) -> Tensor:
    """FastSpeech2Decoder.proc_duration in training mode (decoder/nar_tts.py:206-230)."""
    duration = torch.clamp(torch.round(duration), min=0)
    zero = duration == 0
    duration = torch.clamp(duration, min=round(min_frame_num / reduction_factor),
                           max=None if max_frame_num is None else round(max_frame_num / reduction_factor))
    return torch.where(zero, torch.zeros_like(duration), duration)


def length_regulate(enc_text: Tensor, used_duration: Tensor, out_len: Tensor) -> Tensor:
    """The expansion loop of FastSpeech2Decoder.forward (decoder/nar_tts.py:439-472): row k of utterance i repeated used_duration[i][k]
    times, then zero-padded / cut to max(out_len)."""
    Tm = int(out_len.max())
    rows = []
    for i in range(enc_text.size(0)):
        e = enc_text[i].repeat_interleave(used_duration[i].long(), dim=0)
        if e.size(0) < Tm:
            e = F.pad(e, (0, 0, 0, Tm - e.size(0)))
        rows.append(e[:Tm])
    return torch.stack(rows)


def least_error(pred: Tensor, tgt: Tensor, tgt_len: Tensor, loss_type: str = "L2", is_normalized: bool = True) -> Tensor:
    """LeastError.__call__ (criterion/least_error.py:66-124)."""
    if pred.dim() == 2:
        pred = pred.unsqueeze(-1)
    if tgt.dim() == 2:
        tgt = tgt.unsqueeze(-1)
    mask = make_mask_from_len(tgt_len, max_len=pred.size(1), return_3d=False)
    d = pred - tgt
    loss = {"L1": d.abs(), "L2": d * d, "L1+L2": d.abs() + d * d}[loss_type].mean(dim=-1)
    loss = loss.masked_fill(~mask, 0.0)
    return loss.sum() / mask.sum() if is_normalized else loss.sum(dim=-1).mean()


def fastspeech2_forward_loss(text: Tensor, text_len: Tensor, duration: Tensor, wave: Tensor, wave_len: Tensor, pitch: Tensor,
                             pitch_len: Tensor, P: Dict[str, Tensor], cfg: Dict, norms: Dict[str, "FeatureNormalization"], epoch: int = 1,
                             training: bool = True):
    """TTSEncoder.forward (encoder/tts.py:66-86) -> FastSpeech2Decoder.forward in training mode with ground-truth duration / pitch /
    energy (decoder/nar_tts.py:232-502, reduction_factor 1, no speaker embedding) -> the loss of FastSpeech2.criterion_forward
    (model/nar_tts.py:472-560).  `P` holds 'encoder.*' / 'decoder.*' state-dict entries, `norms` the three stateful normalisations
    ('feat', 'pitch', 'energy').  Returns a dict of outputs, loss terms and the BatchNorm running statistics after the call."""
    d, H, L = cfg["d"], cfg["heads"], cfg["layers"]
    trf = dict(num_layers=L, num_heads=H, d_model=d, emb_scale=False, activation="ReLU", fdfwd_type="conv", layernorm_first=True)
    emb = F.embedding(text, P["encoder.embedding.embed.weight"], padding_idx=0)
    e_mask = make_mask_from_len(text_len, max_len=text.size(1))
    enc_text, e_mask, _, _ = transformer_encoder(emb, e_mask, P, "encoder.encoder.", **trf)
    # acoustic targets: frontend with energy (nar_tts.py:275-278), then the three normalisations (:280-283, :300-303, :320-323)
    feat, feat_len, energy, energy_len = speech2mel(wave.squeeze(-1), wave_len, sr=22050, mag_spec=True, hop_length=256, win_length=1024, n_mels=80,
                                                    fmin=0, fmax=8000, log_base=None, clamp=1e-5, return_energy=True)
    feat, feat_len = norms["feat"](feat, feat_len, epoch=epoch)
    pitch, pitch_len = norms["pitch"](pitch, pitch_len, epoch=epoch)
    energy, energy_len = norms["energy"](energy, energy_len, epoch=epoch)
    dp = "decoder."
    pred_dur = var_predictor(enc_text, P, dp + "duration_predictor.", cfg["dur_layers"])
    used = proc_duration(duration / duration.sum(dim=-1, keepdim=True) * feat_len.unsqueeze(-1))
    pred_pitch = var_predictor(enc_text, P, dp + "pitch_predictor.", cfg["pitch_layers"])
    tok_pitch = average_scalar_by_duration(pitch, used)
    pred_energy = var_predictor(enc_text, P, dp + "energy_predictor.", cfg["energy_layers"])
    tok_energy = average_scalar_by_duration(energy, used)
    x = enc_text + var_embed(tok_pitch, P, dp + "pitch_predictor.") + var_embed(tok_energy, P, dp + "energy_predictor.")
    x = length_regulate(x, used, feat_len)
    dec, dmask, _, _ = transformer_encoder(x, make_mask_from_len(feat_len), P, dp + "decoder.", **trf)
    before = F.linear(dec, P[dp + "feat_pred.weight"], P[dp + "feat_pred.bias"])
    post, bn_stats = conv1d_postnet(before, P, dp + "postnet.", cfg["post_layers"], training)
    after = before + post
    terms = dict(feat_before=least_error(before, feat, feat_len, "L1"), feat_after=least_error(after, feat, feat_len, "L1"),
                 pitch=least_error(pred_pitch, tok_pitch, text_len, "L2"), energy=least_error(pred_energy, tok_energy, text_len, "L2"),
                 duration=least_error(pred_dur, torch.log(used.float() + 1), text_len, "L2"))
    out = dict(enc_text=enc_text, before=before, after=after, tgt_feat=feat, tgt_feat_len=feat_len, pred_pitch=pred_pitch, tgt_pitch=tok_pitch,
               pred_energy=pred_energy, tgt_energy=tok_energy, pred_dur=pred_dur, used_dur=used, loss=sum(terms.values()), bn_stats=bn_stats)
    out.update({"loss_" + k: v for k, v in terms.items()})
    return out


# --------------------------------------------------------------------------------------------------
# f4  recover(): mel -> linear -> waveform    speechain/module/frontend/linear2mel.py:191-224, speech2linear.py:252-351
# --------------------------------------------------------------------------------------------------


def linear2mel_recover(feat: Tensor, feat_len: Tensor, mel_w: Tensor, logging: bool = True, log_base: Optional[float] = 10.0) -> Tensor:
    """LinearSpec2MelSpec.recover (linear2mel.py:191-224): undo the log, least-squares solve against the (n_mels, n_bins) filterbank
    (driver 'gels': the minimum-norm solution of the under-determined system), zero the padded frames of shorter utterances, clamp."""
    x = feat
    if logging:
        x = torch.pow(torch.full_like(x, math.e if log_base is None else log_base), x)
    x = torch.linalg.lstsq(mel_w[None], x.transpose(-2, -1), driver="gels").solution.transpose(-2, -1)
    x = x.clone()
    for i in range(len(feat_len)):
        if feat_len[i] != feat_len.max():
            x[i, feat_len[i]:] = 0
    return torch.clamp(x, min=1e-10)


def griffin_lim(mag: Tensor, window: Tensor, n_fft: int, hop: int, win_length: int, n_iter: int, momentum: float, init_angles: Tensor) -> Tensor:
    """torchaudio.functional.griffinlim as published (torchaudio is absent: this restates the third-party algorithm the reference calls at
    speech2linear.py:279-293 -- parity unpinned there): mag (B, T, F) magnitudes, init_angles (B, T, F) complex -> (B, hop (T - 1))."""
    S = mag.transpose(1, 2).to(torch.complex64)  # (B, F, T)
    angles = init_angles.transpose(1, 2).to(torch.complex64)
    tprev = torch.zeros_like(angles)
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win_length, window=window)
    for _ in range(n_iter):
        inv = torch.istft(S * angles, **kw)
        rebuilt = torch.stft(inv, **kw, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        angles = rebuilt
        if momentum:
            angles = angles - tprev * (momentum / (1 + momentum))
        angles = angles / (angles.abs() + 1e-16)
        tprev = rebuilt
    return torch.istft(S * angles, **kw)


def speech2linear_recover_post(wav: Tensor, feat_len: Tensor, hop: int, preemph: Optional[float], winlen: int = 100):
    """What Speech2LinearSpec.recover does after the Griffin-Lim call (speech2linear.py:294-351): lengths, the sliding-window inverse
    pre-emphasis X[n] = sum_{k < winlen} c^k Y[n - k], silence beyond each utterance.  wav (B, L) -> ((B, L, 1), wav_len)."""
    wav_len = (feat_len - 1) * hop
    assert int(wav_len.max()) == wav.size(1)
    if preemph is not None:
        w = torch.pow(torch.full((winlen,), preemph), torch.arange(winlen - 1, -1, -1)).reshape(1, 1, -1)
        wav = F.conv1d(F.pad(wav.unsqueeze(1), (winlen - 1, 0)), w).squeeze(1)
    wav = wav.clone()
    for i in range(len(wav_len)):
        wav[i, wav_len[i]:] = 0
    return wav.unsqueeze(-1), wav_len


# --------------------------------------------------------------------------------------------------
# speaker embeddings of the TTS decoder   speechain/module/prenet/spk_embed.py:7-230; used at decoder/nar_tts.py:257-263
# --------------------------------------------------------------------------------------------------

def speaker_embed(P: Dict[str, Tensor], prefix: str, enc_output: Tensor, spk_ids: Optional[Tensor] = None, spk_feat: Optional[Tensor] = None,
                  dec_input: Optional[Tensor] = None, spk_emb_comb: str = "concat", dec_comb: bool = False, encdec_same_proj: bool = True):
    """SpeakerEmbedPrenet.forward (:104-153) followed by combine_spk_feat (:155-227).  Which branches exist is read off the parameters, as the
    module's ``hasattr`` checks do: ``spk_lookup.embed.weight`` (closed-set table, row 0 = padding), ``pre_add_proj_lookup`` /
    ``pre_add_proj_pretrain`` (only with 'add' and a dimension other than d_model), ``final_proj_enc`` and ``final_proj_dec``.
    Both vectors are L2-normalised (F.normalize, eps 1e-12) before they are combined; with 'concat' the pretrained vector comes first, then
    the lookup one (:204-213).  Returns (spk_feat_lookup, spk_feat, enc_output, dec_input)."""
    look = None
    if prefix + "spk_lookup.embed.weight" in P:
        look = F.embedding(spk_ids, P[prefix + "spk_lookup.embed.weight"], padding_idx=0)
        if prefix + "pre_add_proj_lookup.weight" in P:
            look = F.linear(look, P[prefix + "pre_add_proj_lookup.weight"], P[prefix + "pre_add_proj_lookup.bias"])
        look = F.normalize(look, dim=-1)
    pre = None
    if spk_feat is not None:
        pre = spk_feat
        if prefix + "pre_add_proj_pretrain.weight" in P:
            pre = F.linear(pre, P[prefix + "pre_add_proj_pretrain.weight"], P[prefix + "pre_add_proj_pretrain.bias"])
        pre = F.normalize(pre, dim=-1)
    pre3 = None if pre is None else (pre.unsqueeze(1) if pre.dim() == 2 else pre)
    look3 = None if look is None else (look.unsqueeze(1) if look.dim() == 2 else look)

    def combine(tgt, name):
        if spk_emb_comb == "add":
            if pre3 is not None:
                tgt = tgt + pre3
            if look3 is not None:
                tgt = tgt + look3
        else:
            if pre3 is not None:
                tgt = torch.cat([tgt, pre3.expand(-1, tgt.size(1), -1)], dim=-1)
            if look3 is not None:
                tgt = torch.cat([tgt, look3.expand(-1, tgt.size(1), -1)], dim=-1)
        return F.linear(tgt, P[prefix + name + ".weight"], P[prefix + name + ".bias"])

    enc_output = combine(enc_output, "final_proj_enc")
    if dec_comb:
        dec_input = combine(dec_input, "final_proj_enc" if encdec_same_proj else "final_proj_dec")
    return look, pre, enc_output, dec_input

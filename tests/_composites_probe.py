"""Helper of tests/test_boundary_composites.py, run as a child process (it rewires ``sys.modules``): builds the REFERENCE's composite classes
-- ASREncoder, ARASRDecoder, TTSEncoder, FastSpeech2Decoder -- from recipe dictionaries and prints their state-dict keys and shapes as JSON.
argv[1] = "plain": an unmodified SpeeChain checkout;  "leaf": after ``speechain_amd.register.register(leaf_only=True)``, i.e. the reference's
own composites on top of the overlaid leaf modules.  Build container only (needs /root/reference; torchaudio / GPUtil stubbed as in
tests/golden/make_golden.py)."""
import copy
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import melscale  # noqa: E402

ta = types.ModuleType("torchaudio")
ta.functional = types.ModuleType("torchaudio.functional")
ta.transforms = types.ModuleType("torchaudio.transforms")
ta.functional.melscale_fbanks = lambda n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk": torch.from_numpy(
    melscale.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm, mel_scale))
ta.transforms.GriffinLim = lambda **kw: None
sys.modules.update({"torchaudio": ta, "torchaudio.functional": ta.functional, "torchaudio.transforms": ta.transforms})
gp = types.ModuleType("GPUtil")
gp.GPU, gp.getGPUs = object, lambda: []
sys.modules["GPUtil"] = gp

mode = sys.argv[1]
if mode == "leaf":
    import speechain_amd.register as reg

    reg.register(leaf_only=True)

# recipes/asr/librispeech/train-clean-100/exp_cfg/100-bpe5k_transformer-large_lr2e-3.yaml:178-249 (d_model 512, 8 heads, ffn 2048, dropout 0.1)
trf = dict(posenc_dropout=0.1, posenc_scale=False, emb_layernorm=False, emb_scale=False, d_model=512, num_heads=8, att_dropout=0.1, fdfwd_dim=2048,
           fdfwd_activation="GELU", fdfwd_dropout=0.1, res_dropout=0.1, layernorm_first=True)
ASR_ENC = dict(frontend=dict(type="frontend.speech2mel.Speech2MelSpec", conf=dict(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)),
               normalize=True, specaug=dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5),
               prenet=dict(type="prenet.conv2d.Conv2dPrenet", conf=dict(conv_dims=[512, 512], conv_kernel=3, conv_stride=2, conv_batchnorm=True,
                                                                       conv_activation="LeakyReLU", lnr_dims=512)),
               encoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf, num_layers=12)))
ASR_DEC = dict(embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=512)),
               decoder=dict(type="transformer.decoder.TransformerDecoder", conf=dict(trf, num_layers=6)), vocab_size=5000)
# recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:146-249
ttrf = dict(posenc_dropout=0.2, posenc_scale=True, emb_scale=False, emb_layernorm=True, d_model=384, num_heads=2, num_layers=4, att_dropout=0.2,
            fdfwd_dim=1536, fdfwd_type="conv", fdfwd_args=dict(kernel_size=3), fdfwd_dropout=0.2, res_dropout=0.2, layernorm_first=True)
TTS_ENC = dict(embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=384)),
               encoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(ttrf)), vocab_size=75)
vp = "prenet.var_pred.Conv1dVarPredictor"
TTS_DEC = dict(
    input_size=384,
    duration_predictor=dict(type=vp, conf=dict(conv_dims=[256, -1], conv_kernel=3, conv_dropout=0.5)),
    pitch_predictor=dict(type=vp, conf=dict(conv_dims=[256, -1, -1, -1, -1], conv_kernel=5, conv_dropout=0.5, conv_emb_kernel=1, conv_emb_dropout=0.0)),
    energy_predictor=dict(type=vp, conf=dict(conv_dims=[256, -1], conv_kernel=3, conv_dropout=0.5, conv_emb_kernel=1, conv_emb_dropout=0.0)),
    feat_frontend=dict(type="frontend.speech2mel.Speech2MelSpec", conf=dict(sr=22050, mag_spec=True, hop_length=256, win_length=1024, n_mels=80,
                                                                           fmin=0, fmax=8000, log_base=None, clamp=1e-5)),
    feat_normalize=True, pitch_normalize=True, energy_normalize=True,
    decoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(ttrf)),
    postnet=dict(type="postnet.conv1d.Conv1dPostnet", conf=dict(conv_dims=[256, -1, -1, -1, 0], conv_kernel=5, conv_batchnorm=True,
                                                                 conv_activation="Tanh", conv_dropout=0.5)))

from speechain.module.decoder.ar_asr import ARASRDecoder  # noqa: E402  (the reference's: register(leaf_only=True) leaves the composites alone)
from speechain.module.decoder.nar_tts import FastSpeech2Decoder  # noqa: E402
from speechain.module.encoder.asr import ASREncoder  # noqa: E402
from speechain.module.encoder.tts import TTSEncoder  # noqa: E402

out = {"origin": {}}
for name, cls, conf in (("ASREncoder", ASREncoder, ASR_ENC), ("ARASRDecoder", ARASRDecoder, ASR_DEC), ("TTSEncoder", TTSEncoder, TTS_ENC),
                        ("FastSpeech2Decoder", FastSpeech2Decoder, TTS_DEC)):
    assert cls.__module__.startswith("speechain."), cls.__module__  # the composite itself is the reference's in both modes
    torch.manual_seed(0)
    m = cls(**copy.deepcopy(conf))
    out[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
    out[name + ".output_size"] = getattr(m, "output_size", None)
    out["origin"][name] = sorted({type(c).__module__.split(".")[0] for c in m.modules() if type(c).__module__.split(".")[0] in ("speechain", "speechain_amd")})
print(json.dumps(out))

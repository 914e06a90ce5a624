"""GPU: the FastSpeech2 path (BASELINE configs[4]) -- TTSEncoder + FastSpeech2Decoder (variance predictors, duration
handling, average_scalar_by_duration, length regulation, decoder stack, Conv1d postnet with BatchNorm1d + Tanh) + the five LeastError
terms -- against the fixture captured from the imported reference (tests/golden/g10_tts.npz) and against the CPU oracle."""
import copy
import math
import os
import sys

import pytest
import torch

from oracle import cpu_ref as R
from speechain_amd import functional as Fn

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

SMALL = dict(d=32, heads=2, layers=2, ffn=64, pred=16, post=16, vocab=20)


def small_conf(c=SMALL, dropout=0.0):
    """The module_conf of recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:150-249 at the fixture's sizes (the same
    dictionaries tests/golden/make_golden.py::tts_confs hands to the reference)."""
    trf = dict(posenc_dropout=dropout, posenc_scale=True, emb_scale=False, emb_layernorm=True, d_model=c["d"], num_heads=c["heads"],
               num_layers=c["layers"], att_dropout=dropout, fdfwd_dim=c["ffn"], fdfwd_type="conv", fdfwd_args=dict(kernel_size=3),
               fdfwd_dropout=dropout, res_dropout=dropout, layernorm_first=True, posenc_maxlen=400)
    pred = lambda n, k, emb: dict(type="prenet.var_pred.Conv1dVarPredictor",
                                  conf=dict(conv_dims=[c["pred"]] + [-1] * (n - 1), conv_kernel=k, conv_dropout=dropout,
                                            **(dict(conv_emb_kernel=1, conv_emb_dropout=0.0) if emb else {})))
    return dict(
        enc_emb=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=c["d"])),
        encoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf)),
        duration_predictor=pred(2, 3, False), pitch_predictor=pred(3, 5, True), energy_predictor=pred(2, 3, True),
        feat_frontend=dict(type="frontend.speech2mel.Speech2MelSpec", conf=dict(sr=22050, mag_spec=True, hop_length=256, win_length=1024, n_mels=80,
                                                                                fmin=0, fmax=8000, log_base=None, clamp=1e-5)),
        feat_normalize=True, pitch_normalize=True, energy_normalize=True,
        decoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf)),
        dec_postnet=dict(type="postnet.conv1d.Conv1dPostnet", conf=dict(conv_dims=[c["post"], -1, 0], conv_kernel=5, conv_batchnorm=True,
                                                                         conv_activation="Tanh", conv_dropout=dropout)))


SEL = {"g_emb": "encoder.embedding.embed.weight", "g_enc0_ffn_in": "encoder.encoder.trfm_layers.0.feed_forward.in_layer.conv_lyr.weight",
       "g_dur_conv0": "decoder.duration_predictor.conv.0.conv_lyr.weight", "g_dur_ln0": "decoder.duration_predictor.conv.2.weight",
       "g_dur_lin": "decoder.duration_predictor.linear.weight", "g_pitch_emb_w": "decoder.pitch_predictor.conv_emb.0.conv_lyr.weight",
       "g_pitch_emb_b": "decoder.pitch_predictor.conv_emb.0.conv_lyr.bias", "g_energy_conv1": "decoder.energy_predictor.conv.4.conv_lyr.weight",
       "g_dec1_q": "decoder.decoder.trfm_layers.1.multihead_att.q_layer.weight", "g_dec_alpha": "decoder.decoder.posenc.alpha",
       "g_feat_pred": "decoder.feat_pred.weight", "g_post0": "decoder.postnet.conv.0.conv_lyr.weight",
       "g_post_bn1_w": "decoder.postnet.conv.1.weight", "g_post_last": "decoder.postnet.conv.8.conv_lyr.weight",
       "g_post_bn_last_b": "decoder.postnet.conv.9.bias"}


@pytest.mark.parametrize("mode", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_fastspeech2_two_steps_vs_reference_fixture(golden, mode):
    """The reference's state dict loads into the drop-in modules key for key; two training steps (SGD move on fifteen parameters in
    between, as the fixture's generator made) reproduce its outputs, loss terms, gradients and buffers.  f32 mode: 5e-4 abs on
    activations, 1e-4 rel on losses, 2e-3 of each gradient's scale; bf16: rounding-level bounds."""
    from speechain_amd.harness import TTSFastSpeech2

    g = golden("g10_tts")
    f32 = mode == torch.float32
    Fn.set_compute_dtype(mode)
    try:
        torch.manual_seed(0)
        model = TTSFastSpeech2(SMALL["vocab"], small_conf()).train()
        sd = {k[len("sd0."):]: v for k, v in g.items() if k.startswith("sd0.")}
        res = model.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        allowed = ("posenc.posenc", "mel_fbanks", "preemph")  # constants the fixture leaves out (rebuilt identically by the constructors)
        assert all(any(a in k for a in allowed) for k in res.missing_keys), res.missing_keys
        model = model.cuda()
        params = dict(model.named_parameters())
        dev = lambda t: t.cuda()
        for step in range(2):
            t = f"s{step}."
            out = model(dev(g[t + "text"]), dev(g["text_len"]), dev(g[t + "dur"]), dev(g["text_len"]), dev(g[t + "wave"]), dev(g["wave_len"]),
                        dev(g[t + "pitch"]), dev(g["pitch_len"]), epoch=1)
            assert torch.equal(out["tgt_feat_len"].cpu(), g[t + "tgt_feat_len"])
            assert torch.equal(out["used_dur"].cpu().float(), g[t + "used_dur"].float())
            act = dict(atol=5e-4, rtol=1e-3) if f32 else dict(atol=0.12, rtol=5e-2)
            for k in ("tgt_feat", "tgt_pitch", "tgt_energy"):  # targets: frontend + normalisation + per-token averages (float32 in both modes)
                assert torch.allclose(out[k].float().cpu(), g[t + k], atol=2e-3, rtol=1e-3), (step, k, float((out[k].float().cpu() - g[t + k]).abs().max()))
            for k in ("enc_text", "before", "after", "pred_pitch", "pred_energy", "pred_dur"):
                assert torch.allclose(out[k].float().cpu(), g[t + k], **act), (step, k, float((out[k].float().cpu() - g[t + k]).abs().max()))
            for k in ("feat_before", "feat_after", "pitch", "energy", "duration"):
                a, b = float(out["loss_" + k]), float(g[t + "loss_" + k])
                assert abs(a - b) < (2e-4 if f32 else 3e-2) * max(1.0, abs(b)), (step, k, a, b)
            model.zero_grad(set_to_none=True)
            out["loss"].backward()
            for name, key in SEL.items():
                gr, ref = params[key].grad.float().cpu(), g[t + name]
                scale = float(ref.abs().max())
                # (bf16: a sanity bound only -- small gradients at the end of long chains, and sums with cancellation such as the positional
                # encoding scale, carry rounding errors of the size of the value itself; the f32 mode is the parity check)
                tol = (2e-3 if f32 else 0.3) * scale + (1e-6 if f32 else 5e-4)
                assert float((gr - ref).abs().max()) < tol, (step, name, float((gr - ref).abs().max()), scale)
            if step == 0:
                with torch.no_grad():
                    for key in SEL.values():
                        params[key] -= 0.05 * params[key].grad
                Fn.weights.invalidate()
        bufs = dict(model.named_buffers())
        for k, ref in g.items():
            if not k.startswith("buf1."):
                continue
            name = k[len("buf1."):]
            assert name in bufs, name
            got = bufs[name].float().cpu().reshape(-1)
            if name.endswith("num_batches_tracked") or name.endswith("_batch"):
                assert torch.equal(got, ref.float().reshape(-1)), name
            else:
                assert torch.allclose(got, ref.float().reshape(-1), atol=(2e-4 if f32 else 3e-2), rtol=(1e-3 if f32 else 5e-2)), \
                    (name, float((got - ref.float().reshape(-1)).abs().max()))
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_length_regulation_and_its_adjoint(dtype):
    """sc_length_regulate_fwd / _bwd against the reference's loop restated in the oracle: tokens without frames (also leading and
    trailing), durations that overshoot the output length (cut) and undershoot it (zero rows), D not a multiple of 8."""
    gen = torch.Generator().manual_seed(5)
    for (B, L, D, T) in [(3, 9, 40, 37), (2, 130, 384, 640), (1, 5, 7, 12)]:
        dur = torch.randint(0, 7, (B, L), generator=gen).float()
        dur[0, 0] = 0
        dur[-1, -2:] = 0
        dur[0, 3] = 9
        x = torch.randn(B, L, D, generator=gen).to(dtype)
        w = torch.randn(B, T, D, generator=gen)
        xr = x.float().clone().requires_grad_()
        ref = R.length_regulate(xr, dur, torch.full((B,), T))
        (ref * w).sum().backward()
        xg = x.cuda().requires_grad_()
        out = Fn.length_regulate(xg, dur.cuda(), T)
        assert out.shape == (B, T, D) and torch.equal(out.float().cpu(), ref.detach())  # a copy: exact in either dtype
        (out.float() * w.cuda()).sum().backward()
        tol = dict(atol=1e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=0.05, rtol=2e-2)
        # (bf16: the incoming gradient is rounded once, the per-token sum is accumulated in float32 and rounded once)
        gin = (w.to(dtype).float() if dtype == torch.bfloat16 else w)
        ref_g = torch.autograd.grad((R.length_regulate(xr, dur, torch.full((B,), T)) * gin).sum(), xr)[0]
        assert torch.allclose(xg.grad.float().cpu(), ref_g, **tol), float((xg.grad.float().cpu() - ref_g).abs().max())


def test_average_by_duration(golden):
    g = golden("g10_tts")
    out = Fn.avg_by_duration(g["avg_frame"].cuda(), g["avg_dur"].cuda())
    assert torch.allclose(out.cpu(), g["avg_out"], atol=1e-6)
    gen = torch.Generator().manual_seed(9)
    frame = torch.randn(4, 200, generator=gen)
    dur = torch.randint(0, 9, (4, 60), generator=gen).float()
    assert torch.allclose(Fn.avg_by_duration(frame.cuda(), dur.cuda()).cpu(), R.average_scalar_by_duration(frame, dur), atol=1e-5)


@pytest.mark.parametrize("mode", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv1d_postnet_eval_and_train_vs_oracle(mode):
    """Conv1dPostnet (5 blocks as in the recipe, kernel 5, BatchNorm1d + Tanh) on a padded batch: training mode (batch statistics over
    all B*T positions, running statistics updated) and eval mode (running statistics) against F.conv1d / F.batch_norm / tanh."""
    from speechain_amd.module.postnet.conv1d import Conv1dPostnet

    torch.manual_seed(3)
    post = Conv1dPostnet(input_size=80, conv_dims=[48, -1, -1, -1, 0], conv_kernel=5, conv_batchnorm=True, conv_activation="Tanh", conv_dropout=0.0)
    for p in post.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
        else:
            torch.nn.init.normal_(p, 1.0 if p.dim() == 1 and p.mean() > 0.5 else 0.0, 0.2)
    x = torch.randn(3, 29, 80)
    P = {"conv." + k: v.detach().clone() for k, v in post.conv.state_dict().items()}
    f32 = mode == torch.float32
    tol = dict(atol=3e-4, rtol=1e-3) if f32 else dict(atol=6e-2, rtol=5e-2)
    Fn.set_compute_dtype(mode)
    try:
        g = post.cuda().train()
        y = g(x.cuda(), torch.tensor([29, 20, 11]).cuda())
        ref, stats = R.conv1d_postnet(x, P, "", 5, True)
        assert torch.allclose(y.float().cpu(), ref, **tol), float((y.float().cpu() - ref).abs().max())
        for k, v in stats.items():
            assert torch.allclose(dict(g.named_buffers())[k].cpu(), v, atol=(1e-5 if f32 else 2e-2), rtol=(1e-4 if f32 else 5e-2)), k
        P.update(stats)
        y2 = g.eval()(x.cuda(), torch.tensor([29, 20, 11]).cuda())
        if f32:  # (bf16 running statistics differ at rounding level from the oracle's, checked above)
            ref2, _ = R.conv1d_postnet(x, P, "", 5, False)
            assert torch.allclose(y2.float().cpu(), ref2, **tol), float((y2.float().cpu() - ref2).abs().max())
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()


@pytest.mark.gpu
@pytest.mark.parametrize("T,k", [(50, 3), (128, 3), (37, 5)])
def test_conv_feed_forward_without_im2col_equals_the_im2col_path(monkeypatch, T, k):
    """functional._ConvLinear (one product over overlapping rows of a zero-padded image, data gradient against the reversed taps,
    weight gradient from overlapping-row views in the grouped launch) against shifted_cat + linear on the conv-type feed-forward of the
    FastSpeech2 recipes: identical outputs (same arithmetic, same dropout indices); gradients within bf16 rounding (the old data gradient
    rounds each tap's contribution to bf16 before col2im adds them)."""
    from speechain_amd import functional as Fn
    from speechain_amd.ddp import GradBucketReducer
    from speechain_amd.module.transformer.feed_forward import PositionwiseFeedForward

    Fn.set_compute_dtype(torch.bfloat16)
    try:
        B, d, dff = 3, 64, 256
        gen = torch.Generator().manual_seed(5)
        x0 = torch.randn(B, T, d, generator=gen).cuda()
        res0 = torch.randn(B, T, d, generator=gen).cuda()
        dy = torch.randn(B, T, d, generator=gen).cuda().bfloat16()

        def run(direct: bool):
            monkeypatch.setattr(Fn, "_CONV_DIRECT", direct)
            Fn.GradSinks.clear()
            Fn.weights.invalidate()
            torch.manual_seed(3)
            ff = PositionwiseFeedForward(d_model=d, fdfwd_dim=dff, fdfwd_type="conv", fdfwd_activation="ReLU", fdfwd_args={"kernel_size": k},
                                         dropout=0.1).cuda().train()
            red = GradBucketReducer(ff.parameters(), bucket_mb=0.25)
            outs = []
            for step in range(2):  # twice: no stale slots, no double counting
                if step == 1:  # a raw in-place update (as FusedAdam makes) + the batched refresh of every cached weight copy
                    for p in ff.parameters():
                        p.data.mul_(1.05)
                    Fn.weights.refresh()
                Fn.DropoutState.reset(1234)
                Fn.DropoutState.advance()
                red.zero_grad()
                x = x0.clone().bfloat16().requires_grad_(True)
                y = ff(x, _residual=res0.bfloat16(), _res_drop=0.1)
                y.backward(dy)
                red.wait()
                torch.cuda.synchronize()
                outs.append((y.detach().float().clone(), x.grad.float().clone(), {n: p.grad.detach().clone() for n, p in ff.named_parameters()}))
            return outs

        a, b = run(True), run(False)
        for (ya, dxa, ga), (yb, dxb, gb) in zip(a, b):
            assert torch.equal(ya, yb), float((ya - yb).abs().max())
            assert torch.allclose(dxa, dxb, atol=3e-2 * float(dxb.abs().max()), rtol=3e-2), float((dxa - dxb).abs().max())
            for n in ga:
                # out_layer: both paths contract the same gradient over the same rows (f32 sums in another tile order); in_layer: its
                # incoming gradient is the out_layer's data gradient, which the two paths round differently (above)
                tol = 2e-3 if n.startswith("out_layer") else 1.5e-2
                assert torch.allclose(ga[n], gb[n], atol=tol * float(gb[n].abs().max()) + 1e-6, rtol=tol), (n, float((ga[n] - gb[n]).abs().max()))
        assert not torch.equal(a[0][0], a[1][0])  # (the second step ran on the updated weights)
    finally:
        Fn.GradSinks.clear()
        Fn.set_compute_dtype(None)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,r", [(3, 50, 64, 1), (2, 7, 8, 2), (4, 128, 384, 1)])
def test_pad_rows_through_the_c_abi(B, T, C, r):
    """sc_pad_rows / sc_zero_pad_rows against a torch construction of the image: r zero rows on either side of every utterance, zero rows
    behind the last one; the second entry point writes exactly those zero rows and leaves the data rows alone."""
    from speechain_amd._lib import lib, stream

    gen = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, T, C, generator=gen).cuda().bfloat16()
    rows = (B * (T + 2 * r) + 31) // 32 * 32 + 32
    want = torch.zeros(rows, C, dtype=torch.bfloat16, device="cuda")
    want[: B * (T + 2 * r)].view(B, T + 2 * r, C)[:, r: r + T] = x
    out = torch.full((rows, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    lib.call("sc_pad_rows", Fn._p(x), B, T, C, r, Fn._p(out), rows, stream())
    assert torch.equal(out, want)
    out2 = torch.full((rows, C), 7.0, dtype=torch.bfloat16, device="cuda")
    lib.call("sc_zero_pad_rows", B, T, C, r, Fn._p(out2), rows, stream())
    keep = torch.zeros(rows, dtype=torch.bool, device="cuda")
    keep[: B * (T + 2 * r)].view(B, T + 2 * r)[:, r: r + T] = True
    assert torch.equal(out2[keep], torch.full_like(out2[keep], 7.0)) and torch.equal(out2[~keep], torch.zeros_like(out2[~keep]))


@pytest.mark.gpu
@pytest.mark.parametrize("loss_type", ["L1", "L2", "L1+L2"])
@pytest.mark.parametrize("normalized", [True, False])
def test_fused_least_error_equals_the_elementwise_form(loss_type, normalized):
    """LeastError through sc_least_error_fwd / _bwd (the GPU path of criterion/least_error.py) against the reference's sequence of
    elementwise steps (least_error.py:96-129, run here on the same device through `update_range`-free torch arithmetic): value within
    1e-6 relative, gradient within 1e-6 of its scale (f32) / bf16 rounding; ragged lengths, an empty utterance, D = 1 (pitch / energy / duration tracks) and D = 80 (mel)."""
    from speechain_amd.criterion.least_error import LeastError
    from speechain_amd.utilbox.train_util import make_mask_from_len

    def reference(pred, tgt, tgt_len):
        pred, tgt = pred.float(), tgt.float()
        mask = make_mask_from_len(tgt_len, max_len=pred.size(1), return_3d=False)
        d = pred - tgt
        loss = {"L1": d.abs, "L2": lambda: d * d, "L1+L2": lambda: d.abs() + d * d}[loss_type]().mean(dim=-1)
        loss = loss.masked_fill(~mask, 0.0)
        return loss.sum() / mask.sum() if normalized else loss.sum(dim=-1).mean()

    crit = LeastError(loss_type=loss_type, is_normalized=normalized)
    gen = torch.Generator().manual_seed(3)
    for (B, T, D), dtype in (((4, 37, 80), torch.float32), ((4, 37, 80), torch.bfloat16), ((5, 129, 1), torch.float32), ((3, 640, 80), torch.bfloat16)):
        tgt = torch.randn(B, T, D, generator=gen).cuda()
        lens = torch.randint(1, T + 1, (B,), generator=gen)
        lens[0], lens[-1] = T, 0  # the whole track; empty
        lens = lens.cuda()
        p0 = (torch.randn(B, T, D, generator=gen).cuda() * 2).to(dtype)
        p0[1, 2, 0] = tgt[1, 2, 0].to(dtype)  # (an exact zero difference where representable: |x|' = 0 there)
        a = p0.clone().requires_grad_(True)
        b = p0.clone().requires_grad_(True)
        got = crit(pred=a if D > 1 else a.squeeze(-1), tgt=tgt if D > 1 else tgt.squeeze(-1), tgt_len=lens)
        want = reference(b, tgt, lens)
        assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)) + 1e-7, (float(got), float(want))
        (got * 3.0).backward()
        (want * 3.0).backward()
        scale = float(b.grad.float().abs().max())
        tol = (1e-6 if dtype == torch.float32 else 1e-2) * scale
        assert float((a.grad.float() - b.grad.float()).abs().max()) <= tol, float((a.grad.float() - b.grad.float()).abs().max())
        assert float(a.grad[-1].float().abs().max()) == 0.0  # the empty utterance


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_positional_encoding_scale_gradient(dtype):
    """The gradient of ScaledPositionalEncoding's scalar alpha (pos_enc.py: x + alpha * table) through sc_dot_table against the torch
    form sum(dy * table) -- (B, T, D) with T not a multiple of anything and D = 384 (the FastSpeech2 width) and 20."""
    gen = torch.Generator().manual_seed(11)
    for B, T, D in ((3, 37, 384), (2, 50, 20)):
        table = torch.randn(1, 64, D, generator=gen).cuda()
        x = torch.randn(B, T, D, generator=gen).cuda().to(dtype).requires_grad_(True)
        alpha = torch.nn.Parameter(torch.tensor(0.7).cuda())
        w = torch.randn(B, T, D, generator=gen).cuda().to(dtype)
        y = Fn.posenc_add(x, table[:, :T], alpha, 1.0, 0.0)
        (y.float() * w.float()).sum().backward()
        want = (w.float() * table[0, :T]).sum()
        assert abs(float(alpha.grad) - float(want)) <= (1e-5 if dtype == torch.float32 else 2e-2) * (abs(float(want)) + 1.0), (float(alpha.grad), float(want))
        assert torch.allclose(x.grad.float(), w.float(), atol=1e-6 if dtype == torch.float32 else 1e-2)

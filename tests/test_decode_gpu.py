"""GPU: the inference callers of SURVEY.md row f3 through the HIP path -- cached-key/value decoding (sc_attention_decode,
sc_beam_table_update behind TransformerDecoder.forward_step / ARASRDecoder), CTC prefix scores (sc_ctc_prefix_scores / _state behind
CTCPrefixScorer) and beam_searching -- against the golden vectors of the imported reference (tests/golden/g11_decode.npz) and the oracle."""
import math

import pytest
import torch

from conftest import sub

pytestmark = pytest.mark.gpu

V = 23
DEC_CONF = dict(vocab_size=V, embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=128)),
                decoder=dict(type="transformer.decoder.TransformerDecoder",
                             conf=dict(num_heads=2, num_layers=2, fdfwd_dim=128, fdfwd_activation="GELU", posenc_dropout=0.0,
                                       att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0, emb_scale=False, posenc_maxlen=100)))
CASES = dict(a=dict(beam_size=4, min_f2t_ratio=1.5),
             b=dict(beam_size=4, min_f2t_ratio=-8, ctc_weight=0.3, temperature=1.3, length_penalty=1.2, ctc_temperature=1.1),
             c=dict(beam_size=3, min_f2t_ratio=2.0, eos_filtering=True, eos_threshold=1.2),
             d=dict(beam_size=2, min_f2t_ratio=-9, ctc_weight=0.5))


def build(g):
    from speechain_amd.module.decoder.ar_asr import ARASRDecoder
    from speechain_amd.module.postnet.token import TokenPostnet

    dec = ARASRDecoder(**DEC_CONF).cuda().eval()
    missing, unexpected = dec.load_state_dict({k: v.float() for k, v in sub(g, "sd.decoder.").items()}, strict=False)
    assert not unexpected and all(m.endswith("posenc.posenc") for m in missing), (missing, unexpected)
    ctc = TokenPostnet(input_size=128, vocab_size=V).cuda().eval()
    ctc.load_state_dict({k: v.float() for k, v in sub(g, "sd.ctc_layer.").items()})
    return dec, ctc


def test_attention_decode_against_dense_attention():
    """sc_attention_decode + sc_beam_table_update over 70 steps with random beam re-orderings: the context of every row equals dense
    softmax attention over that row's own history (gathered on the host side of the test), f32 and bf16."""
    from speechain_amd import functional as Fn

    gen = torch.Generator().manual_seed(3)
    N, H, d, Tmax = 12, 2, 128, 80
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        cache = torch.zeros(N, Tmax, 2 * d, dtype=dtype, device="cuda")
        tables = [torch.zeros(N, Tmax, dtype=torch.int32, device="cuda") for _ in range(2)]
        hist_k = torch.zeros(N, 0, d)
        hist_v = torch.zeros(N, 0, d)
        beam_idx = None
        for pos in range(70):
            qkv = torch.randn(N, 3 * d, generator=gen).to(dtype)
            t_out = tables[pos & 1]
            Fn.beam_table_update(tables[pos & 1 ^ 1] if pos else None, t_out, beam_idx, pos)
            out = Fn.attention_decode(qkv.cuda(), cache, t_out, pos, H, 1.0 / math.sqrt(d))
            if beam_idx is not None:
                hist_k, hist_v = hist_k[beam_idx.cpu()], hist_v[beam_idx.cpu()]
            q, k, v = qkv.float().split(d, dim=1)
            hist_k, hist_v = torch.cat([hist_k, k[:, None]], 1), torch.cat([hist_v, v[:, None]], 1)
            qh = q.view(N, H, 1, 64)
            kh, vh = hist_k.view(N, -1, H, 64).transpose(1, 2), hist_v.view(N, -1, H, 64).transpose(1, 2)
            want = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), -1) @ vh).reshape(N, d)
            assert float((out.float().cpu() - want).abs().max()) < tol, (dtype, pos, float((out.float().cpu() - want).abs().max()))
            beam_idx = torch.randint(0, N, (N,), generator=gen).cuda() if pos % 3 else None  # (every third step keeps the order)


@pytest.mark.parametrize("dtype", [None, torch.bfloat16], ids=["f32", "bf16"])
def test_cached_decoder_step_equals_prefix_decoding(golden, dtype):
    """ARASRDecoder.forward_step fed token by token (two beams per sentence, same tokens) gives, at every position, the logits
    ``forward`` gives on the prefix -- and, in f32, those of the imported reference (g11 pref_logits)."""
    from speechain_amd import functional as Fn
    from speechain_amd.utilbox.train_util import make_mask_from_len

    g = golden("g11_decode")
    dec, _ = build(g)
    Fn.set_compute_dtype(dtype)
    try:
        enc, enc_len = g["enc"].cuda(), g["enc_len"].cuda()
        mask = make_mask_from_len(enc_len, max_len=enc.size(1)).cuda()
        text, text_len = g["pref_text"].cuda(), g["pref_text_len"]
        with torch.no_grad():
            full = dec(enc, mask, text, text_len.cuda())[0].float()
        tol = 1e-4 if dtype is None else 0.03 * float(full.abs().max()) + 0.03  # bf16: two roundings of values of that size
        if dtype is None:
            assert float((full.cpu() - g["pref_logits"]).abs().max()) < tol
        beam = 2
        cache = dec.start_decoding(enc, mask, beam, 10)
        for pos in range(text.size(1)):
            step = dec.forward_step(text[:, pos].repeat_interleave(beam), cache).float()
            for b in range(text.size(0)):
                if pos < int(text_len[b]):  # (beyond a prefix's end forward() sees padding-masked positions; a step does not)
                    for s in range(beam):
                        err = float((step[b * beam + s] - full[b, pos]).abs().max())
                        assert err < tol, (pos, b, s, err)
    finally:
        Fn.set_compute_dtype(None)


def test_ctc_prefix_scorer_against_reference_trace(golden):
    from speechain_amd.infer_func.ctc_decoding import CTCPrefixScorer

    g = golden("g11_decode")
    sc = CTCPrefixScorer(x=g["ctc_x"].cuda(), enc_lens=g["ctc_lens"].cuda(), batch_size=2, beam_size=3, blank_index=0, eos_index=8)
    text, state = torch.zeros(6, 0, dtype=torch.long, device="cuda"), None
    for step in range(3):
        psi, state = sc.forward_step(text, state)
        want = g[f"ctc_psi{step}"]
        assert torch.allclose(psi.cpu(), want, atol=2e-4, rtol=1e-5), (step, float((psi.cpu() - want).abs().max()))
        beam_idx, tokens = g[f"ctc_beam_idx{step}"].cuda(), g[f"ctc_tokens{step}"].cuda()
        text = torch.cat([text[beam_idx], tokens[:, None]], dim=1)
        state = sc.permute_mem(state, beam_idx, tokens.view(2, 3) + (torch.arange(3, device="cuda") * 9)[None, :])
        assert torch.allclose(state[0].cpu(), g[f"ctc_r{step}"], atol=2e-4, rtol=1e-5), step
        assert torch.allclose(state[1][:, 0].cpu(), g[f"ctc_psimem{step}"], atol=2e-4, rtol=1e-5), step
    psi, _ = sc.forward_step(text, state)
    assert torch.allclose(psi.cpu(), g["ctc_psi3"], atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("cached", [True, False], ids=["cached", "prefix"])
def test_beam_search_against_reference(golden, cached):
    """beam_searching on the HIP path (f32) returns the reference's hypotheses, lengths and confidences for the four argument sets of
    g11 -- with the cached decoder and with the decoder wrapped as a plain function (the reference's calling convention)."""
    from speechain_amd.infer_func.beam_search import beam_searching
    from speechain_amd.utilbox.train_util import make_mask_from_len

    g = golden("g11_decode")
    dec, ctc = build(g)
    enc, enc_len = g["enc"].cuda(), g["enc_len"].cuda()
    mask = make_mask_from_len(enc_len, max_len=enc.size(1)).cuda()
    fn = dec if cached else (lambda enc_feat, enc_feat_mask, text, text_len: dec(enc_feat, enc_feat_mask, text, text_len))
    # (cached: the first pass over the argument sets launches every step eagerly, the second and third replay one hipGraph per position)
    for name, kw in list(CASES.items()) * (3 if cached else 1):
        res = beam_searching(enc_feat=enc, enc_feat_mask=mask, asr_decode_fn=fn, ctc_decode_fn=ctc, vocab_size=V, sos_eos=V - 1,
                             padding_idx=0, **kw)
        assert torch.equal(res["hypo_text"].cpu(), g[f"{name}.hypo_text"]), (name, res["hypo_text"], g[f"{name}.hypo_text"])
        assert torch.equal(res["hypo_text_len"].cpu(), g[f"{name}.hypo_text_len"]), name
        assert torch.allclose(res["hypo_text_confid"], g[f"{name}.hypo_text_confid"].float(), atol=2e-4), name
        assert torch.allclose(res["feat_token_len_ratio"].cpu(), g[f"{name}.feat_token_len_ratio"].float(), atol=1e-5), name


@pytest.mark.parametrize("G,parts,L,K", [(32, 16, 5000, 32), (3, 4, 92, 8), (2, 16, 64, 64), (5, 1, 300, 16), (2, 40, 50, 40)])
def test_two_stage_topk_equals_one_selection(G, parts, L, K):
    """Fn.topk_groups (every row's own top K, then sc_topk_merge per group of `parts` rows) returns exactly what one sc_topk_rows over the
    (G, parts * L) view returns -- values AND indices, heavy ties, -inf runs and -1e20 included (the beam search's selection must not
    depend on which form ran)."""
    from speechain_amd import functional as Fn

    gen = torch.Generator().manual_seed(G * 11 + L)
    x = torch.randn(G * parts, L, generator=gen)
    x[0, : L // 2] = float("-inf")
    x[parts - 1] = torch.randint(0, 4, (L,), generator=gen).float()  # heavy ties inside a row ...
    if G > 1:
        x[parts: 2 * parts] = torch.randint(0, 3, (parts, L), generator=gen).float()  # ... and across the rows of a group
    if G > 2:
        x[2 * parts, ::3] = -1e20
        x[2 * parts + 1:3 * parts] = float("-inf")  # (a group whose other rows are dead beams)
    x = x.cuda()
    v2, i2 = Fn.topk_groups(x, parts, K)
    v1, i1 = Fn.topk_rows(x.view(G, parts * L), K)
    assert torch.equal(v2, v1)
    assert torch.equal(i2, i1)


@pytest.mark.parametrize("R,L,K", [(32, 80000, 32), (3, 92, 8), (5, 5000, 64), (2, 40, 40), (4, 1025, 2), (6, 9000, 16), (3, 12288, 32), (4, 5000, 1)])
def test_topk_rows_against_torch(R, L, K):
    """sc_topk_rows: values equal torch.topk's (sorted, descending); indices point at those values; among equal values the lower index
    comes first and wins the last place (rows with many ties, -inf runs and the reference's -1e20 included)."""
    from speechain_amd import functional as Fn

    gen = torch.Generator().manual_seed(R * 7 + L)
    x = torch.randn(R, L, generator=gen)
    x[0, : L // 2] = float("-inf")
    if R > 1:
        x[1] = torch.randint(0, 6, (L,), generator=gen).float()  # heavy ties
    if R > 2:
        x[2, ::3] = -1e20
    x = x.cuda()
    vals, idx = Fn.topk_rows(x, K)
    want, _ = torch.topk(x, K, dim=1, largest=True, sorted=True)
    assert torch.equal(vals, want)
    assert torch.equal(torch.gather(x, 1, idx), vals)
    for r in range(R):
        v, i = vals[r].cpu(), idx[r].cpu()
        assert len(set(i.tolist())) == K
        for a in range(K - 1):
            assert v[a] > v[a + 1] or (v[a] == v[a + 1] and i[a] < i[a + 1]), (r, a)
        # the last place among equals goes to the lowest indices
        last = v[-1]
        eq_all = (x[r].cpu() == last).nonzero().flatten()
        taken = i[v == last]
        assert torch.equal(taken.sort().values, eq_all[: len(taken)])


def test_asr_encoder_sequencing_and_inference():
    """ASREncoder built from recipe-style dictionaries (frontend -> normalisation -> prenet -> encoder; SpecAugment only in training)
    equals its parts applied by hand, and feeds beam_searching through the ARASRDecoder surface end to end (waveforms in, token ids out)."""
    from speechain_amd import functional as Fn
    from speechain_amd.infer_func.beam_search import beam_searching
    from speechain_amd.module.decoder.ar_asr import ARASRDecoder
    from speechain_amd.module.encoder.asr import ASREncoder
    from speechain_amd.module.postnet.token import TokenPostnet

    torch.manual_seed(0)
    conf = dict(frontend=dict(type="frontend.speech2mel.Speech2MelSpec", conf=dict(n_mels=80, hop_length=0.010, win_length=0.025, sr=16000, preemphasis=0.97)),
                normalize=dict(norm_type="utterance"), specaug=dict(time_warp=False),
                prenet=dict(type="prenet.conv2d.Conv2dPrenet", conf=dict(conv_dims=[8, 8], lnr_dims=128)),
                encoder=dict(type="transformer.encoder.TransformerEncoder",
                             conf=dict(num_heads=2, num_layers=2, fdfwd_dim=128, posenc_dropout=0.0, att_dropout=0.0, fdfwd_dropout=0.0, res_dropout=0.0)))
    enc = ASREncoder(**conf).cuda().eval()
    wave = (torch.randn(2, 8000, 1) * 0.1).cuda()
    wave_len = torch.tensor([8000, 6400]).cuda()
    with torch.no_grad():
        out, mask, _, _ = enc(wave, wave_len.clone())
        f, fl = enc.frontend(wave, wave_len.clone())
        f, fl = enc.normalize(f, fl)
        f, fl = enc.prenet(f, fl)
        from speechain_amd.utilbox.train_util import make_mask_from_len
        want, _, _, _ = enc.encoder(f, make_mask_from_len(fl, max_len=f.size(1)).cuda())
    assert torch.equal(out, want) and out.shape[0] == 2 and mask.shape == (2, 1, out.size(1))
    assert hasattr(enc, "specaug") and enc.output_size == 128
    V = 19
    dec = ARASRDecoder(vocab_size=V, embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=128)),
                       decoder=dict(type="transformer.decoder.TransformerDecoder", conf=dict(num_heads=2, num_layers=1, fdfwd_dim=64))).cuda().eval()
    ctc = TokenPostnet(input_size=128, vocab_size=V).cuda().eval()
    res = beam_searching(enc_feat=out, enc_feat_mask=mask, asr_decode_fn=dec, ctc_decode_fn=ctc, vocab_size=V, beam_size=3, ctc_weight=0.3,
                         min_f2t_ratio=2.0)
    assert res["hypo_text"].shape[0] == 2 and int(res["hypo_text_len"].max()) <= out.size(1) // 2
    assert bool(torch.isfinite(res["hypo_text_confid"]).all())


def test_ctc_prefix_scorer_medium_size_against_oracle():
    """CTCPrefixScorer on the HIP kernels vs the oracle's restatement of the reference at sizes that exercise what the small fixture cannot:
    more frames than lanes (several frames per lane in the scans), four rows of a sentence per workgroup, ragged lengths, six steps with
    random parents and tokens (repeated last labels included)."""
    from oracle import decode_ref as D
    from speechain_amd.infer_func.ctc_decoding import CTCPrefixScorer

    gen = torch.Generator().manual_seed(11)
    B, beam, T, V = 3, 4, 150, 300
    x = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 2.0, dim=-1).repeat_interleave(beam, dim=0)
    lens = torch.tensor([150, 97, 64]).repeat_interleave(beam)
    ref = D.CtcPrefixScores(x, lens, B, beam, blank_index=0, eos_index=V - 1)
    sc = CTCPrefixScorer(x=x.cuda(), enc_lens=lens.cuda(), batch_size=B, beam_size=beam, blank_index=0, eos_index=V - 1)
    text = torch.zeros(B * beam, 0, dtype=torch.long)
    s_ref, s_gpu = None, None
    for step in range(6):
        p_ref, s_ref = ref.forward_step(text, s_ref)
        p_gpu, s_gpu = sc.forward_step(text.cuda(), s_gpu)
        fin = p_ref > -1e19
        assert torch.equal(fin, p_gpu.cpu() > -1e19), step
        err = float((p_gpu.cpu()[fin] - p_ref[fin]).abs().max())
        assert err < 2e-3, (step, err)  # (sums of up to 150 terms of magnitude up to ~1e3 in f32, two different summation orders)
        parents = (torch.arange(B)[:, None] * beam + torch.randint(0, beam, (B, beam), generator=gen)).reshape(-1)
        tokens = torch.randint(1, V - 1, (B * beam,), generator=gen)
        if step >= 1:
            tokens[::3] = text[parents[::3], -1]  # repeat the parent's last label: the phi = r^b branch
        text = torch.cat([text[parents], tokens[:, None]], dim=1)
        tok_idx = tokens.view(B, beam) + (torch.arange(beam) * V)[None, :]
        s_ref = ref.permute_mem(s_ref, parents, tok_idx)
        s_gpu = sc.permute_mem(s_gpu, parents.cuda(), tok_idx.cuda())
        fin_r = s_ref[0] > -1e19
        assert torch.equal(fin_r, s_gpu[0].cpu() > -1e19), step
        assert float((s_gpu[0].cpu()[fin_r] - s_ref[0][fin_r]).abs().max()) < 2e-3, step


def test_cached_and_prefix_beam_search_agree_at_recipe_width():
    """d 512, 8 heads, V 5000, beam 16, 100 frames, CTC weight 0.2 (two decoder layers to keep it short): the cached path (eager, then replayed)
    and the stateless calling convention on the same HIP kernels pick the same hypotheses, f32 mode."""
    from speechain_amd.infer_func.beam_search import beam_searching
    from speechain_amd.module.decoder.ar_asr import ARASRDecoder
    from speechain_amd.module.postnet.token import TokenPostnet
    from speechain_amd.utilbox.train_util import make_mask_from_len

    torch.manual_seed(4)
    V = 5000
    dec = ARASRDecoder(vocab_size=V, embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=512)),
                       decoder=dict(type="transformer.decoder.TransformerDecoder",
                                    conf=dict(num_heads=8, num_layers=2, fdfwd_dim=2048, fdfwd_activation="GELU", emb_scale=False))).cuda().eval()
    with torch.no_grad():
        dec.postnet.linear.weight.mul_(3.0)  # sharper token distributions: well-separated candidates
    ctc = TokenPostnet(input_size=512, vocab_size=V).cuda().eval()
    enc = torch.randn(2, 100, 512).cuda()
    mask = make_mask_from_len(torch.tensor([100, 77]), max_len=100).cuda()
    kw = dict(vocab_size=V, beam_size=16, ctc_weight=0.2, min_f2t_ratio=4.0)
    want = beam_searching(enc_feat=enc, enc_feat_mask=mask, asr_decode_fn=lambda enc_feat, enc_feat_mask, text, text_len:
                          dec(enc_feat, enc_feat_mask, text, text_len), ctc_decode_fn=ctc, **kw)
    for _ in range(2):  # eager, then replayed
        got = beam_searching(enc_feat=enc, enc_feat_mask=mask, asr_decode_fn=dec, ctc_decode_fn=ctc, **kw)
        assert torch.equal(got["hypo_text"], want["hypo_text"]) and torch.equal(got["hypo_text_len"], want["hypo_text_len"])
        assert torch.allclose(got["hypo_text_confid"], want["hypo_text_confid"], atol=1e-3)


@pytest.mark.parametrize("dtype,ctc,eos_stats", [(torch.float32, True, True), (torch.bfloat16, True, False), (torch.float32, False, False)])
def test_fused_step_tail_equals_the_separate_launches(dtype, ctc, eos_stats):
    """sc_beam_candidates_topk + sc_beam_merge_advance (candidate row in LDS, merge + record + next rows in one launch) against
    sc_beam_candidates + sc_topk_rows + sc_topk_merge + sc_beam_advance on the same inputs: the same record and the same next rows."""
    from speechain_amd import functional as Fn

    B, beam, V, K = 5, 8, 3000, 16
    N = B * beam
    g = torch.Generator().manual_seed(11)
    logits = (torch.randn(N, V, generator=g) * 3).to(dtype).cuda()
    scores = torch.randn(N, generator=g).cuda()
    psi = (torch.randn(N, V, generator=g) - 5).cuda() if ctc else None
    psi_prev = torch.randn(N, generator=g).cuda() if ctc else None
    eos = V - 1
    logits[::3, eos] += 12.0  # <eos> among the best candidates of a third of the rows: the skipping rule is exercised
    width = 2 * K + (2 * beam if eos_stats else 0)

    def state():
        return (torch.zeros(B, width, dtype=torch.float64, device="cuda"), scores.clone(), torch.arange(N, device="cuda"),
                torch.zeros(N, dtype=torch.long, device="cuda"), torch.zeros(N, dtype=torch.long, device="cuda"))

    pack_a, sc_a, tok_a, prev_a, par_a = state()
    cand, es, om = Fn.beam_candidates(logits, sc_a, 1.3, psi, psi_prev, 0.2 if ctc else 0.0, 0, eos, want_eos_stats=eos_stats)
    v, i = Fn.topk_groups(cand, beam, K)
    Fn.beam_advance(v, i, es if eos_stats else None, om if eos_stats else None, beam, V, eos, pack_a, sc_a, tok_a, prev_a, par_a)
    pack_b, sc_b, tok_b, prev_b, par_b = state()
    v2, i2 = Fn.beam_step_select(logits, 1.3, psi, psi_prev, 0.2 if ctc else 0.0, 0, eos, eos_stats, beam, K, pack_b, sc_b, tok_b, prev_b, par_b)
    torch.cuda.synchronize()
    assert torch.equal(i2, i) and torch.allclose(v2, v, atol=2e-5, rtol=1e-6)
    assert torch.equal(tok_b, tok_a) and torch.equal(par_b, par_a) and torch.equal(prev_b, prev_a) and torch.allclose(sc_b, sc_a, atol=2e-5, rtol=1e-6)
    assert torch.allclose(pack_b, pack_a, atol=2e-5, rtol=1e-6)
    assert int((tok_b == eos).sum()) == 0  # no row continues with <sos/eos>


@pytest.mark.parametrize("B,Tq,H,Tk", [(32, 16, 8, 256), (3, 4, 2, 77), (2, 32, 8, 1030), (5, 1, 4, 65), (2, 16, 2, 700)])
def test_few_queries_against_many_keys_split_over_the_waves(B, Tq, H, Tk):
    """The encoder-decoder attention of a decoding position (<= 32 queries per sentence and head, beams as queries): the four waves of
    `attn_fwd_stream_kernel<false, true>` take every fourth key tile each and merge their partial softmaxes.  Against fp32 softmax attention
    on the same bf16 operands, ragged encoder lengths (tiles without a live key included), and against the unsplit kernel (SC_ATTN_NO_SPLIT
    is read once per process, so the comparison is with fp32 only here)."""
    from speechain_amd import functional as Fn

    g = torch.Generator().manual_seed(B * 100 + Tk)
    d = H * 64
    q = torch.randn(B, Tq, d, generator=g).bfloat16().cuda()
    kv = torch.randn(B, Tk, 2 * d, generator=g).bfloat16().cuda()
    lens = torch.randint(max(1, Tk // 5), Tk + 1, (B,), generator=g)
    lens[0] = Tk
    mask = (torch.arange(Tk)[None, :] < lens[:, None])[:, None, :].cuda()
    scale = 0.125
    Fn.set_compute_dtype(torch.bfloat16)
    Fn.set_attention_probs(False)
    try:
        with torch.no_grad():
            out, _ = Fn.attention_core(q, kv, 0, 0, 1, H, d, mask, False, scale, 0.0)
        qf = q.float().view(B, Tq, H, 64).permute(0, 2, 1, 3)
        kf = kv.float()[..., :d].view(B, Tk, H, 64).permute(0, 2, 1, 3)
        vf = kv.float()[..., d:].view(B, Tk, H, 64).permute(0, 2, 1, 3)
        sc = ((qf @ kf.transpose(-1, -2)) * scale).masked_fill(~mask[:, :, None, :], float("-inf"))
        ref = (torch.softmax(sc, dim=-1) @ vf).permute(0, 2, 1, 3).reshape(B, Tq, d)
        e = float((out.float() - ref).norm() / ref.norm())
        assert e < 8e-3, e  # (bf16 probabilities and output, as the training kernel: 4e-3 measured)
    finally:
        Fn.set_compute_dtype(None)
        Fn.set_attention_probs(True)


def test_replayed_decoding_follows_parameter_updates():
    """The per-position hipGraphs of a decoding session read the cached bf16 weight copies by address and never run the layers' host-side
    lookups again: after an in-place parameter update (a training step between two validation decodes) the replayed search must use the new
    weights -- it has to equal the stateless calling convention (no cache, no graphs) on the updated model, and differ from the old result."""
    from speechain_amd import functional as Fn
    from speechain_amd.infer_func.beam_search import beam_searching
    from speechain_amd.module.decoder.ar_asr import ARASRDecoder
    from speechain_amd.utilbox.train_util import make_mask_from_len

    torch.manual_seed(7)
    V = 300
    dec = ARASRDecoder(vocab_size=V, embedding=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=512)),
                       decoder=dict(type="transformer.decoder.TransformerDecoder",
                                    conf=dict(num_heads=8, num_layers=2, fdfwd_dim=2048, fdfwd_activation="GELU", emb_scale=False))).cuda().eval()
    with torch.no_grad():
        dec.postnet.linear.weight.mul_(3.0)
    enc = torch.randn(2, 64, 512).cuda()
    mask = make_mask_from_len(torch.tensor([64, 50]), max_len=64).cuda()
    kw = dict(vocab_size=V, beam_size=4, min_f2t_ratio=4.0)
    stateless = lambda enc_feat, enc_feat_mask, text, text_len: dec(enc_feat, enc_feat_mask, text, text_len)
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        for _ in range(3):  # eager, captured + replayed, replayed
            before = beam_searching(enc_feat=enc, enc_feat_mask=mask, asr_decode_fn=dec, **kw)
        with torch.no_grad():  # an "optimizer step": every matrix of the decoder stack and the LayerNorm in front of the products
            for n, p in dec.named_parameters():
                if p.dim() == 2 and "embed" not in n:
                    p.add_(0.05 * torch.randn_like(p) * p.std())
                elif n.endswith("layernorm.weight") or "norm" in n and n.endswith("weight"):
                    p.mul_(1.1)
        after = beam_searching(enc_feat=enc, enc_feat_mask=mask, asr_decode_fn=dec, **kw)  # replayed
        want = beam_searching(enc_feat=enc, enc_feat_mask=mask, asr_decode_fn=stateless, **kw)
        assert not torch.equal(after["hypo_text_confid"], before["hypo_text_confid"])  # (the update is large enough to move the scores)
        assert torch.allclose(after["hypo_text_confid"], want["hypo_text_confid"], rtol=2e-2, atol=2e-2), (after["hypo_text_confid"], want["hypo_text_confid"],
                                                                                                            before["hypo_text_confid"])
        assert not torch.allclose(before["hypo_text_confid"], want["hypo_text_confid"], rtol=2e-2, atol=2e-2)
    finally:
        Fn.set_compute_dtype(None)
        Fn.weights.invalidate()


@pytest.mark.parametrize("beam,plen", [(16, 0), (16, 1), (32, 4), (16, 9), (8, 2), (5, 3), (20, 1)])
def test_prefix_scores_from_the_state_launch_s_prepared_factors(beam, plen):
    """sc_ctc_prefix_state with pt / pa (exp(phi - a) and a of the new rows, in the matrix-core kernel's layout) + sc_ctc_prefix_scores_fast
    reading them: bit-identical scores to the launch that derives them from r itself (ragged lengths, an impossible row, a row whose
    forward variables are all impossible)."""
    from speechain_amd import functional as Fn

    B, T, V, blank, eos = 3, 61, 300, 0, 299
    gen = torch.Generator().manual_seed(90 + plen)
    x = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 5.0, dim=-1).cuda()
    enc_lens = torch.tensor([T, 47, 30], dtype=torch.int32).cuda()
    N = B * beam
    eos_frame = (enc_lens.repeat_interleave(beam) - 1).to(torch.int32)
    r = torch.randn(T, 2, N, generator=gen) * 20.0 - 30.0
    r[: max(plen - 1, 0)] = -1e20
    r[:, :, 5] = -1e20
    r = r.cuda()
    last = torch.randint(1, V - 1, (N,), generator=gen).cuda()
    beam_idx = (torch.arange(N) // beam * beam + torch.randint(0, beam, (N,), generator=gen)).cuda()
    tokens = torch.randint(1, V - 1, (N,), generator=gen).cuda()
    prep = Fn.ctc_prefix_prepared(B, beam, T, x.device)
    r_new = Fn.ctc_prefix_state(x, enc_lens, r, last if plen > 0 else None, plen, beam_idx, tokens, beam, blank, prepared=prep)
    r_ref = Fn.ctc_prefix_state(x, enc_lens, r, last if plen > 0 else None, plen, beam_idx, tokens, beam, blank)
    assert torch.equal(r_new, r_ref)
    fac = Fn.ctc_prefix_factors(x)
    want = Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r_new, tokens, plen + 1, beam, blank, eos, factors=fac)
    got = Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r_new, tokens, plen + 1, beam, blank, eos, factors=fac, prepared=prep)
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("beam,plen", [(8, 0), (16, 3), (6, 5)])
def test_factorised_scores_vector_kernel(beam, plen):
    """V % 4 != 0 keeps sc_ctc_prefix_scores_fast on ctc_psi_fast_kernel (one candidate per lane): same bound against the exact kernel."""
    from speechain_amd import functional as Fn

    B, T, V, blank, eos = 2, 47, 203, 0, 202
    gen = torch.Generator().manual_seed(7 + plen)
    x = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 6.0, dim=-1).cuda()
    enc_lens = torch.tensor([T, 31], dtype=torch.int32).cuda()
    N = B * beam
    r = torch.randn(T, 2, N, generator=gen) * 30.0 - 40.0
    r[: max(plen - 1, 0)] = -1e20
    r[:, :, 1] = -1e20
    r = r.cuda()
    eos_frame = (enc_lens.repeat_interleave(beam) - 1).to(torch.int32)
    last = torch.randint(1, V - 1, (N,), generator=gen).cuda()
    want = Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r, last if plen > 0 else None, plen, beam, blank, eos)
    got = Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r, last if plen > 0 else None, plen, beam, blank, eos, factors=Fn.ctc_prefix_factors(x))
    assert torch.equal(got[:, blank], want[:, blank]) and torch.equal(got[:, eos], want[:, eos])
    assert torch.allclose(got, want, atol=2e-4, rtol=2e-6), float((got - want).abs().max())


@pytest.mark.parametrize("beam", [3, 8, 16, 20, 32])
@pytest.mark.parametrize("plen", [0, 1, 3, 9])
def test_factorised_ctc_prefix_scores_equal_the_exact_kernel(plen, beam):
    """sc_ctc_prefix_scores_fast (exp(phi + x) = exp(phi - a) exp(x - bmax), the second factor made once per batch) against the exact
    two-sweep kernel on peaky log-probabilities with ragged lengths, impossible rows and forward variables far apart: every score within
    2e-4 absolute + 2e-6 relative, <eos> / blank columns identical.  The factorised form is the matrix-core kernel (ctc_psi_mfma_kernel:
    the same sums as f32 MFMA products, up to 16 rows of a sentence per tile -- partly filled tiles for beams 3, 8, 20); the vector
    kernel (V % 4 != 0) is covered by `test_factorised_scores_vector_kernel`."""
    from speechain_amd import functional as Fn

    B, T, V, blank, eos = 3, 61, 300, 0, 299
    gen = torch.Generator().manual_seed(40 + plen)
    logits = torch.randn(B, T, V, generator=gen) * 6.0  # peaky: log-probabilities from about -0.01 down to -50
    x = torch.log_softmax(logits, dim=-1).cuda()
    enc_lens = torch.tensor([T, 47, 30], dtype=torch.int32).cuda()
    N = B * beam
    r = torch.randn(T, 2, N, generator=gen) * 30.0 - 40.0  # forward variables spread over ~100 nats
    r[: max(plen - 1, 0)] = -1e20  # frames before the prefix could have been emitted
    r[:, :, 5] = -1e20  # an impossible row
    r = r.cuda()
    eos_frame = (enc_lens.repeat_interleave(beam) - 1).to(torch.int32)
    last = torch.randint(1, V - 1, (N,), generator=gen).cuda()
    want = Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r, last if plen > 0 else None, plen, beam, blank, eos)
    got = Fn.ctc_prefix_scores(x, enc_lens, eos_frame, r, last if plen > 0 else None, plen, beam, blank, eos, factors=Fn.ctc_prefix_factors(x))
    assert torch.equal(got[:, blank], want[:, blank]) and torch.equal(got[:, eos], want[:, eos])
    assert torch.allclose(got, want, atol=2e-4, rtol=2e-6), float((got - want).abs().max())

"""TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of the SpeeChain ASR *inference* callers.

    beam_search()        speechain/infer_func/beam_search.py:106-550   (batched beam search, n-best containers :21-103)
    CtcPrefixScores      speechain/infer_func/ctc_decoding.py:6-196     (CTC prefix scores, Algorithm 2 of Watanabe et al. 2017)
    asr_decode_logits()  speechain/module/decoder/ar_asr.py:49-78       (embedding -> TransformerDecoder -> TokenPostnet on a prefix)

Like the reference, ``beam_search`` re-runs the decoder on the whole growing prefix at every step (no key/value cache): that is
the behaviour the cached HIP path is checked against.  Pinned against the imported reference by ``tests/golden/make_golden.py``
(fixture ``g11_decode.npz``).  Pure functions: explicit weights in, tensors out.
"""
from typing import Callable, Dict, List, Optional

import torch

from . import cpu_ref

Tensor = torch.Tensor
EPS = 1e-20        # beam_search.py:17
MINUS_INF = -1e20  # beam_search.py:18, ctc_decoding.py:46 (a finite stand-in: sums of it stay finite)


def asr_decode_logits(enc_feat: Tensor, enc_feat_mask: Tensor, text: Tensor, text_len: Tensor, P: Dict[str, Tensor], cfg: Dict) -> Tensor:
    """(N, L, V) logits of every prefix position (decoder/ar_asr.py:49-78).  P keys: ``embed.weight``, ``decoder.*``, ``postnet.*``."""
    emb = torch.nn.functional.embedding(text, P["embed.weight"], padding_idx=0)
    mask = cpu_ref.make_mask_from_len(text_len, max_len=text.size(1))
    dec, _, _, _ = cpu_ref.transformer_decoder(emb, enc_feat, mask, enc_feat_mask, P, "decoder.", num_layers=cfg["dec_layers"],
                                               num_heads=cfg["num_heads"], d_model=cfg["d_model"], emb_scale=cfg.get("emb_scale", False),
                                               activation=cfg.get("activation", "ReLU"))
    return cpu_ref.token_postnet(dec, P, "postnet.")


class CtcPrefixScores:
    """ctc_decoding.py:6-196.  ``x``: (N, T, V) CTC log-probabilities of the N = batch * beam hypothesis rows, ``enc_lens`` (N,).
    State between steps: (r (T, 2, N) log forward variables of the kept prefixes [0: ends in a label, 1: ends in blank], psi (N, V))."""

    def __init__(self, x: Tensor, enc_lens: Tensor, batch_size: int, beam_size: int, blank_index: int, eos_index: int):
        self.blank, self.eos, self.batch, self.beam = blank_index, eos_index, batch_size, beam_size
        N, self.T, self.V = x.shape
        x = x.clone()
        pad = ~cpu_ref.make_mask_from_len(enc_lens, max_len=self.T, return_3d=False)  # frames past the utterance (:49-57)
        x[pad] = MINUS_INF
        x[:, :, blank_index] = torch.where(pad, torch.zeros(()), x[:, :, blank_index])  # ... where blank is certain
        self.x_label = x.transpose(0, 1).contiguous()                      # (T, N, V)
        self.x_blank = self.x_label[:, :, blank_index]                     # (T, N)
        self.last = enc_lens - 1

    def forward_step(self, g: Tensor, state):
        """g: (N, |g|) prefixes without the leading <sos>.  Returns (psi - psi_prev (N, V), (r (T, 2, N, V), psi))."""
        T, N, V = self.T, self.x_label.size(1), self.V
        plen = g.size(1)
        if state is None:  # (:88-100) empty prefix: only blanks so far
            r_prev = torch.full((T, 2, N), MINUS_INF)
            r_prev[:, 1] = torch.cumsum(self.x_blank, dim=0)
            psi_prev = 0.0
        else:
            r_prev, psi_prev = state
        r = torch.full((T, 2, N, V), MINUS_INF)
        if plen == 0:
            r[0, 0] = self.x_label[0]                                      # (:116-117)
        r_sum = torch.logsumexp(r_prev, dim=1)                             # (T, N)  (:119)
        phi = r_sum[:, :, None].repeat(1, 1, V)
        if plen > 0:                                                       # (:123-124) repeating the last label needs a blank in between
            last = g[:, -1]
        else:
            last = torch.zeros(N, dtype=torch.long)
        phi[:, torch.arange(N), last] = r_prev[:, 1]
        start = max(1, plen)
        for t in range(start, T):                                          # (:132-142)
            nb = torch.logsumexp(torch.stack([r[t - 1, 0], phi[t - 1]]), dim=0) + self.x_label[t]
            b = torch.logsumexp(torch.stack([r[t - 1, 0], r[t - 1, 1]]), dim=0) + self.x_blank[t][:, None]
            r[t, 0], r[t, 1] = nb, b
        # prefix probability (:144-149): sum over the frame at which the new label first appears
        phix = torch.cat([phi[:1], phi[:-1]], dim=0) + self.x_label
        psi = torch.logsumexp(torch.cat([phix[start:T], r[start - 1, 0][None]], dim=0), dim=0)
        rows = torch.arange(N)
        psi[rows, self.eos] = r_sum[self.last[rows // self.beam], rows]    # (:151-155)
        psi[:, self.blank] = MINUS_INF                                     # (:158)
        return psi - psi_prev, (r, psi)

    def permute_mem(self, memory, beam_idx: Tensor, token_idx: Tensor):
        """Keeps, for every new row, the state of (parent row, chosen token) (:162-196).  token_idx: (batch, beam) = token + V * slot."""
        r, psi = memory
        N, V = psi.shape
        r, psi = r[:, :, beam_idx], psi[beam_idx]
        # the reference indexes the flattened (N * V) axis with  token + V * slot + V * beam * sentence  (:180-184)
        flat = (token_idx + (torch.arange(self.batch) * self.beam)[:, None] * V).reshape(-1)
        psi = psi.reshape(-1)[flat][:, None].repeat(1, V)
        r = r.reshape(r.size(0), 2, N * V)[:, :, flat]
        return r, psi


class _NBest:
    """beam_search.py:21-103: the n-best list of one sentence (score = sum of log-probs / (length + eps) ** length_penalty)."""

    def __init__(self, beam: int, max_len: int, length_penalty: float):
        self.beam, self.max_len, self.lp = beam, max_len - 1, length_penalty
        self.items: List = []
        self.worst = 1e9

    def add(self, hyp: Tensor, sum_logprobs: float):
        score = sum_logprobs / ((len(hyp) + EPS) ** self.lp)
        if len(self.items) < self.beam or score > self.worst:
            self.items.append((score, hyp))
            if len(self.items) > self.beam:
                order = sorted((s, i) for i, (s, _) in enumerate(self.items))
                del self.items[order[0][1]]
                self.worst = order[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs: float, curr_len: int) -> bool:
        if len(self.items) < self.beam:
            return False
        return best_sum_logprobs / ((curr_len + EPS) ** self.lp) < self.worst


def beam_search(enc_feat: Tensor, enc_feat_mask: Tensor, decode_fn: Callable, vocab_size: int, sos_eos: Optional[int] = None,
                padding_idx: int = 0, beam_size: int = 1, min_f2t_ratio: float = 3.0, length_penalty: float = 1.0,
                temperature: float = 1.0, eos_filtering: bool = False, eos_threshold: float = 1.5, ctc_weight: float = 0.0,
                ctc_logits_fn: Optional[Callable] = None, ctc_temperature: float = 1.0) -> Dict[str, Tensor]:
    """beam_search.py:106-550 without the language-model branch.  ``decode_fn(enc_feat, enc_feat_mask, text, text_len) -> logits
    (N, L, V)``; ``ctc_logits_fn(enc_feat) -> (N, T, V)``."""
    B, T = enc_feat.size(0), enc_feat.size(1)
    enc_len = enc_feat_mask.squeeze(1).sum(-1)
    max_len = int(T / min_f2t_ratio) if min_f2t_ratio > 0 else int(-min_f2t_ratio)      # (:250-252)
    if sos_eos is None:
        sos_eos = vocab_size - 1
    N = B * beam_size
    enc_feat = enc_feat.repeat_interleave(beam_size, dim=0)                             # (:258-274) rows of one sentence are adjacent
    enc_feat_mask = enc_feat_mask.repeat_interleave(beam_size, dim=0)
    scorer, memory = None, None
    if ctc_logits_fn is not None and ctc_weight > 0:                                    # (:276-292)
        logits = ctc_logits_fn(enc_feat).clone()
        logits[:, :, sos_eos] = MINUS_INF
        scorer = CtcPrefixScores(torch.log_softmax(logits / ctc_temperature, dim=-1), enc_feat_mask.squeeze(1).sum(-1), B, beam_size,
                                 padding_idx, sos_eos)
    nbest = [_NBest(beam_size, max_len, length_penalty) for _ in range(B)]
    done = [False] * B
    beam_scores = torch.full((N,), float("-inf"))
    beam_scores[::beam_size] = 0.0                                                      # (:303-311) one live beam per sentence at first
    text = torch.full((N, 1), sos_eos, dtype=torch.long)
    text_len = torch.ones(N, dtype=torch.long)
    while int(text_len.max()) < max_len:
        logits = decode_fn(enc_feat, enc_feat_mask, text, text_len)[:, -1, :]
        tok = torch.log_softmax(logits / temperature, dim=-1)                           # (:334)
        if scorer is not None:                                                          # (:337-345)
            tok[:, padding_idx] = MINUS_INF
            ctc_tok, memory = scorer.forward_step(text[:, 1:], memory)
            tok = (1 - ctc_weight) * tok + ctc_weight * ctc_tok
        cand = (tok + beam_scores[:, None]).view(B, -1)
        top_s, top_i = torch.topk(cand, 2 * beam_size, dim=1, largest=True, sorted=True)  # (:419-421)
        nxt = []
        for b in range(B):
            if done[b]:
                nxt.extend([(0, padding_idx, 0)] * beam_size)                           # (:428-431)
                continue
            kept = []
            for rank in range(2 * beam_size):
                flat, score = int(top_i[b, rank]), top_s[b, rank]
                slot, token = flat // vocab_size, flat % vocab_size
                row = b * beam_size + slot
                if token == sos_eos:                                                    # (:452-473)
                    if rank >= beam_size:
                        continue
                    if eos_filtering:
                        others = torch.cat([tok[row, :sos_eos], tok[row, sos_eos + 1:]]).max()
                        if tok[row, sos_eos] <= eos_threshold * others:
                            continue
                    nbest[b].add(text[row, 1:].clone(), float(score))
                else:
                    kept.append((score, token, row))
                if len(kept) == beam_size:
                    break
            done[b] = nbest[b].is_done(float(top_s[b].max()), int(text_len[b * beam_size:(b + 1) * beam_size].max()) - 1)  # (:495-506)
            nxt.extend(kept)
        if all(done):
            break
        beam_scores = torch.tensor([float(x[0]) for x in nxt])
        tokens = torch.tensor([int(x[1]) for x in nxt], dtype=torch.long)
        beam_idx = torch.tensor([int(x[2]) for x in nxt], dtype=torch.long)
        text = torch.cat([text[beam_idx], tokens[:, None]], dim=1)                      # (:518-519)
        text_len = (text != padding_idx).sum(-1)
        enc_feat, enc_feat_mask = enc_feat[beam_idx], enc_feat_mask[beam_idx]
        if memory is not None:                                                          # (:523-527)
            memory = scorer.permute_mem(memory, beam_idx, tokens.view(B, beam_size) + (torch.arange(beam_size) * vocab_size)[None, :])
    for b in range(B):                                                                  # (:529-541) ran into the length limit
        if not done[b]:
            for slot in range(beam_size):
                row = b * beam_size + slot
                nbest[b].add(text[row, 1:].clone(), float(beam_scores[row]))
    best = [sorted(n.items, key=lambda x: x[0])[-1] for n in nbest]                     # (:543-560)
    lens = torch.tensor([len(h) for _, h in best], dtype=torch.long)
    if int(lens.min()) != int(lens.max()):
        out = torch.full((B, min(int(lens.max()), max_len)), padding_idx, dtype=torch.long)
        for i, (_, h) in enumerate(best):
            out[i, :lens[i]] = h
    else:
        out = torch.stack([h for _, h in best])
    return dict(hypo_text=out, hypo_text_len=lens, feat_token_len_ratio=enc_len / (lens + 1e-10),
                hypo_text_confid=torch.tensor([s for s, _ in best]))

"""``beam_searching``, drop-in for ``speechain.infer_func.beam_search.beam_searching`` (speechain/infer_func/beam_search.py:106-550):
same arguments, same result dictionary (``hypo_text``, ``hypo_text_len``, ``feat_token_len_ratio``, ``hypo_text_confid``), same
selection rules -- 2 x beam candidates per sentence, <sos/eos> candidates of the first ``beam`` ranks close a hypothesis, the n-best list
with its length-penalised scores and early stop (:21-103), padding of finished sentences, the length limit ``feat_maxlen / min_f2t_ratio``.

What differs underneath:
  * a decode function that offers ``start_decoding`` / ``forward_step`` (``speechain_amd.module.decoder.ar_asr.ARASRDecoder``) is fed
    ONE token per hypothesis and step against cached keys / values (``csrc/decode.hip``); the reference re-runs the decoder on the whole
    prefix (:325-333).  Any other callable is used exactly as the reference uses it;
  * the encoder output is not repeated per beam (:258-274): the beams of a sentence are adjacent rows and never leave it;
  * the CTC prefix scores come from ``sc_ctc_prefix_scores`` (no (T, 2, N, V) tensor, see ``infer_func/ctc_decoding.py``);
  * the step's bookkeeping reads ONE small device-to-host copy (the 2 x beam best candidates per sentence) where the reference reads a
    scalar per candidate (``.item()``, :446-506); hypotheses live on the host as integer arrays;
  * with a cached decoder the rows of the next position (the first ``beam`` candidates of a sentence that are not <sos/eos>, :423-509) are
    ALSO written on the device, in the launch that packs the record (``sc_beam_advance``): the device runs position p + 1 while the host
    works through the record of position p -- n-best lists, <eos> filter, stopping rules, all as before and from the same numbers.  A sentence
    the host has declared finished keeps decoding on the device until the batch ends; the host ignores its candidates, as the reference
    ignores a finished sentence;
  * with a cached decoder the device side of a step -- ancestry table, 6 decoder layers, token head, log-softmax, CTC prefix scores,
    score mixing, top-k -- touches only buffers that live as long as the decode function and, from the second call on, is replayed from
    one hipGraph per position (``SC_DECODE_NO_GRAPH=1``: eager launches of the same code).
"""
import contextlib
import gc
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from speechain_amd import functional as Fn
from speechain_amd.infer_func.ctc_decoding import CTCPrefixScorer

# (A/B knob.  Round 2: the CTC side of a position on a second stream, 88.0 -> 85.3 ms per batch.  Round 3: with the position's launches down from ~95
#  to ~55 the fork and join of the branch inside the replayed graph (6-20 us each) cost more than the overlap gives: 59.2 -> 57.6 ms on ONE stream.)
_CTC_BRANCH = os.environ.get("SC_DECODE_CTC_BRANCH", "0") == "1"
_FUSED_TAIL = os.environ.get("SC_DECODE_FUSED_TAIL", "1") == "1"  # (A/B knob: 0 = candidate matrix in HBM, separate top-k launches)
_CTC_PSI_FAST = os.environ.get("SC_CTC_PSI_FAST", "1") == "1"  # (A/B knob: 0 = one v_exp per (row, token, frame))
_CTC_PSI_PREPARED = os.environ.get("SC_CTC_PSI_PREPARED", "1") == "1"  # (A/B knob: 0 = every score workgroup derives exp(phi - a) from r itself)

eps = 1e-20
minus_inf = -1e20
REPLAY_GRAPHS = True  # set False to launch every step eagerly (instrumented runs: per-launch timing sees nothing inside a replay)


class BeamHypotheses:
    """N-best list of one sentence (beam_search.py:21-103): ``add`` keeps the ``beam_size`` best length-normalised scores,
    ``is_done`` tells whether no running hypothesis can still beat the worst kept one."""

    def __init__(self, beam_size: int, max_length: int, length_penalty: float):
        self.max_length = max_length - 1
        self.beam_size = beam_size
        self.length_penalty = length_penalty
        self.beams: List = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs: float):
        score = sum_logprobs / ((len(hyp) + eps) ** self.length_penalty)
        if len(self) < self.beam_size or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.beam_size:
                ranked = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, curr_len: int = None) -> bool:
        if len(self) < self.beam_size:
            return False
        if curr_len is None:
            curr_len = self.max_length
        return best_sum_logprobs / ((curr_len + eps) ** self.length_penalty) < self.worst_score


def _first_output(out):
    return out[0] if isinstance(out, (tuple, list)) else out


class _Session:
    """Device side of the cached beam search for one geometry (sentences, beam, vocabulary, frames, length limit, scoring constants):
    the buffers every step reads and writes, and one hipGraph per position.  Kept on the decode function between calls."""

    def __init__(self, fn, B, beam, V, T, max_len, dev, ctc_weight, temperature, pad, sos_eos, eos_filtering):
        self.fn, self.B, self.beam, self.V, self.T, self.max_len, self.dev = fn, B, beam, V, T, max_len, dev
        self.ctc_weight, self.temperature, self.pad, self.sos_eos, self.eos_filtering = ctc_weight, temperature, pad, sos_eos, eos_filtering
        N = B * beam
        self.tokens = torch.zeros(N, dtype=torch.long, device=dev)
        self.prev_tokens = torch.zeros(N, dtype=torch.long, device=dev)
        self.beam_idx = torch.zeros(N, dtype=torch.long, device=dev)
        self.scores = torch.zeros(N, dtype=torch.float32, device=dev)
        self.pack = torch.zeros(B, (6 if eos_filtering else 4) * beam, dtype=torch.float64, device=dev)
        # pinned host mirrors: the step's record comes down and the next rows go up without pageable staging copies (each of those costs a
        # host-side allocation and a blocking transfer -- four per step)
        self.pack_host = [torch.zeros(self.pack.shape, dtype=torch.float64).pin_memory() for _ in range(2)]  # (position p -> mirror p & 1)
        self.fetched = [torch.cuda.Event() for _ in range(2)]
        # the next rows are chosen on the device (sc_beam_advance) when the two-stage top-k path runs; otherwise the host uploads them
        self.ahead = 2 * beam <= 64 and os.environ.get("SC_DECODE_AHEAD", "1") == "1"  # (A/B knob: 0 = the host uploads the rows, step by step)
        self.up_scores = torch.zeros(N, dtype=torch.float32).pin_memory()
        self.up_tokens = torch.zeros(N, dtype=torch.long).pin_memory()
        self.up_parent = torch.zeros(N, dtype=torch.long).pin_memory()
        self.ctc = ctc_weight > 0
        if self.ctc:
            self.x = torch.empty(B, T, V, dtype=torch.float32, device=dev)
            self.enc_lens = torch.empty(B, dtype=torch.int32, device=dev)
            self.eos_frame = torch.empty(N, dtype=torch.int32, device=dev)
            self.r = [torch.empty(T, 2, N, dtype=torch.float32, device=dev) for _ in range(2)]
            self.psi = [torch.empty(N, V, dtype=torch.float32, device=dev) for _ in range(2)]
            self.psi_sel = torch.empty(N, dtype=torch.float32, device=dev)
            # factors of the prefix scores' exponential, made once per batch in begin() (Fn.ctc_prefix_factors, into these static buffers)
            self.E = torch.empty(B, T, V, dtype=torch.float32, device=dev) if _CTC_PSI_FAST else None
            self.bmax = torch.empty(B, V, dtype=torch.float32, device=dev) if _CTC_PSI_FAST else None
            # P = exp(phi - a) of the rows, written by the state launch of a position for the score launch of the next
            self.prep = (Fn.ctc_prefix_prepared(B, beam, T, dev)
                         if (self.E is not None and V % 4 == 0 and T >= 2 and _CTC_PSI_PREPARED) else None)
        self.cache, self.graphs, self.pool, self.calls = None, {}, None, 0
        self.use_graphs = os.environ.get("SC_DECODE_NO_GRAPH", "0") != "1"

    def begin(self, enc_feat, enc_feat_mask, scorer: Optional[CTCPrefixScorer]):
        self.calls += 1
        if self.graphs:
            Fn.weights.sync()  # (the replayed positions read cached weight copies by address: a parameter update since the capture must reach them)
        self.cache = self.fn.start_decoding(enc_feat, enc_feat_mask, self.beam, self.max_len, reuse=self.cache)
        self.tokens.fill_(self.sos_eos)
        self.scores.fill_(float("-inf"))
        self.scores[::self.beam] = 0.0  # one live beam per sentence at the first step (:303-311)
        if self.ctc:
            self.x.copy_(scorer.x)
            self.enc_lens.copy_(scorer.enc_lens)
            self.eos_frame.copy_(scorer.eos_frame)
            self.r[0].copy_(Fn.ctc_prefix_init(self.x, self.enc_lens, self.beam, self.pad))
            if self.E is not None:
                self.bmax.copy_(self.x.max(dim=1).values)
                torch.sub(self.x, self.bmax[:, None, :], out=self.E)
                self.E.exp_()

    def _device_step(self, pos: int):
        B, beam, V, w = self.B, self.beam, self.V, self.ctc_weight
        with torch.no_grad():
            # The CTC side of a step (forward variables of the rows chosen at the last step, then the prefix scores of all their extensions:
            # 25 + 100 us) reads the tokens and its own state only.  SC_DECODE_CTC_BRANCH=1 runs it on a second stream next to the decoder's
            # launches and joins in front of the candidate scores (the default of round 2; see _CTC_BRANCH for why not any more).
            psi = None
            side = Fn._Side(self.x, tag="ctcdec") if (self.ctc and _CTC_BRANCH) else None
            if self.ctc:
                with (side if side is not None else contextlib.nullcontext()):
                    if pos > 0:
                        Fn.ctc_prefix_state(self.x, self.enc_lens, self.r[pos & 1 ^ 1], self.prev_tokens, pos - 1, self.beam_idx, self.tokens, beam,
                                            self.pad, out=self.r[pos & 1], psi_prev=self.psi[pos & 1 ^ 1], psi_sel=self.psi_sel, prepared=self.prep)
                    psi = Fn.ctc_prefix_scores(self.x, self.enc_lens, self.eos_frame, self.r[pos & 1], self.tokens if pos > 0 else None, pos,
                                               beam, self.pad, self.sos_eos, out=self.psi[pos & 1],
                                               factors=(self.E, self.bmax) if self.E is not None else None,
                                               prepared=self.prep if pos > 0 else None)
            self.cache.pos = pos
            logits = self.fn.forward_step(self.tokens, self.cache, self.beam_idx if pos > 0 else None)
            if side is not None:
                side.join()
            fast_tail = self.ahead and beam > 1 and beam * 2 * beam <= 1024 and 2 * beam <= V <= 12288 and _FUSED_TAIL
            if fast_tail:
                # candidate scores + every row's top 2 * beam without the (N, V) matrix, then merge + record + next rows: two launches
                Fn.beam_step_select(logits, self.temperature, psi, self.psi_sel if (self.ctc and pos > 0) else None, w, self.pad, self.sos_eos,
                                    self.eos_filtering, beam, 2 * beam, self.pack, self.scores, self.tokens, self.prev_tokens, self.beam_idx)
                return
            # log-softmax of the token head, CTC mixing (:334-345) and the rows' running scores in one kernel
            cand, eos_score, other_max = Fn.beam_candidates(logits, self.scores, self.temperature, psi, self.psi_sel if (self.ctc and pos > 0) else None,
                                                            w, self.pad, self.sos_eos, want_eos_stats=self.eos_filtering)
            # (two stages: every hypothesis row's own top 2 * beam, then the merge per sentence -- 512 workgroups over 5000 columns instead of
            #  32 over 80 000; same result as one selection over the sentence's beam * V candidates, ties included)
            if self.ahead and beam > 1 and beam * 2 * beam <= 1024 and V >= 2 * beam:
                # (merge, record and next rows in one launch behind the per-row top-k)
                Fn.beam_select_advance(cand.view(B * beam, V), beam, 2 * beam, eos_score if self.eos_filtering else None,
                                       other_max if self.eos_filtering else None, self.sos_eos, self.pack, self.scores, self.tokens, self.prev_tokens,
                                       self.beam_idx)
                return
            top_scores, top_ids = Fn.topk_groups(cand.view(B * beam, V), beam, 2 * beam) if 2 * beam <= 64 else \
                torch.topk(cand.view(B, beam * V), 2 * beam, dim=1, largest=True, sorted=True)
            if self.ahead:
                Fn.beam_advance(top_scores, top_ids, eos_score if self.eos_filtering else None, other_max if self.eos_filtering else None, beam, V,
                                self.sos_eos, self.pack, self.scores, self.tokens, self.prev_tokens, self.beam_idx)
                return
            if top_scores.dtype == torch.float32 and top_scores.is_contiguous() and top_ids.is_contiguous():
                Fn.beam_pack(top_scores, top_ids, eos_score if self.eos_filtering else None, other_max if self.eos_filtering else None, beam,
                             self.pack)  # (one launch instead of four casts, a cat and a copy)
            else:
                fetch = [top_scores.to(torch.float64), top_ids.to(torch.float64)]
                if self.eos_filtering:
                    fetch += [eos_score.view(B, beam).to(torch.float64), other_max.view(B, beam).to(torch.float64)]
                self.pack.copy_(torch.cat(fetch, dim=1))
            self.prev_tokens.copy_(self.tokens)

    def launch(self, pos: int):
        """Enqueues position `pos` on the rows described by tokens / beam_idx / scores and the copy of its record to the host."""
        if self.use_graphs and REPLAY_GRAPHS and self.calls >= 2:  # (the first call runs eagerly: it also fills the cached bf16 weights)
            g = self.graphs.get(pos)
            if g is None:
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # (no cyclic collection while the stream is capturing: a pass that happens to free an older session's graphs or tensors
                #  calls into the runtime in the middle of the capture, which aborts the process)
                was_enabled = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                        self._device_step(pos)
                finally:
                    if was_enabled:
                        gc.enable()
                self.graphs[pos] = g
            g.replay()
        else:
            self._device_step(pos)
        self.pack_host[pos & 1].copy_(self.pack, non_blocking=True)  # the step's one device-to-host copy
        self.fetched[pos & 1].record()

    def fetch(self, pos: int) -> np.ndarray:
        """The packed candidates of position `pos` on the host (waits for that position's copy only)."""
        self.fetched[pos & 1].synchronize()
        return self.pack_host[pos & 1].numpy().copy()

    def step(self, pos: int) -> np.ndarray:
        self.launch(pos)
        return self.fetch(pos)

    def upload(self, scores: np.ndarray, tokens: np.ndarray, parent: np.ndarray):
        """The next step's rows: host arrays -> pinned mirrors -> device, asynchronously (the mirrors are rewritten only after the next
        step's synchronisation)."""
        self.up_scores.numpy()[:] = scores
        self.up_tokens.numpy()[:] = tokens
        self.up_parent.numpy()[:] = parent
        self.scores.copy_(self.up_scores, non_blocking=True)
        self.tokens.copy_(self.up_tokens, non_blocking=True)
        self.beam_idx.copy_(self.up_parent, non_blocking=True)


def _session_for(fn, key, *args) -> _Session:
    store = getattr(fn, "_sc_decode_sessions", None)
    if store is None:
        store = {}
        try:
            object.__setattr__(fn, "_sc_decode_sessions", store)
        except Exception:
            return _Session(fn, *args)
    if key not in store:
        if len(store) >= 4:  # (a session holds caches and graphs: keep the few most recent geometries)
            store.pop(next(iter(store)))
        store[key] = _Session(fn, *args)
    return store[key]


def beam_searching(enc_feat: torch.Tensor, enc_feat_mask: torch.Tensor, asr_decode_fn: Callable, vocab_size: int, sos_eos: int = None,
                   padding_idx: int = 0, beam_size: int = 1, min_f2t_ratio: float = 3.0, length_penalty: float = 1.0,
                   temperature: float = 1.0, eos_filtering: bool = False, eos_threshold: float = 1.5, ctc_weight: float = 0.0,
                   ctc_decode_fn: Optional[Callable] = None, ctc_temperature: float = 1.0, lm_weight: float = 0.2,
                   lm_temperature: float = 1.0, lm_decode_fn: Optional[Callable] = None, lm_window_size: int = None,
                   ilm_sub_weight: float = 0.0, sent_per_beam: int = 1) -> Dict[str, torch.Tensor]:
    if sent_per_beam > 1:
        raise NotImplementedError("Currently, sent_per_beam > 1 is not supported....")
    assert 0.0 <= ctc_weight, f"ctc_weight should be a non-negative float number, but got ctc_weight={ctc_weight}!"
    assert 0.0 <= lm_weight, f"lm_weight should be a non-negative float number, but got lm_weight={lm_weight}!"
    assert 0.0 <= ilm_sub_weight, f"ilm_sub_weight should be a non-negative float number, but got ilm_sub_weight={ilm_sub_weight}!"
    assert ctc_temperature > 0.0, f"ctc_temperature should be a positive float number, but got ctc_temperature={ctc_temperature}"
    assert temperature > 0.0, f"temperature should be a positive float number, but got temperature={temperature}"
    assert lm_temperature > 0.0, f"lm_temperature should be a positive float number, but got lm_temperature={lm_temperature}"
    assert beam_size + 1 <= vocab_size, "beam_size cannot be larger than vocab_size - 1 (exclude <sos/eos>)!"

    B, feat_maxlen = enc_feat.size(0), enc_feat.size(1)
    dev = enc_feat.device
    enc_feat_len = enc_feat_mask.squeeze(dim=1).sum(dim=-1)
    hypo_maxlen = int(feat_maxlen / min_f2t_ratio) if min_f2t_ratio > 0 else int(-min_f2t_ratio)
    if sos_eos is None:
        sos_eos = vocab_size - 1
    N, V = B * beam_size, vocab_size
    cached = hasattr(asr_decode_fn, "start_decoding") and hasattr(asr_decode_fn, "forward_step")
    use_lm = lm_decode_fn is not None and lm_weight > 0
    if cached and not use_lm and feat_maxlen % 32 != 0:
        # the replayed session is keyed by its geometry: pad the frames to a multiple of 32 (masked out everywhere: cross-attention by the
        # mask, the CTC scorer by the lengths) so that utterances of nearby lengths share one session and its hipGraphs
        extra = 32 - feat_maxlen % 32
        enc_feat = torch.nn.functional.pad(enc_feat, (0, 0, 0, extra))
        enc_feat_mask = torch.cat([enc_feat_mask, torch.zeros(B, enc_feat_mask.size(1), extra, dtype=enc_feat_mask.dtype, device=dev)], dim=-1)
    # ---- CTC scorer: one copy of the log-probabilities per sentence ----
    ctc_scorer, ctc_memory = None, None
    if ctc_decode_fn is not None and ctc_weight > 0:
        with torch.no_grad():
            ctc_logits = _first_output(ctc_decode_fn(enc_feat)).float().clone()
        ctc_logits[:, :, sos_eos] = minus_inf  # CTC never predicts <sos/eos> (:279)
        ctc_scorer = CTCPrefixScorer(x=torch.log_softmax(ctc_logits / ctc_temperature, dim=-1), enc_lens=enc_feat_len, batch_size=B,
                                     beam_size=beam_size, blank_index=padding_idx, eos_index=sos_eos)

    # ---- decoder state ----
    session = None
    if cached and not use_lm:
        w = float(ctc_weight) if ctc_scorer is not None else 0.0
        frames, cache_len = enc_feat.size(1), (max(hypo_maxlen, 1) + 15) // 16 * 16  # (padded frames; cache positions in steps of 16)
        key = (B, beam_size, V, frames, cache_len, str(enc_feat.dtype), str(dev), w, float(temperature), padding_idx, sos_eos,
               bool(eos_filtering), str(Fn.compute_dtype()))
        session = _session_for(asr_decode_fn, key, B, beam_size, V, frames, cache_len, dev, w, float(temperature), padding_idx,
                               sos_eos, bool(eos_filtering))
        session.begin(enc_feat, enc_feat_mask, ctc_scorer)
    if session is None:  # (with a language model in the loop the decoder is called the reference's way as well)
        enc_rep = enc_feat.repeat_interleave(beam_size, dim=0)
        mask_rep = enc_feat_mask.repeat_interleave(beam_size, dim=0)

    generated_hyps = [BeamHypotheses(beam_size, hypo_maxlen, length_penalty) for _ in range(B)]
    done = [False] * B
    beam_scores = torch.full((N,), float("-inf"), dtype=torch.float, device=dev)
    beam_scores[::beam_size] = 0.0  # one live beam per sentence at the first step (:303-311)
    text = np.full((N, 1), sos_eos, dtype=np.int64)  # the hypotheses, on the host
    text_len = np.ones(N, dtype=np.int64)
    slot_offset = (torch.arange(beam_size, device=dev) * V)[None, :]

    def prefix_step():
        """The reference's calling convention: decoder (and language model) on the whole prefix; returns the packed candidates."""
        nonlocal ctc_memory
        with torch.no_grad():
            text_dev = torch.from_numpy(text).to(dev)
            len_dev = torch.from_numpy(text_len).to(dev)
            logits = _first_output(asr_decode_fn(enc_feat=enc_rep, enc_feat_mask=mask_rep, text=text_dev, text_len=len_dev))[:, -1, :]
            next_token_scores = torch.log_softmax(logits.float() / temperature, dim=-1)
            if ctc_scorer is not None:
                next_token_scores[:, padding_idx] = minus_inf  # the blank is never a token (:339)
                ctc_token_scores, ctc_memory = ctc_scorer.forward_step(text_dev[:, 1:], ctc_memory)
                next_token_scores = (1 - ctc_weight) * next_token_scores + ctc_weight * ctc_token_scores
            if use_lm:  # (:348-398) a language model is called the way the reference calls it: on the (windowed) prefix
                lm_text = text_dev if lm_window_size is None else text_dev[:, -lm_window_size:]
                lm_len = len_dev if lm_window_size is None else torch.clamp(len_dev, max=lm_window_size)
                lm_out = _first_output(lm_decode_fn(text=lm_text, text_len=lm_len))[:, -1, :]
                next_token_scores = next_token_scores + lm_weight * torch.log_softmax(lm_out.float() / lm_temperature, dim=-1)
                if ilm_sub_weight > 0:
                    zeros = torch.zeros(N, 1, enc_feat.size(2), dtype=enc_feat.dtype, device=dev)
                    ones = torch.ones(N, 1, 1, dtype=torch.bool, device=dev)
                    ilm = _first_output(asr_decode_fn(enc_feat=zeros, enc_feat_mask=ones, text=lm_text, text_len=lm_len))[:, -1, :]
                    next_token_scores = next_token_scores - ilm_sub_weight * torch.log_softmax(ilm.float(), dim=-1)
            cand = (next_token_scores + beam_scores[:, None]).view(B, beam_size * V)
            top_scores, top_ids = torch.topk(cand, 2 * beam_size, dim=1, largest=True, sorted=True)
            fetch = [top_scores, top_ids]
            if eos_filtering:
                others = next_token_scores.clone()
                others[:, sos_eos] = float("-inf")
                fetch += [next_token_scores[:, sos_eos].view(B, beam_size), others.max(dim=-1).values.view(B, beam_size)]
            return torch.cat([f.to(torch.float64) for f in fetch], dim=1).cpu().numpy()  # the step's one device-to-host copy

    step_no = 0
    ahead = session is not None and session.ahead
    launched = -1  # (the last position enqueued on the device)
    final_scores_host = None
    while int(text_len.max()) < hypo_maxlen:
        if ahead:
            # the device picks its own next rows: position step_no + 1 is enqueued before the host reads the record of step_no (every running row
            # is step_no + 2 tokens long after this iteration, so a next iteration exists exactly when step_no + 2 < hypo_maxlen)
            if launched < step_no:
                session.launch(step_no)
                launched = step_no
            if step_no + 2 < hypo_maxlen:
                session.launch(step_no + 1)
                launched = step_no + 1
            host = session.fetch(step_no)
        else:
            host = session.step(step_no) if session is not None else prefix_step()
        step_no += 1
        h_scores = host[:, :2 * beam_size].astype(np.float32)
        h_ids = host[:, 2 * beam_size:4 * beam_size].astype(np.int64)
        if eos_filtering:
            h_eos = host[:, 4 * beam_size:5 * beam_size].astype(np.float32)
            h_ref = host[:, 5 * beam_size:6 * beam_size].astype(np.float32)

        # ---- selection (:423-509) ----
        # The reference walks a sentence's 2 * beam candidates in rank order: an <eos> candidate ends a hypothesis (if it ranks inside the
        # beam and passes the filter) and is skipped, the others are kept until `beam` rows are filled.  A row holds one <eos> token, so at most
        # `beam` of the candidates are <eos> and the kept ones are simply the first `beam` others -- taken for all sentences at once; only
        # the <eos> candidates in front of the last kept rank (rare) go through a Python loop.
        new_scores = np.zeros(N, dtype=np.float32)
        new_tokens = np.full(N, padding_idx, dtype=np.int64)
        new_parent = np.zeros(N, dtype=np.int64)  # (finished sentences: padding tokens continuing row 0, score 0 (:428-431))
        slots_all, tokens_all = h_ids // V, h_ids % V
        is_eos = tokens_all == sos_eos
        order = np.argsort(is_eos, axis=1, kind="stable")[:, :beam_size]  # ranks of the first `beam` non-<eos> candidates, in rank order
        r_last = order[:, -1]
        ranks = np.arange(2 * beam_size)[None, :]
        eos_live = is_eos & (ranks < beam_size) & (ranks < r_last[:, None])
        active = ~np.asarray(done, dtype=bool)
        bcol = np.arange(B)[:, None]
        act = np.nonzero(active)[0]
        new_scores.reshape(B, beam_size)[act] = h_scores[bcol, order][act]
        new_tokens.reshape(B, beam_size)[act] = tokens_all[bcol, order][act]
        new_parent.reshape(B, beam_size)[act] = (bcol * beam_size + slots_all[bcol, order])[act]
        for b in np.nonzero(eos_live.any(axis=1) & active)[0]:
            for rank in np.nonzero(eos_live[b])[0]:
                slot = int(slots_all[b, rank])
                if eos_filtering and h_eos[b, slot] <= eos_threshold * h_ref[b, slot]:
                    continue
                generated_hyps[b].add(torch.from_numpy(text[b * beam_size + slot, 1:].copy()), float(h_scores[b, rank]))
        best_now, len_now = h_scores.max(axis=1), text_len.reshape(B, beam_size).max(axis=1) - 1
        for b in act:
            done[b] = generated_hyps[b].is_done(best_sum_logprobs=float(best_now[b]), curr_len=int(len_now[b]))
        if all(done):
            break

        # ---- the next step's rows ----
        text = np.concatenate([text[new_parent], new_tokens[:, None]], axis=1)
        text_len = (text != padding_idx).sum(-1)
        if ahead:
            final_scores_host = new_scores  # (the device rows are already a position further)
            continue
        if session is not None:
            session.upload(new_scores, new_tokens, new_parent)
            beam_scores = session.scores
            continue
        beam_scores = torch.from_numpy(new_scores).to(dev)
        last_tokens = torch.from_numpy(new_tokens).to(dev)
        beam_idx_dev = torch.from_numpy(new_parent).to(dev)
        enc_rep, mask_rep = enc_rep[beam_idx_dev], mask_rep[beam_idx_dev]
        if ctc_memory is not None:
            ctc_memory = ctc_scorer.permute_mem(ctc_memory, beam_idx_dev, last_tokens.view(B, beam_size) + slot_offset)

    # ---- hypotheses cut by the length limit (:529-541) ----
    final_scores = final_scores_host if final_scores_host is not None else beam_scores.cpu().numpy()
    for b in range(B):
        if done[b]:
            continue
        for slot in range(beam_size):
            row = b * beam_size + slot
            generated_hyps[b].add(torch.from_numpy(text[row, 1:].copy()), float(final_scores[row]))

    # ---- best hypothesis of every sentence, padded (:543-580) ----
    best = [sorted(h.beams, key=lambda x: x[0])[-1] for h in generated_hyps]
    hypo_text_len = torch.tensor([len(h) for _, h in best], dtype=torch.long, device=dev)
    if int(hypo_text_len.min()) != int(hypo_text_len.max()):
        sent_max_len = min(int(hypo_text_len.max()), hypo_maxlen)
        hypo_text = torch.full((B, sent_max_len), padding_idx, dtype=torch.long, device=dev)
        for i, (_, h) in enumerate(best):
            hypo_text[i, :len(h)] = h.to(dev)
    else:
        hypo_text = torch.stack([h for _, h in best]).to(dev)
    return dict(hypo_text=hypo_text, hypo_text_len=hypo_text_len, feat_token_len_ratio=enc_feat_len / (hypo_text_len + 1e-10),
                hypo_text_confid=torch.Tensor([s for s, _ in best]))

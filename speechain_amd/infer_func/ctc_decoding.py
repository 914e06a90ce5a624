"""``CTCPrefixScorer``, drop-in for ``speechain.infer_func.ctc_decoding.CTCPrefixScorer`` (speechain/infer_func/ctc_decoding.py:6-196):
CTC prefix scores of every candidate token of every hypothesis (Algorithm 2 of Watanabe et al., "Hybrid CTC/attention architecture for
end-to-end speech recognition"), same constructor, ``forward_step(g, state)`` and ``permute_mem(memory, beam_idx, token_idx)``.

What differs underneath: the reference materialises the forward variables of ALL candidates, a (T, 2, N, V) tensor per step
(5 GB at T 249, N 512, V 5000), in T rounds of stacked logsumexp launches.  ``sc_ctc_prefix_scores`` runs the recursion of one
(hypothesis, candidate) pair per lane in registers and writes only psi (N, V); the forward variables of the candidates that survive the
beam selection -- all that the next step reads -- are recomputed for those N pairs by ``sc_ctc_prefix_state``.  The log-probabilities are
kept once per sentence (B, T, V): the beams of a sentence share them (speechain/infer_func/beam_search.py:258-274 repeats them).

Quirk kept: the <eos> score reads the last frame through ``last_frame_index[i // beam_size]`` with ``last_frame_index`` indexed by
hypothesis ROW (ctc_decoding.py:47,151-155): row ``i // beam_size`` is a row of sentence ``i // beam_size // beam_size``."""
import torch

from speechain_amd import functional as Fn


class CTCPrefixScorer:
    def __init__(self, x: torch.Tensor, enc_lens: torch.Tensor, batch_size: int, beam_size: int, blank_index: int, eos_index: int):
        """x: (batch_size * beam_size, T, V) log-probabilities, rows of a sentence adjacent and equal (or (batch_size, T, V): one copy
        per sentence); enc_lens: (batch_size * beam_size,) or (batch_size,)."""
        Fn.require_gpu(x, "CTCPrefixScorer")
        self.blank_index, self.eos_index = blank_index, eos_index
        self.batch_size, self.beam_size = batch_size, beam_size
        self.max_enc_len, self.vocab_size = x.size(1), x.size(-1)
        self.device = x.device
        self.minus_inf = -1e20
        N = batch_size * beam_size
        if x.size(0) == N and beam_size > 1:
            x = x[::beam_size]
        if enc_lens.numel() == N and beam_size > 1:
            row_lens = enc_lens
            enc_lens = enc_lens[::beam_size]
        else:
            row_lens = enc_lens.repeat_interleave(beam_size)
        self.enc_lens = enc_lens.to(device=x.device, dtype=torch.int32).contiguous()
        # the frame the <eos> score of row i reads (see the quirk above)
        rows = torch.arange(N, device=x.device)
        self.eos_frame = (row_lens.to(x.device)[rows // beam_size] - 1).to(torch.int32).contiguous()
        self.x = x.float().contiguous()  # (B, T, V); frames past an utterance are handled in the kernels (blank certain, labels impossible)

    def forward_step(self, g: torch.Tensor, state):
        """g: (N, |g|) prefixes without <sos>.  Returns (psi - psi_prev (N, V), (memory, psi)) like the reference; ``memory`` holds what
        ``permute_mem`` needs to rebuild the forward variables of the chosen candidates."""
        N, plen = g.size(0), g.size(1)
        if state is None:
            r_prev, psi_prev = None, None
        else:
            r_prev, psi_prev = state
        last = g[:, -1].contiguous() if plen > 0 else None
        psi = Fn.ctc_prefix_scores(self.x, self.enc_lens, self.eos_frame, r_prev, last, plen, self.beam_size, self.blank_index, self.eos_index)
        delta = psi if psi_prev is None else psi - psi_prev
        return delta, ((r_prev, last, plen), psi)

    def permute_mem(self, memory, beam_idx: torch.Tensor, token_idx: torch.Tensor):
        """Forward variables r (T, 2, N) and prefix scores psi (N, V: every column the chosen candidate's score) of the new rows:
        row n continues row beam_idx[n] with token ``token_idx[n] - V * slot`` (the reference's packing, beam_search.py:523-527)."""
        (r_prev, last, plen), psi = memory
        N, V = psi.shape
        slot = torch.arange(self.beam_size, device=psi.device) * V
        tokens = (token_idx - slot[None, :]).reshape(-1).contiguous()
        r = Fn.ctc_prefix_state(self.x, self.enc_lens, r_prev, last, plen, beam_idx, tokens, self.beam_size, self.blank_index)
        psi_sel = psi[beam_idx, tokens]
        return r, psi_sel[:, None].expand(N, V)

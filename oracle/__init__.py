"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the SpeeChain hot path.

Nothing under ``speechain_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as
the checker / the timed CPU baseline -- never as the thing shipped.

Parity status: pinned by golden vectors generated from the imported reference modules in the
build container (``tests/golden/make_golden.py``); the two exceptions live in the absent
third-party ``torchaudio`` -> "parity unpinned" at those boundaries: the mel filterbank matrix
(``oracle/melscale.py``) and the Griffin-Lim iteration (``cpu_ref.griffin_lim``), both restated
from the published algorithms.  ``oracle/decode_ref.py`` holds the
inference callers (beam search, CTC prefix scores), pinned by ``g11_decode``.
"""

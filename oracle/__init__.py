"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the SpeeChain hot path.

Nothing under ``speechain_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as
the checker / the timed CPU baseline -- never as the thing shipped.

Parity status: pinned by golden vectors generated from the imported reference modules in the
build container (``tests/golden/make_golden.py``); the one exception is the mel filterbank
matrix, whose arithmetic lives in the absent third-party ``torchaudio`` -> "parity unpinned"
at that boundary (see ``oracle/melscale.py`` and DESIGN.md).
"""

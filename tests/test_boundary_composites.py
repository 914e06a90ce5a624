"""The drop-in boundary, through the reference's OWN composite classes: SpeeChain builds its models from YAML ``type:``
strings (``import_class("speechain.module." + type)``, module/encoder/asr.py:45,72,78; decoder/ar_asr.py:33,41; encoder/tts.py:43,52,58;
decoder/nar_tts.py:56-146) and from direct imports of SpecAugment / FeatureNormalization / TokenPostnet.  After
``register(leaf_only=True)`` the reference's ASREncoder, ARASRDecoder, TTSEncoder and FastSpeech2Decoder, constructed from the recipes'
dictionaries, must come out with exactly the state-dict keys and shapes of an unmodified checkout -- so its checkpoints load, strictly -- while
every member module is a speechain_amd one.  Build container only: skipped where /root/reference is absent (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/speechain"), reason="needs the reference checkout (build container only)")


def _probe(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_composites_probe.py"), mode], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_reference_composites_over_the_overlaid_leaves_keep_their_state_dicts():
    plain, leaf = _probe("plain"), _probe("leaf")
    for name in ("ASREncoder", "ARASRDecoder", "TTSEncoder", "FastSpeech2Decoder"):
        assert plain["origin"][name] == ["speechain"], plain["origin"]
        assert "speechain_amd" in leaf["origin"][name], leaf["origin"]
        a, b = plain[name], leaf[name]
        assert set(a) == set(b), (name, sorted(set(a) ^ set(b))[:20])
        diff = {k: (a[k], b[k]) for k in a if a[k] != b[k]}
        assert not diff, (name, list(diff.items())[:10])
        assert plain[name + ".output_size"] == leaf[name + ".output_size"], name
        assert len(a) > 10
    # every member of the overlaid build except the composite itself is a speechain_amd module: the only 'speechain' class left is the wrapper
    assert leaf["origin"]["ASREncoder"] == ["speechain", "speechain_amd"]

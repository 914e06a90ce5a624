"""Installs the speechain_amd classes under the ``speechain.module.*`` / ``speechain.criterion.*`` names.

SpeeChain resolves its YAML ``type:`` strings with ``import_class("speechain.module." + type)``
(speechain/utilbox/import_util.py:17-22; call sites module/encoder/asr.py:45,72,78, module/decoder/ar_asr.py:33,41, ...),
and imports ``SpecAugment`` / ``TokenPostnet`` / the criteria directly.  After ``register()`` those names resolve to the
HIP-backed implementations, so an unmodified SpeeChain checkout builds its models out of them:

    import speechain_amd.register as r; r.register()      # before `import speechain.model...`

With SpeeChain importable, only the listed leaf modules are replaced (its package ``__path__`` keeps serving everything
else); without it, lightweight namespace packages are created so the dotted names still import.
"""
import importlib
import sys
import types

OVERLAY = {
    "speechain.module.abs": "speechain_amd.module.abs",
    "speechain.module.frontend.speech2linear": "speechain_amd.module.frontend.speech2linear",
    "speechain.module.frontend.linear2mel": "speechain_amd.module.frontend.linear2mel",
    "speechain.module.frontend.speech2mel": "speechain_amd.module.frontend.speech2mel",
    "speechain.module.frontend.delta_feat": "speechain_amd.module.frontend.delta_feat",
    "speechain.module.augment.specaug": "speechain_amd.module.augment.specaug",
    "speechain.module.norm.feat_norm": "speechain_amd.module.norm.feat_norm",
    "speechain.module.transformer.pos_enc": "speechain_amd.module.transformer.pos_enc",
    "speechain.module.transformer.attention": "speechain_amd.module.transformer.attention",
    "speechain.module.transformer.feed_forward": "speechain_amd.module.transformer.feed_forward",
    "speechain.module.transformer.encoder": "speechain_amd.module.transformer.encoder",
    "speechain.module.transformer.decoder": "speechain_amd.module.transformer.decoder",
    "speechain.module.prenet.conv2d": "speechain_amd.module.prenet.conv2d",
    "speechain.module.prenet.linear": "speechain_amd.module.prenet.linear",
    "speechain.module.prenet.conv1d": "speechain_amd.module.prenet.conv1d",
    "speechain.module.prenet.var_pred": "speechain_amd.module.prenet.var_pred",
    "speechain.module.prenet.embed": "speechain_amd.module.prenet.embed",
    "speechain.module.prenet.spk_embed": "speechain_amd.module.prenet.spk_embed",
    "speechain.module.postnet.conv1d": "speechain_amd.module.postnet.conv1d",
    "speechain.module.encoder.tts": "speechain_amd.module.encoder.tts",
    "speechain.module.encoder.asr": "speechain_amd.module.encoder.asr",
    "speechain.module.decoder.ar_asr": "speechain_amd.module.decoder.ar_asr",
    "speechain.infer_func.ctc_decoding": "speechain_amd.infer_func.ctc_decoding",
    "speechain.infer_func.beam_search": "speechain_amd.infer_func.beam_search",
    "speechain.module.decoder.nar_tts": "speechain_amd.module.decoder.nar_tts",
    "speechain.criterion.least_error": "speechain_amd.criterion.least_error",
    "speechain.module.postnet.token": "speechain_amd.module.postnet.token",
    "speechain.criterion.cross_entropy": "speechain_amd.criterion.cross_entropy",
    "speechain.criterion.ctc": "speechain_amd.criterion.ctc",
    "speechain.utilbox.train_util": None,  # only patched (make_mask_from_len), never replaced wholesale
}


def _ensure_package(name: str):
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
        parent, _, child = name.rpartition(".")
        if parent:
            setattr(_ensure_package(parent), child, pkg)
        return pkg


# SpeeChain's own composite classes (they only string other modules together): with ``leaf_only=True`` they stay the reference's and
# resolve their members -- by ``import_class`` string or by direct import -- to the overlaid leaves (tests/test_boundary_composites.py)
COMPOSITES = ("speechain.module.encoder.asr", "speechain.module.encoder.tts", "speechain.module.decoder.ar_asr",
              "speechain.module.decoder.nar_tts")


def register(verbose: bool = False, leaf_only: bool = False):
    installed = []
    for target, source in OVERLAY.items():
        if source is None or (leaf_only and target in COMPOSITES):
            continue
        mod = importlib.import_module(source)
        parent, _, child = target.rpartition(".")
        pkg = _ensure_package(parent)
        sys.modules[target] = mod
        setattr(pkg, child, mod)
        installed.append(target)
    # patch the mask helper in place when SpeeChain's utilbox is importable (it has many other functions we keep)
    try:
        tu = importlib.import_module("speechain.utilbox.train_util")
        from speechain_amd.utilbox.train_util import make_mask_from_len

        tu.make_mask_from_len = make_mask_from_len
        installed.append("speechain.utilbox.train_util.make_mask_from_len")
    except Exception:
        pass
    if verbose:
        print("speechain_amd registered:", *installed, sep="\n  ")
    return installed

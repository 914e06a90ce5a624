"""speechain_amd -- MI355X-native (gfx950) implementation of SpeeChain's ASR/TTS compute hot path.

The package mirrors the reference's module layout (``speechain_amd.module.frontend.speech2mel.Speech2MelSpec`` is
the drop-in for ``speechain.module.frontend.speech2mel.Speech2MelSpec`` ...).  ``speechain_amd.register.register()``
installs the classes under the ``speechain.module.*`` names so an unmodified SpeeChain resolves its YAML ``type:``
strings to them.  All arithmetic runs in hand-written HIP kernels behind the C ABI of ``include/speechain_amd.h``.
"""
__version__ = "0.1.0"

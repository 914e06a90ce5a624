"""``import_class`` with the reference's contract (speechain/utilbox/import_util.py:17-22): a dotted ``speechain.module.<...>.<Class>``
string from a YAML ``type:`` field -> the class.  Resolved inside this package first (the HIP-backed drop-ins), so composite modules
built here (encoder/tts.py, decoder/nar_tts.py) find their parts whether or not SpeeChain itself is importable."""
import importlib
from functools import lru_cache


@lru_cache(maxsize=None)
def import_class(class_string: str):
    module_name, _, class_name = class_string.rpartition(".")
    names = [module_name]
    if module_name.startswith("speechain."):
        names.insert(0, "speechain_amd." + module_name[len("speechain."):])
    err = None
    for name in names:
        try:
            return getattr(importlib.import_module(name), class_name)
        except (ImportError, AttributeError) as e:
            err = e
    raise ImportError(f"cannot resolve {class_string}: {err}")

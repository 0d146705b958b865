"""TEST INFRASTRUCTURE ONLY -- import the real wesep reference (when present) with its
absent third-party modules stubbed.

Only `oracle/make_golden.py` uses this, in the authoring container where
`/root/reference` exists.  Nothing on the GPU box may call it (the reference does
not exist there).  Recipe follows SURVEY.md Appendix C.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUBS = [
    "torchaudio", "torchaudio.transforms", "torchaudio.compliance",
    "torchaudio.compliance.kaldi", "wespeaker", "wespeaker.models",
    "wespeaker.models.speaker_model", "silero_vad", "soundfile",
]


class _Raiser(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _missing(*a, **k):
            raise RuntimeError(f"stubbed third-party symbol {self.__name__}.{name} was called")
        return _missing


def import_reference():
    """Returns the reference's `wesep.models.get_model` (stubs installed first)."""
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = _Raiser(name)
        if "." in name:  # `import a.b.c as x` resolves through parent attributes
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from wesep.models import get_model  # noqa: the reference's own factory
    return get_model

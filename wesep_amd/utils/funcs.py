"""`clip_gradients` with the reference's signature (wesep/utils/funcs.py:79-88)."""
from ..optim import clip_gradients  # noqa: F401

from .speaker import FiLM, LinearLayer, SpeakerFuseLayer, SpeakerTransform  # noqa: F401

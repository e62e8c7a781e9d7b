from .optimizer import AdamUniform  # noqa: F401

from .MICFormer_self import Head, MicFormer  # noqa: F401

from . import base, strong_label, weak_label  # noqa: F401

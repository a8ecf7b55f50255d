"""pb_sed_amd: MI355X-native (gfx950) FBCRNN / BiCRNN hot path for pb_sed.

The compute path is hand-written HIP behind the C-ABI of ``include/pbsed.h``
(``pb_sed_amd/libpbsed_mi355.so``); there is no CPU or eager fallback.
"""
from . import _lib  # noqa: F401

"""2fast2q_amd -- MI355X-native feature counting for 2FAST2Q.

The directory name starts with a digit (it mirrors the upstream project name), so import it with
``importlib.import_module("2fast2q_amd")`` or through the ``fast2q_amd`` alias module at the
repository root.  ``binding`` is the ctypes layer over libf2q_hip.so (include/f2q.h); ``fast2q``
is the host harness that mirrors the reference's interface for the path (features_loader,
reads_counter, aligner, compiling, the ``2fast2q -c`` CLI).
"""
from . import binding  # noqa: F401
from .binding import Counter, F2QError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"

"""kmergutsjava_amd -- MI355X (gfx950) implementation of the kmer_guts hot path of KBase's
KmerGutsJava behind the reference's own entry points (KmerGutsJava.main / run / status).

The arithmetic lives in libkmerguts_hip.so (hand-written HIP, C ABI in include/kmerguts_hip.h);
importing the package does not load it, using it does, and a missing library is an error:
there is no CPU fallback.
"""
__version__ = "0.1.0"

from .kmer_guts_java import KmerGutsJava  # noqa: E402,F401  (imports no native code until used)

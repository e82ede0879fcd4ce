"""MI355X-native all-pairs banded DTW + UPGMA (drop-in for the alignment/clustering path of
dkohlsdorf/audio_pattern_discovery).  The compute lives in csrc/ behind the C ABI declared in
include/apd.h; the modules here are the host-side mirror of the reference's Rust interface."""
__version__ = "0.1.0"

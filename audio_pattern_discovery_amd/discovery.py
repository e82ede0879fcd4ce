"""Host mirror of src/discovery.rs: the `Discovery` config struct and `alignment_params`."""
import ctypes as C
from dataclasses import dataclass, fields

from . import _lib
from .alignments import AlignmentParams


@dataclass
class Discovery:
    """discovery.rs:7-26 -- same field names as project/config/Discovery.toml."""
    dft_win: int = 256
    dft_step: int = 128
    ceps_filter: int = 32
    vat_moving: int = 15
    vat_percentile: float = 0.95
    vat_min_len: int = 150
    alignment_workers: int = 4
    clustering_percentile: float = 0.05
    warping_band_percentage: float = 1.0
    insertion_penalty: float = 1.0
    deletion_penalty: float = 1.0
    match_penalty: float = 1.0
    auto_encoder: int = 10
    learning_rate: float = 0.1
    epochs: int = 25
    epoch_drop: float = 5.0
    drop: float = 0.5

    @staticmethod
    def from_toml(file):
        """discovery.rs:28-36 through apd_discovery_parse_toml.  Flat `key = value  # comment` files only (what the
        reference ships); a missing, duplicate or unknown key is an error, as with serde."""
        with open(file) as fp:                      # a missing file raises, like .expect() at :31
            text = fp.read()
        d = _lib.DiscoveryC()
        if _lib.lib().apd_discovery_parse_toml(text.encode(), C.byref(d)) != _lib.APD_OK:
            raise KeyError("not a complete Discovery.toml (missing, duplicate, unknown or ill-typed key)")
        return Discovery(**{f.name: getattr(d, f.name) for f in fields(Discovery)})

    def align_config(self):
        return _lib.AlignConfig(self.warping_band_percentage, self.insertion_penalty,
                                self.deletion_penalty, self.match_penalty)

    def alignment_params(self, n_size):
        """discovery.rs:38-45 through apd_discovery_alignment_params (f32 product, truncation)."""
        out = _lib.AlignmentParamsC()
        cfg = self.align_config()
        _lib.check(_lib.lib().apd_discovery_alignment_params(C.byref(cfg), int(n_size), C.byref(out)))
        return AlignmentParams(int(out.warping_band), out.insertion_penalty, out.deletion_penalty,
                               out.match_penalty)

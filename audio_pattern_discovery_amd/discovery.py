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
        """discovery.rs:29-36.  Flat `key = value  # comment` files only (what the reference ships)."""
        kinds = {f.name: f.type for f in fields(Discovery)}
        vals = {}
        with open(file) as fp:                      # a missing file raises, like .expect() at :31
            for line in fp:
                line = line.split("#", 1)[0].strip()
                if not line or "=" not in line:
                    continue
                k, v = [t.strip() for t in line.split("=", 1)]
                if k not in kinds:
                    raise KeyError("unknown Discovery key %r" % k)
                vals[k] = int(v) if kinds[k] in (int, "int") else float(v)
        missing = [k for k in kinds if k not in vals]
        if missing:
            raise KeyError("missing Discovery keys: %s" % missing)   # serde would fail the same way
        return Discovery(**vals)

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

"""vpd_amd -- MI355X-native VPD student train / apply path (HIP kernels behind
the reference's Python surface).  See DESIGN.md."""
from ._lib import VpdHipError, lib  # noqa: F401

__all__ = ["VpdHipError", "lib"]

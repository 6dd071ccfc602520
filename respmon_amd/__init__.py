"""respmon_amd -- MI355X (gfx950) implementation of respmon's Eulerian-magnification calibration and
ROI motion-extraction hot path behind the reference's own Python surface.  See DESIGN.md."""
__version__ = "0.1.0"

from .tools import reduce_bounding_box, Benchmarker  # noqa: F401


def __getattr__(name):
    # heavy modules (they need the HIP library) are imported on first use
    if name == "RespiratoryMonitor":
        from .base import RespiratoryMonitor
        return RespiratoryMonitor
    if name in ("transforms", "pyramid", "base", "device", "synth", "dist"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)

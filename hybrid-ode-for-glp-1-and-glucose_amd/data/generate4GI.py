"""Same module path as the reference's data/generate4GI.py: `from data.generate4GI import FourGIModel` resolves to the
device implementation (hode/datagen.py) when this package precedes the reference on sys.path."""
from hode.datagen import FourGIModel, grid_points  # noqa: F401

__all__ = ["FourGIModel"]

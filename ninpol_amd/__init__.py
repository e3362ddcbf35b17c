"""ninpol_amd -- MI355X-native nodal interpolation (IDW / LS / GLS weights) behind the
`ninpol.Interpolator` API.  `from ninpol_amd import Interpolator, Grid` mirrors
`ninpol/__init__.py:1-2`.  Importing the package loads nothing native; constructing an
`Interpolator` or `Grid` loads libninpol_amd.so and fails loudly if it is missing."""
from .grid import Grid
from .interpolator import Interpolator, pinned_pool
from .mesh import CellBlock, Mesh
from ._lib import NinpolError

__all__ = ["Interpolator", "Grid", "Mesh", "CellBlock", "NinpolError", "pinned_pool"]
__version__ = "0.1.0"

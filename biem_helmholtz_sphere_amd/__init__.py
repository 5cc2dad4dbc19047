"""MI355X-native implementation of the biem() dense assembly-and-solve hot path.

Drop-in for the public surface of ``biem_helmholtz_sphere`` (reference ``__init__.py:2-24``) on this path:
``biem``, ``BIEMResultCalculator``, ``plane_wave``, ``point_source``, ``max_memory``, ``max_n_end`` and the
typing helpers, plus the coordinate-tree factory the reference imports from ``ultrasphere``.
"""
__version__ = "0.1.0"

from ._biem import (
    BIEMKwargs,
    BIEMResultCalculator,
    BIEMResultCalculatorProtocol,
    UinCallable,
    biem,
    biem_u,
    max_memory,
    max_n_end,
    plane_wave,
    point_source,
)
from ._coords import SphericalCoordinates, create_from_branching_types

__all__ = [
    "BIEMKwargs",
    "BIEMResultCalculator",
    "BIEMResultCalculatorProtocol",
    "UinCallable",
    "biem",
    "biem_u",
    "max_memory",
    "max_n_end",
    "plane_wave",
    "point_source",
    "SphericalCoordinates",
    "create_from_branching_types",
]

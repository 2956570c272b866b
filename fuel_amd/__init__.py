"""fuel_amd -- MI355X (gfx950) implementation of FUEL's mapping-and-planning hot path.

Host mirror of the reference's class interfaces (SDFMap / EDTEnvironment / FrontierFinder /
BsplineOptimizer) over the C-ABI in include/fuelmi.h, implemented by hand-written HIP kernels in
fuel_amd/csrc/.  No CPU fallback exists: importing works anywhere, but any call needs
fuel_amd/libfuelmi.so (built by __graft_entry__.build()) and a gfx950 device.
"""
from ._lib import FuelmiError, LIB_PATH, lib  # noqa: F401
from .host import (DeviceBuffer, RegisteredHostBuffer, BsplineBatchProblem, BsplineOptimizer, NonUniformBspline, EDTEnvironment, FrontierFinder,  # noqa: F401
                   SDFMap, DEFAULT_BSPLINE, DEFAULT_MAP, SMOOTHNESS, DISTANCE, FEASIBILITY, START,
                   END, GUIDE, WAYPOINTS, VIEWCONS, MINTIME, GUIDE_PHASE, NORMAL_PHASE)

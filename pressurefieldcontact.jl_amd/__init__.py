"""MI355X-native pressure-field contact hot path.  Import through ``pfc_pkg.load()``.

geometry   host-side eMesh primitives and OBB-tree construction (build time)
scenario   host-side mirror of the reference's MechanismScenario contact API, backed by libpfc_hip (C ABI)
configs    the synthetic workloads of BASELINE.md (C1..C5)
parallel   one-process-per-GPU sharding of independent items + all-gather of per-item results
_lib       ctypes binding / build of csrc/libpfc_hip.so
"""
from . import geometry, _lib, scenario, configs, parallel  # noqa: F401
from .scenario import (MechanismScenario, ContactProperties, Regularized, Bristle, ContactInstructions,  # noqa: F401
                       MeshCache, relative_pose, relative_twist)

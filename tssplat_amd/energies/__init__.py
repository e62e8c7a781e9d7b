from .smooth_barrier import SmoothnessBarrierEnergy, SmoothnessBarrierFunc  # noqa: F401

from .smooth_barrier import SmoothnessBarrierEnergy, SmoothnessBarrierFunc  # noqa: F401
from .graphed import GraphedSmoothnessBarrier  # noqa: F401
from .train_loop import FusedEnergyAdamLoop  # noqa: F401

from .E_tracker import EssTracker  # noqa: F401
from .pnp_tracker import PnpTracker  # noqa: F401

"""Drop-in alias: `remfx.models` -> `remfx_amd.models` so the reference's Hydra `_target_` strings
(cfg/model/*.yaml) and `from remfx.models import ...` lines resolve to the MI355X build."""
from remfx_amd.models import *  # noqa: F401,F403
from remfx_amd import models as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("__")})

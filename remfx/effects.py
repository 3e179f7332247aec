"""Drop-in alias: `remfx.effects` -> `remfx_amd.effects` (the five RandomPedalboard* class names and the
Pedalboard_Effects label order, reference remfx/effects.py:297-616, 699-707) so that cfg/effects/all.yaml's
`_target_` strings and `from remfx.effects import ...` lines resolve to the MI355X build."""
from remfx_amd.effects import *  # noqa: F401,F403
from remfx_amd import effects as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("__")})

"""Effect descriptors of the data side (reference remfx/effects.py:297-616, 699-707).

The removal path needs two things from this module: the five class NAMES (they are the
dict keys of RemFXChainInference.model and of cfg ``ckpts:`` / ``inference_effects_ordering``,
models.py:81,96; cfg/exp/remfx_detect.yaml:63-85) and the label order ``Pedalboard_Effects``
(effects.py:699-707) that defines column k of the (B, 5) label tensors.  The reference renders
the effects with pedalboard (JUCE C++) on the CPU while it builds the dataset; that rendering is
SURVEY 8(f) rank 3, not the hot path.  The classes here take the reference's constructor
arguments (so ``cfg/effects/all.yaml`` instantiates unchanged), keep the parameter ranges, draw
parameters the same way (uniform in [min, max]) and raise when asked to render.
"""
import torch


class _RandomEffect(torch.nn.Module):
    """Keeps every ``min_* / max_*`` range of the reference constructor; ``draw()`` samples them."""

    defaults = {}

    def __init__(self, sample_rate: float, **ranges):
        super().__init__()
        unknown = set(ranges) - set(self.defaults)
        if unknown:                                   # same failure mode as a wrong kwarg upstream
            raise TypeError(f"{type(self).__name__}.__init__() got unexpected keyword arguments {sorted(unknown)}")
        self.sample_rate = sample_rate
        self.ranges = dict(self.defaults)
        self.ranges.update(ranges)
        for k, v in self.ranges.items():
            setattr(self, k, v)

    def draw(self, generator=None):
        """One parameter set: uniform between each (min_x, max_x) pair (effects.py:323-334 and siblings)."""
        out = {}
        for k, lo in self.ranges.items():
            if not k.startswith("min_"):
                continue
            name = k[4:]
            hi = self.ranges.get("max_" + name, self.ranges.get("max_" + name.replace("seconds", "sconds"), lo))
            u = float(torch.rand((), generator=generator))
            out[name] = lo + (hi - lo) * u
        return out

    def forward(self, x: torch.Tensor):
        raise NotImplementedError(
            f"{type(self).__name__}: rendering audio effects (pedalboard / JUCE on the CPU in the reference) is the "
            "dataset-generation side, SURVEY 8(f) rank 3; the removal hot path consumes rendered clips or synthetic noise")


class RandomPedalboardReverb(_RandomEffect):
    defaults = dict(min_room_size=0.0, max_room_size=1.0, min_damping=0.0, max_damping=1.0, min_wet_dry=0.0,
                    max_wet_dry=0.7, min_width=0.0, max_width=1.0)


class RandomPedalboardChorus(_RandomEffect):
    defaults = dict(min_rate_hz=0.25, max_rate_hz=4.0, min_depth=0.0, max_depth=0.6, min_centre_delay_ms=5.0,
                    max_centre_delay_ms=10.0, min_feedback=0.1, max_feedback=0.6, min_mix=0.1, max_mix=0.7)


class RandomPedalboardDelay(_RandomEffect):
    # `max_delay_sconds` is the reference's own spelling (effects.py:346; cfg/effects/all.yaml)
    defaults = dict(min_delay_seconds=0.1, max_delay_sconds=1.0, min_feedback=0.05, max_feedback=0.6, min_mix=0.0,
                    max_mix=0.7)


class RandomPedalboardDistortion(_RandomEffect):
    defaults = dict(min_drive_db=-20.0, max_drive_db=12.0)


class RandomPedalboardCompressor(_RandomEffect):
    defaults = dict(min_threshold_db=-42.0, max_threshold_db=-6.0, min_ratio=1.5, max_ratio=4.0, min_attack_ms=1.0,
                    max_attack_ms=50.0, min_release_ms=10.0, max_release_ms=250.0)


# label order: column k of dry / wet label tensors (effects.py:699-707)
Pedalboard_Effects = [
    RandomPedalboardReverb,
    RandomPedalboardChorus,
    RandomPedalboardDelay,
    RandomPedalboardDistortion,
    RandomPedalboardCompressor,
]

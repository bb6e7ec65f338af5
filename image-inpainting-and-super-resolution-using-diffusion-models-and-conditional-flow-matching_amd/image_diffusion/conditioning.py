"""Sampler selectors (what `AD/image_diffusion/conditioning.py:12-78` provides): each type only carries the hyper-parameters
of one conditional sampler; `image_diffusion.sampling` dispatches on the type.

Declarative layout: a subclass names its parameters once in `PARAMS` (in the reference's positional order); construction
from positionals / keywords and from a config mapping are derived from that list.
"""
from typing import Dict, Tuple, Type


class Conditioning:
    PARAMS: Tuple[str, ...] = ()
    KEY: str = ""

    def __init__(self, *args, **kwargs):
        names = self.PARAMS
        if len(args) > len(names):
            raise TypeError(f"{type(self).__name__} takes {len(names)} parameters ({', '.join(names)})")
        given = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in names or k in given:
                raise TypeError(f"{type(self).__name__}: unexpected or repeated parameter {k!r}")
            given[k] = v
        missing = [n for n in names if n not in given]
        if missing:
            raise TypeError(f"{type(self).__name__}: missing {', '.join(missing)}")
        for n in names:
            setattr(self, n, given[n])

    @classmethod
    def from_configdict(cls, config):
        return cls(**{n: config[n] for n in cls.PARAMS})

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{n}={getattr(self, n)!r}' for n in self.PARAMS)})"


class Amortized(Conditioning):
    """Condition concatenated on the channel axis of the network input; optional Langevin corrector (sampling.py:80-133)."""
    KEY, PARAMS = "amortized", ("p_cond", "n_corrector", "delta")


class ReconstructionGuidance(Conditioning):
    """Gradient guidance through the x0 predictor (sampling.py:136-206); needs the U-Net backward: not built yet."""
    KEY, PARAMS = "reconstruction_guidance", ("gamma", "start_fraction", "update_rule", "n_corrector", "delta")


class Replacement(Conditioning):
    """Known pixels are overwritten by the (optionally re-noised) condition each step (sampling.py:209-260)."""
    KEY, PARAMS = "replacement", ("delta", "start_fraction", "noise", "n_corrector")


_REGISTRY: Dict[str, Type[Conditioning]] = {c.KEY: c for c in (Amortized, ReconstructionGuidance, Replacement)}


def get_conditioning(type_: str) -> Type[Conditioning]:
    try:
        return _REGISTRY[type_.lower()]
    except KeyError:
        raise NotImplementedError(f"Unknown conditioning {type_}") from None

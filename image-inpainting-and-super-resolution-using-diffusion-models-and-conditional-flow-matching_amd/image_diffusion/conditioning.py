"""Counterpart of `AD/image_diffusion/conditioning.py:12-78`: parameter holders that select a sampler."""
from typing import Type


class Conditioning:
    @classmethod
    def from_configdict(cls, config):
        return cls()


class Amortized(Conditioning):
    def __init__(self, p_cond: float, n_corrector: int, delta: float):
        self.p_cond, self.n_corrector, self.delta = p_cond, n_corrector, delta

    @classmethod
    def from_configdict(cls, config):
        return cls(p_cond=config["p_cond"], n_corrector=config["n_corrector"], delta=config["delta"])


class ReconstructionGuidance(Conditioning):
    def __init__(self, gamma: float, start_fraction: float, update_rule: str, n_corrector: int, delta: float) -> None:
        self.gamma, self.start_fraction, self.update_rule = gamma, start_fraction, update_rule
        self.n_corrector, self.delta = n_corrector, delta

    @classmethod
    def from_configdict(cls, config):
        return cls(gamma=config["gamma"], start_fraction=config["start_fraction"], update_rule=config["update_rule"],
                   n_corrector=config["n_corrector"], delta=config["delta"])


class Replacement(Conditioning):
    def __init__(self, delta: float, start_fraction: float, noise: bool, n_corrector: int) -> None:
        self.delta, self.start_fraction, self.noise, self.n_corrector = delta, start_fraction, noise, n_corrector

    @classmethod
    def from_configdict(cls, config):
        return cls(delta=config["delta"], start_fraction=config["start_fraction"], noise=config["noise"],
                   n_corrector=config["n_corrector"])


def get_conditioning(type_: str) -> Type[Conditioning]:
    t = type_.lower()
    if t == "amortized":
        return Amortized
    if t == "reconstruction_guidance":
        return ReconstructionGuidance
    if t == "replacement":
        return Replacement
    raise NotImplementedError(f"Unknown conditioning {type_}")

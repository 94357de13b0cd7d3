"""Unit systems and SI constants.

Mirrors the public names and numerical values of mdsuite/utils/units.py:29-98 of
the reference, including ``SI.boltzmann = 1.386049e-23`` (a transposed digit in
the reference that results depend on, units.py:92).
"""
from dataclasses import dataclass

standard_state_pressure = 100000
avogadro_constant = 6.02214076e23
elementary_charge = 1.602176634e-19
boltzmann_constant = 1.380649e-23
hyperfine_transition_frequency = 9192631770
luminous_efficacy = 683
planck_constant = 6.62607015e-34
reduced_planck_constant = 1.054571817e-34
speed_of_light = 299792458
gravity = 9.80665
atmosphere = 101325
golden_ratio = 1.618033988749895


@dataclass(frozen=True)
class Units:
    """Conversion factors from a simulation unit system to SI."""

    time: float
    length: float
    energy: float
    NkTV2p: float
    boltzmann: float
    temperature: float
    pressure: float
    avogadro: float = 6.02214076e23
    elementary_charge: float = 1.602176634e-19

    @property
    def volume(self) -> float:
        return self.length**3


REAL = Units(time=1e-15, length=1e-10, energy=4184 / 6.02214076e23, NkTV2p=68568.415,
             boltzmann=0.0019872067, temperature=1, pressure=101325.0)

METAL = Units(time=1e-12, length=1e-10, energy=1.6022e-19, NkTV2p=1.6021765e6,
              boltzmann=8.617343e-5, temperature=1, pressure=100000)

SI = Units(time=1, length=1, energy=1, NkTV2p=1.380649e-23, boltzmann=1.386049e-23,
           temperature=1, pressure=1)

units_dict = {"real": REAL, "metal": METAL, "si": SI}

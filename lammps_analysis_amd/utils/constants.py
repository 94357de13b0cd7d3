"""Dataset-name constants (mdsuite/utils/constants.py:35-38 of the reference)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class DatasetKeys:
    OBSERVABLES = "Observables"

"""Exceptions with the names the reference uses (mdsuite/utils/exceptions.py)."""


class CannotPerformThisAnalysis(Exception):
    """Raised when the data does not allow the requested analysis, e.g. too few peaks
    in a radial distribution function."""

from .experiment import Experiment  # noqa: F401

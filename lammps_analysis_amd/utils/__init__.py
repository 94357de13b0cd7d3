from . import config, constants, helpers, meta_functions, scale_functions, units  # noqa: F401

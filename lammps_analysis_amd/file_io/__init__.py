from . import file_read, script_input  # noqa: F401

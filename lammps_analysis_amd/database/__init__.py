from . import (calculator_database, data_manager, mdsuite_properties, scheme,  # noqa: F401
               simulation_database)

"""MI355X-native trajectory-analysis engine behind the MDSuite calculator API.

``import lammps_analysis_amd as mds`` gives the names user scripts take from
``mdsuite``: ``mds.Project``, ``mds.Experiment``, ``mds.units``, ``mds.config`` and
the sub-packages ``database``, ``file_io``, ``calculators``, ``transformations``,
``utils``.  All numerics run in hand-written HIP kernels (``libmdx.so``) reached
through a ctypes C-ABI; nothing here imports PyTorch, TensorFlow or JAX.
"""
import logging
import sys

from .utils import config as _config_module
from .utils import units  # noqa: F401
from .utils.config import config  # noqa: F401
from . import utils, database, file_io  # noqa: F401,E402
from .experiment import Experiment  # noqa: F401,E402
from .project import Project  # noqa: F401,E402

__version__ = "0.1.0"
__all__ = ["Project", "Experiment", "units", "config"]

logger = logging.getLogger("mdsuite")
logger.setLevel(logging.DEBUG)
if not logger.handlers:
    _channel = logging.StreamHandler(sys.stdout)
    _channel.setLevel(logging.INFO)
    _channel.setFormatter(logging.Formatter("%(asctime)s %(levelname)s: %(message)s"))
    logger.addHandler(_channel)

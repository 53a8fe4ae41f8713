from vtc_amd.host.model import *  # noqa: F401,F403
from vtc_amd.host.model import __all__  # noqa: F401

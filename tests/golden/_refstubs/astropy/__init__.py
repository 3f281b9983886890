"""astropy stand-in (units only) for the golden-vector harness."""
from . import units  # noqa: F401

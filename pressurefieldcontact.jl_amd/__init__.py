"""MI355X-native pressure-field contact hot path.  Import through ``pfc_pkg.load()``."""
from . import geometry  # noqa: F401

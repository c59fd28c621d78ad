"""Import alias: the package directory is `permafrost-engine_amd/` (not a valid Python
identifier); this module makes it importable as `permafrost_engine_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "permafrost-engine_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
del _os, _f

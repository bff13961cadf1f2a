"""Import alias for the package directory ``ct-vae_amd/``.

The product lives in ``ct-vae_amd/`` (the name the build contract asks for); a hyphen is not a
valid Python identifier, so this two-line shim makes the same directory importable as
``ctvae_amd`` by pointing the package search path at it and running its ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ct-vae_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

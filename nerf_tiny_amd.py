"""Import shim: the package directory is ``nerf-tiny_amd/`` (not a valid Python identifier), so
``import nerf_tiny_amd`` resolves here and forwards to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "nerf-tiny_amd")]
__package__ = __name__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))

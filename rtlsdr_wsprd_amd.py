"""Import shim: the package directory is named `rtlsdr-wsprd_amd` (not a valid Python
identifier), so `import rtlsdr_wsprd_amd` is routed to it here."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rtlsdr-wsprd_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))

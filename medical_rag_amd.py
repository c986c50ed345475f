"""Import shim: makes the package directory `medical-rag_amd/` importable as
`medical_rag_amd` (a hyphen cannot appear in a module name)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "medical-rag_amd")]
__package__ = __name__
with open(_os.path.join(__path__[0], "__init__.py"), encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os

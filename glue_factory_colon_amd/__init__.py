"""Import alias: the product package lives in `glue-factory-colon_amd/` (a name
Python cannot import directly because of the hyphens).  This module makes it
importable as `glue_factory_colon_amd` by pointing `__path__` at that directory
and executing its `__init__.py` in this namespace.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "glue-factory-colon_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

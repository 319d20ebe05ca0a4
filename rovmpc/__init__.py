"""Import alias: ``import rovmpc`` loads the package that lives in
``catenary-model-estimation-and-mpc-control-for-rov-tethered-systems_amd/`` (a directory name
Python cannot import directly because of the hyphens)."""
import os as _os

_PKG = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "catenary-model-estimation-and-mpc-control-for-rov-tethered-systems_amd")
__path__.insert(0, _PKG)          # submodules (rovmpc.engine, ...) resolve inside the real package
with open(_os.path.join(_PKG, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG, "__init__.py"), "exec"))
del _f

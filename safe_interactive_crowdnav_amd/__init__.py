"""Importable alias for the product package.

The product lives in ``safe-interactive-crowdnav_amd/`` (the repo-mandated directory
name, which is not a valid Python identifier).  This alias package points its
``__path__`` at that directory so that

    from safe_interactive_crowdnav_amd.forecaster import HumanTrajectoryForecasterSim

resolves to ``safe-interactive-crowdnav_amd/forecaster.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "safe-interactive-crowdnav_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f

"""jmid-mi355x: MI355X-native JMID / iMID trajectory-diffusion predictor.

Drop-in for ``sicnav_diffusion.JMID.mid_sim_wrapper`` of
sepsamavi/safe-interactive-crowdnav (reference ``sicnav_diffusion/JMID/mid_sim_wrapper.py:207``):
a Python host layer that keeps the predictor's call surface
(``HumanTrajectoryForecasterSim``) over a C-ABI shared library
(``csrc/libjmid_hip.so``, declared in ``include/jmid_hip.h``) of hand-written
HIP kernels for gfx950.  Import as ``safe_interactive_crowdnav_amd``.
"""
__version__ = "0.1.0"


REFERENCE_MODULE = "sicnav_diffusion.JMID.mid_sim_wrapper"
_STAND_INS = []          # names of the empty parent packages install() registered (uninstall() removes them again)


def install(**defaults):
    """Make ``sicnav_diffusion.JMID.mid_sim_wrapper`` resolve to this package's forecaster module, so that the reference's
    caller (``sicnav_diffusion/policy/sicnav_acados.py:24``: ``from sicnav_diffusion.JMID.mid_sim_wrapper import
    HumanTrajectoryForecasterSim``) runs byte-unchanged: call ``install()`` once before the policy module is imported.

    ``defaults`` (``precision=``, ``device_id=``, ``rng_compat=``, ``self_check=``, ``device_topk=``) become the defaults of the
    class's keyword arguments, which the reference's two-positional-argument construction never passes.  The parent packages
    are the reference's own when its tree is importable; otherwise empty stand-ins are registered so that the import
    statement still resolves (tests, or a deployment that ships the policy file alone).  Returns the installed module.
    """
    import importlib
    import sys
    import types

    from . import forecaster

    unknown = set(defaults) - set(forecaster.DEFAULTS)
    if unknown:
        raise TypeError(f"install() got unknown defaults {sorted(unknown)}; known: {sorted(forecaster.DEFAULTS)}")
    forecaster.DEFAULTS.update(defaults)
    parent = None
    parts = REFERENCE_MODULE.split(".")
    for i in range(1, len(parts)):
        name = ".".join(parts[:i])
        mod = sys.modules.get(name)
        if mod is None:
            try:
                mod = importlib.import_module(name)
            except ModuleNotFoundError as ex:
                # only "this very package is not importable" (the reference tree is not on sys.path) gets a stand-in; a reference
                # package whose own __init__ fails (a missing einops / acados dependency) must surface, not be masked
                if ex.name != name:
                    raise
                mod = types.ModuleType(name)
                mod.__path__ = []
                sys.modules[name] = mod
                _STAND_INS.append(name)
        if parent is not None and not hasattr(parent, parts[i - 1]):
            setattr(parent, parts[i - 1], mod)
        parent = mod
    sys.modules[REFERENCE_MODULE] = forecaster
    setattr(parent, parts[-1], forecaster)
    return forecaster


def uninstall():
    """Undo ``install()`` (the reference's own module is imported again on the next ``import``)."""
    import sys

    from . import forecaster

    if sys.modules.get(REFERENCE_MODULE) is forecaster:
        del sys.modules[REFERENCE_MODULE]
        parent = sys.modules.get(REFERENCE_MODULE.rsplit(".", 1)[0])
        if parent is not None and getattr(parent, "mid_sim_wrapper", None) is forecaster:
            delattr(parent, "mid_sim_wrapper")
    while _STAND_INS:                  # the empty parents of a tree-less install: the real packages must be importable afterwards
        name = _STAND_INS.pop()
        mod = sys.modules.get(name)
        if mod is not None and getattr(mod, "__path__", None) == []:
            del sys.modules[name]
            head, _, leaf = name.rpartition(".")
            par = sys.modules.get(head) if head else None
            if par is not None and getattr(par, leaf, None) is mod:
                delattr(par, leaf)

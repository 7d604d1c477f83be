"""jmid-mi355x: MI355X-native JMID / iMID trajectory-diffusion predictor.

Drop-in for ``sicnav_diffusion.JMID.mid_sim_wrapper`` of
sepsamavi/safe-interactive-crowdnav (reference ``sicnav_diffusion/JMID/mid_sim_wrapper.py:207``):
a Python host layer that keeps the predictor's call surface
(``HumanTrajectoryForecasterSim``) over a C-ABI shared library
(``csrc/libjmid_hip.so``, declared in ``include/jmid_hip.h``) of hand-written
HIP kernels for gfx950.  Import as ``safe_interactive_crowdnav_amd``.
"""
__version__ = "0.1.0"

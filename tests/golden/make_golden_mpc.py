"""Golden capture of the caller-side forecast handling, generated from the REFERENCE'S OWN LINES.

`SICNavAcados.predict` / `select_action` (sicnav_diffusion/policy/sicnav_acados.py) cannot be imported here (casadi,
acados_template and rvo2 are absent), and the two passages on the predictor's side of the solver are plain
NumPy / einops:
    :1644-1680   predict_ret_best() result -> forecasts, initial sample weights, 't (h s) d' stage layout,
                 per-human goal point / preferred speed, the FullState list
    :1388-1413   per-stage Acados parameter vectors p_0 .. p_horiz (goal states / actions, cost diagonals, the MID
                 samples of stage t and t+1, optional static obstacles)
This script reads exactly those line ranges from /root/reference at run time, executes them against recording
stand-ins for `self`, `solver` and `state`, and stores inputs + everything they produced as tests/golden/mpc_glue_*.npz.
Nothing of the reference's text is written to the repo.  Run in the build container:  python tests/golden/make_golden_mpc.py
"""
import os
import sys
import textwrap
import types

import einops
import numpy as np

REF = "/root/reference/sicnav_diffusion/policy/sicnav_acados.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_lines(lo, hi):
    with open(REF) as f:
        lines = f.readlines()[lo - 1:hi]
    return textwrap.dedent("".join(lines))


class FullState:            # crowd_sim_plus/envs/utils/state_plus.py:FullState keeps exactly these fields
    def __init__(self, px, py, vx, vy, radius, gx, gy, v_pref, theta):
        self.px, self.py, self.vx, self.vy, self.radius = px, py, vx, vy, radius
        self.gx, self.gy, self.v_pref, self.theta = gx, gy, v_pref, theta
        self.position = (px, py)


class FullyObservableJointState:
    def __init__(self, self_state, human_states, static_obs):
        self.self_state, self.human_states, self.static_obs = self_state, human_states, static_obs


class Recorder:
    def __init__(self):
        self.sets = {}

    def set(self, stage, name, value):
        self.sets[(stage, name)] = np.array(value, dtype=np.float64)

    def constraints_set(self, *a):
        pass


def capture(tag, N, k, H, horiz, joint, outdoor, seed):
    rng = np.random.default_rng(seed)
    top_k_forecasts = rng.standard_normal((N, k, H + 1, 2)).cumsum(axis=2)
    lw = np.log(rng.dirichlet(np.ones(k)))
    top_k_weights = np.repeat(lw[None], N, axis=0) if joint else np.log(rng.dirichlet(np.ones(k), size=N))
    hum = [FullState(*rng.standard_normal(4), 0.3, 0.0, 0.0, 1.0, 0.0) for _ in range(N)]
    hum[0].vx = hum[0].vy = 0.0            # exercises the theta = 0 branch
    nx, nu, n_stat = 4 + 4 * N, 2, 3
    static_obs = rng.standard_normal((n_stat, 4)).tolist()
    state = types.SimpleNamespace(self_state=FullState(0.0, 0.0, 0.0, 0.0, 0.3, 4.0, 0.0, 1.0, 0.0), human_states=hum,
                                  static_obs=static_obs)

    class Forecaster:
        def update_state_hists(self, *a):
            pass

        def predict_ret_best(self):
            return top_k_forecasts, top_k_weights

    mpc_env = types.SimpleNamespace(human_pred_MID=True, Q_diag=rng.uniform(0.1, 2.0, nx), R_diag=rng.uniform(0.1, 2.0, nu),
                                    term_Q_diag=rng.uniform(0.1, 2.0, nx), num_stat_obs=n_stat, nx_r=4)
    self = types.SimpleNamespace(human_goal_cvmm=False, human_pred_MID=True, human_pred_MID_joint=joint, horiz=horiz,
                                 hum_traj_forecaster=Forecaster(), env=types.SimpleNamespace(global_time=1.25),
                                 all_forecasts=[], time_step=0.25, human_max_vel=1.0, human_goal_cvmm_horizon=3.0,
                                 mpc_env=mpc_env, outdoor_robot_setting=outdoor)
    ns = dict(self=self, state=state, np=np, einops=einops, FullState=FullState,
              FullyObservableJointState=FullyObservableJointState)
    exec(ref_lines(1639, 1682), ns)                 # `if self.human_goal_cvmm or self.human_pred_MID: ... else: ...`
    forecasts_reshaped = ns["forecasts_reshaped"]
    joint_state = ns["joint_state"]
    # second passage: the per-stage parameter vectors
    solver = Recorder()
    goal_states = rng.standard_normal((nx, horiz + 1))
    goal_actions = rng.standard_normal((nu, horiz))
    x_guess = rng.standard_normal((nx, horiz + 1))
    u_guess = rng.standard_normal((nu, horiz))
    ns2 = dict(self=self, np=np, solver=solver, MID_samples=forecasts_reshaped, goal_states=goal_states,
               goal_actions=goal_actions, x_guess=x_guess, u_guess=u_guess, joint_state=joint_state)
    if outdoor:
        exec(ref_lines(1382, 1383), ns2)            # properly_shaped_static_obs
    exec(ref_lines(1388, 1413), ns2)
    p = np.stack([solver.sets[(i, "p")] for i in range(horiz + 1)])
    hs = joint_state.human_states
    np.savez(os.path.join(OUT, f"mpc_glue_{tag}.npz"), N=N, k=k, H=H, horiz=horiz, joint=joint, outdoor=outdoor,
             time_step=0.25, top_k_forecasts=top_k_forecasts, top_k_weights=top_k_weights,
             human_pxpyvxvy=np.array([[h.px, h.py, h.vx, h.vy] for h in hum]),
             static_obs=np.array(static_obs), goal_states=goal_states, goal_actions=goal_actions,
             Q_diag=mpc_env.Q_diag, R_diag=mpc_env.R_diag, term_Q_diag=mpc_env.term_Q_diag,
             forecasts=ns["forecasts"], forecasts_init_weights=ns["forecasts_init_weights"],
             forecasts_reshaped=forecasts_reshaped,
             gx=np.array([h.gx for h in hs]), gy=np.array([h.gy for h in hs]), v_pref=np.array([h.v_pref for h in hs]),
             theta=np.array([h.theta for h in hs]), p_stages=p)
    print(tag, "p_stages", p.shape)


if __name__ == "__main__":
    capture("joint", N=3, k=15, H=8, horiz=5, joint=True, outdoor=False, seed=1)
    capture("indep_outdoor", N=5, k=20, H=12, horiz=8, joint=False, outdoor=True, seed=2)
    capture("exact_horizon", N=2, k=4, H=7, horiz=6, joint=True, outdoor=False, seed=3)   # H == horiz + 1 (fewer: IndexError there)

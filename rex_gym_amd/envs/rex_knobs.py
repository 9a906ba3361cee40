"""`env.rex` for the batched env: the slice of the reference's `Rex` object (model/rex.py:643-692) that an EnvRandomizer
reaches for from its `randomize_env(env)` hook (rex_gym_env.py:345-346) -- the URDF masses and their setters -- mapped
onto the per-env body parameters of the HIP simulator (include/rexsim.h `rex_set_body_params`).

The simulator keeps one mass scale for the base group (base_link + the two chassis links, merged into one body) and one
for the leg links, so a setter accepts any list whose entries are ONE common multiple of the URDF values (what a
randomizer that draws one ratio per group produces) and raises ValueError for independent per-link draws.  As in the
reference (`changeDynamics(mass=...)`) the inertia tensors keep their load-time values.
"""
import numpy as np

# rex.urdf: base_link 1.20 + chassis_front 0.05 + chassis_rear 0.05 (the links matched by _CHASSIS_NAME_PATTERN, rex.py:18,217)
_BASE_MASSES = (1.20, 0.05, 0.05)
# per leg, joint order of the URDF: motor_*_shoulder 0.10, motor_*_leg 0.5, *_leg_cover_joint 0.1, foot_motor_* 0.1, *_toe 0.005;
# Rex._leg_masses_urdf lists the non-motor links first (cover, foot, toe per leg), then the motor links (rex.py:173-179,226-227)
_LEG_LINK_MASSES = (0.1, 0.1, 0.005) * 4
_MOTOR_LINK_MASSES = (0.10, 0.5) * 4


class RexKnobs:
    def __init__(self, env):
        self._env = env

    def GetBaseMassesFromURDF(self):
        return list(_BASE_MASSES)

    def GetLegMassesFromURDF(self):
        return list(_LEG_LINK_MASSES + _MOTOR_LINK_MASSES)

    def _scale(self, masses, urdf, what):
        m = np.asarray(masses, dtype=np.float64)
        if m.shape[-1] != len(urdf):       # the reference's own check (rex.py:668-670, 685-687)
            raise ValueError(f"The length of {what} {m.shape[-1]} and the number of its links {len(urdf)} are not the same.")
        ratio = m / np.asarray(urdf)
        scale = ratio.mean(axis=-1)
        if not np.allclose(ratio, scale[..., None], rtol=1e-3, atol=0.0):
            raise ValueError(f"{what}: the simulator scales the links of a group by one factor per env; per-link ratios "
                             f"{np.round(ratio, 4).tolist()} differ")
        return scale

    def _apply(self, **kw):
        env = self._env
        idx = env._randomize_indices
        params = env.set_body_params()
        row = {"base_mass_scale": 0, "leg_mass_scale": 1, "foot_friction": 2}
        for name, v in kw.items():
            v = env._torch.as_tensor(np.asarray(v, dtype=np.float32), device=env.device)
            if idx is None:
                params[row[name]] = v
            else:
                params[row[name], idx.long()] = v

    def SetBaseMasses(self, base_mass):
        """base_mass: the 3 chassis-link masses (or [n, 3] for the n envs being reset)."""
        self._apply(base_mass_scale=self._scale(base_mass, _BASE_MASSES, "base_mass"))

    def SetLegMasses(self, leg_masses):
        """leg_masses: 12 leg-link + 8 motor-link masses (or [n, 20])."""
        self._apply(leg_mass_scale=self._scale(leg_masses, _LEG_LINK_MASSES + _MOTOR_LINK_MASSES, "leg_masses"))

    def SetFootFriction(self, foot_friction):
        """lateral friction of the four toes (a float or [n]); the URDF value is 0.5."""
        self._apply(foot_friction=foot_friction)

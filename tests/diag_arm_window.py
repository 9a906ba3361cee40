"""Worker of tests/test_gpu_parity.py::test_mark_arm_window_with_quiet_arm_rows.  Runs in a subprocess with
    REX_LIB_PATH = rex_gym_amd/librexsim_hip_diag.so   (the mark-arm kernels built with -DREX_DIAG_ARM_REST_INSIDE=1)
    REX_ORACLE_DIAG = 1                                (orclib loads the oracle twin built with the same define)
the 200-step parity window of BASELINE configs[4]'s shard (2 048 mark-arm envs, tasks drawn per env from walk / gallop / turn-IK, mass
and friction drawn per reset) in which the three arm joints the reference commands 0.1 rad beyond their bounds are started and commanded
0.3 rad INSIDE them (0.05 rad, recorded in profiles/r06_parity_mixed_arm_arm_rest_0.05_inside.json, still lets 15 % of the envs swing a joint onto a bound): the arm's limit rows stay quiet, and what is left is the arithmetic of the two float paths.
usage: diag_arm_window.py OUT.json [envs-per-wave]"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)


def main():
    out_path = sys.argv[1]
    assert os.environ.get("REX_ORACLE_DIAG") == "1" and os.environ.get("REX_LIB_PATH", "").endswith("librexsim_hip_diag.so")
    import parity_window as pw
    name = "mixed_arm_2048"
    env = pw.make_env(name)
    threads = len(os.sched_getaffinity(0))
    rec = pw.window(name, env, steps=200, seed=23, threads=min(threads, 32))
    floor = pw.float32_floor(name, env, steps=200, seed=23, threads=min(threads, 32))
    rec.pop("abs_error_by_step", None)
    rec["float32_floor"] = floor
    rec["envs_per_wave"] = env._L.rex_envs_per_wave(env._h)
    rec["diagnostic"] = ("arm rest targets (-1.2, -1.2, 0, 0, 1.2, 0): 0.3 rad inside the +-1.5 rad bounds of m1, m2, m5 (the reference's "
                         "ARM_POSES['rest'] puts them 0.1 rad beyond); HIP library and oracle both built with -DREX_DIAG_ARM_REST_INSIDE")
    env.close()
    with open(out_path, "w") as f:
        json.dump(rec, f)


if __name__ == "__main__":
    main()

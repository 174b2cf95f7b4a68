"""MJCF compiler: structural counts and known-answer constants (SURVEY D), shipped JSON == fresh compile."""
import os

import numpy as np
import pytest

from dial_mpc_amd import mjcf
from dial_mpc_amd.envs.base_env import load_model

REF = "/root/reference/dial_mpc/models"


def test_go2_counts_and_constants():
    m = load_model("unitree_go2", "mjx_scene_force.xml")
    assert (m["nq"], m["nv"], m["nu"], m["nbody"], m["njnt"]) == (19, 18, 12, 14, 13)
    assert (m["ncon"], m["nlim"], m["nefc"], m["iterations"], m["ls_iterations"], m["eulerdamp"]) == (4, 12, 28, 2, 5, 0)
    assert abs(np.sum(m["body_mass"]) - 16.206408) < 1e-6
    assert abs(m["meaninertia"] - 2.803496) < 1e-6
    diw = m["dof_invweight0"]
    assert np.allclose(diw[:6], [0.066313] * 3 + [5.711471] * 3, atol=1e-6)
    assert np.allclose(diw[6:9], [28.590520, 28.529458, 79.886225], atol=1e-5)
    calf = [m["names"]["body"].index(n) for n in ("FR_calf", "FL_calf", "RR_calf", "RL_calf")]
    assert np.allclose(m["body_invweight0"][calf, 0], [1.764529, 1.764556, 1.771599, 1.771576], atol=1e-6)
    assert np.allclose(m["con_friction"][0], [1, 1, 0.02, 0.01, 0.01])
    assert np.allclose(m["con_solimp"][0], [0.4575, 0.975, 0.016, 0.5, 2.0])
    assert np.allclose(m["con_margin"], 0.001)
    assert np.all(np.isinf(m["act_ctrlrange"]))            # Brax rewrites unlimited ranges to +-inf
    assert m["names"]["geom"] == ["floor", "FR", "FL", "RR", "RL"]


def test_go2_home_keyframe_quantities():
    m = load_model("unitree_go2", "mjx_scene_force.xml")
    q = np.array(m["keyframes"]["home"])
    kin = mjcf.host_kinematics(m, q)
    M = mjcf.host_mass_matrix(m, kin)
    assert np.allclose(kin["subtree_com"][1], [-0.000733, 0, 0.249583], atol=1e-6)
    assert np.allclose(np.diag(M)[:9], [16.206408] * 3 + [0.177287, 0.491251, 0.535889, 0.033572, 0.030390, 0.016326], atol=1e-6)
    fr = m["names"]["site"].index("FR_foot")
    b = m["site_bodyid"][fr]
    p = kin["xpos"][b] + kin["xmat"][b] @ m["site_pos"][fr]
    assert np.allclose(p, [0.192157, -0.142, 0.003627], atol=1e-6)  # 13.87 mm inside the floor (r = 0.0175)


def test_h1_counts():
    m = load_model("unitree_h1", "mjx_scene_h1_walk.xml")
    assert (m["nq"], m["nv"], m["nu"], m["nbody"]) == (26, 25, 19, 21)
    assert (m["ncon"], m["nlim"], m["nefc"]) == (4, 19, 35)
    assert list(m["con_kind"]) == [1, 2, 1, 2]               # two capsule-end contacts per foot, pair-major
    assert m["names"]["body"].index("torso_link") == 12 and m["names"]["body"].index("pelvis") == 1
    assert np.allclose(m["act_ctrlrange"][3], [-300, 300]) and np.allclose(m["act_ctrlrange"][4], [-40, 40])
    assert np.allclose(m["con_solimp"][0], [0.9, 0.95, 0.001, 0.5, 2.0]) and np.allclose(m["con_margin"], 0)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
@pytest.mark.parametrize("robot,xml", [("unitree_go2", "mjx_scene_force.xml"), ("unitree_h1", "mjx_scene_h1_walk.xml"),
                                       ("unitree_go2", "mjx_scene_force_crate.xml"), ("unitree_h1", "mjx_scene_h1_push_crate.xml")])
def test_shipped_json_matches_fresh_compile(robot, xml):
    fresh = mjcf.compile_mjcf(os.path.join(REF, robot, xml))
    shipped = load_model(robot, xml)
    for k, v in fresh.items():
        if isinstance(v, np.ndarray):
            if v.size == 0:                      # (an empty table loses its trailing dimensions in JSON)
                assert np.asarray(shipped[k]).size == 0, k
                continue
            assert np.allclose(v, np.asarray(shipped[k], dtype=v.dtype), rtol=0, atol=0, equal_nan=True), k


def _write_stl(path, tris):
    import struct
    with open(path, "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", len(tris)))
        for t in tris:
            f.write(struct.pack("<12fH", 0, 0, 0, *np.asarray(t, np.float32).ravel(), 0))


def test_mesh_mass_properties_of_a_box(tmp_path):
    """Inertia-from-geom for mesh geoms (H1 loco arm links): hull and exact integrals of an offset box."""
    lo, hi = np.array([0.1, -0.2, 0.3]), np.array([0.5, 0.4, 0.6])
    c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = [c[[q[0], q[1], q[2]]] for q in quads] + [c[[q[0], q[2], q[3]]] for q in quads]
    p = str(tmp_path / "box.stl")
    _write_stl(p, tris)
    ext = hi - lo
    I_ref = np.prod(ext) / 12.0 * np.diag([ext[1] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[1] ** 2])
    for mode in ("convex", "exact"):
        vol, com, I = mjcf._mesh_mass_props(p, np.ones(3), mode)
        assert abs(vol - np.prod(ext)) < 1e-7
        np.testing.assert_allclose(com, (lo + hi) / 2, atol=1e-6)
        np.testing.assert_allclose(I, I_ref, atol=1e-7)


def test_h1_loco_model_dimensions_and_welded_arms():
    from dial_mpc_amd.envs.base_env import load_model
    m = load_model("unitree_h1", "mjx_scene_h1_loco.xml")
    assert (m["nq"], m["nv"], m["nu"], m["nbody"], m["ncon"], m["nlim"], m["nefc"]) == (18, 17, 11, 21, 8, 11, 43)
    assert (m["iterations"], m["ls_iterations"]) == (1, 1)
    assert abs(m["body_mass"][12] - 24.457) < 1e-9            # torso with the arm masses folded in
    arms = np.asarray(m["body_mass"][13:21])
    assert np.all(arms > 0.3) and np.all(arms < 2.0)           # hull-inferred masses of the welded arm links
    np.testing.assert_allclose(arms[:4], arms[4:], rtol=2e-3)  # left/right symmetry of the meshes
    assert list(m["dof_damping"][6:]) == [2.0] * 11

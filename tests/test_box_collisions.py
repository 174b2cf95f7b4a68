"""Box narrow phases of the crate scenes, pinned to GEOMETRY (CPU).

MJX routes boxes through ``collision_convex.py`` -- third-party code that is not under /root/reference and whose manifold
bookkeeping changed between releases.  What can be pinned without it is what every implementation must agree on: the
distance between the two shapes, a normal along which they separate, a contact point between the two surfaces.  These
tests check the oracle's restatement (oracle/dial_oracle.c: sphere_box, plane_box, capsule_box, box_box) against brute
force -- SciPy minimisers over the shapes' parametrisations and dense direction sampling of the support functions -- and
never against the HIP code (tests/test_wave_emu.py and tests/test_gpu_parity.py compare that with the oracle).
"""
import numpy as np
import pytest
from scipy.optimize import minimize

import oracle as O
from conftest import setup_case

KIND_PLANE_BOX, KIND_SPHERE_BOX, KIND_CAPSULE_BOX, KIND_BOX_BOX = 5, 6, 7, 8


@pytest.fixture(scope="module")
def orc():
    _dc, _env, model, task, cfg = setup_case("unitree_go2_trot", 8, 4)
    return O.Oracle(model, task, cfg, np.float64)


def rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def box_closest(c, R, h, p):
    """closest point of the box to p (world) and whether p is inside"""
    l = R.T @ (p - c)
    q = np.clip(l, -h, h)
    return c + R @ q, bool(np.all(np.abs(l) <= h))


def assert_frame(frame):
    assert np.allclose(frame @ frame.T, np.eye(3), atol=1e-9)
    assert np.linalg.det(frame) > 0.99


def test_sphere_box_is_the_closest_point_of_the_box(orc):
    rng = np.random.default_rng(0)
    for _ in range(300):
        c, R, h = rng.normal(size=3), rand_rot(rng), rng.uniform(0.05, 0.6, 3)
        s, r = c + R @ (rng.normal(size=3) * h * 1.5), rng.uniform(0.01, 0.2)
        q, inside = box_closest(c, R, h, s)
        dist, pos, frame = orc.box_contact(KIND_SPHERE_BOX, 0, (s, np.eye(3), [r, 0, 0]), (c, R, h))
        assert_frame(frame)
        n = frame[0]
        if not inside:
            # brute force: the constrained minimiser over the box's own coordinates agrees with the clamp
            res = minimize(lambda x: np.sum((c + R @ x - s) ** 2), np.zeros(3), bounds=[(-a, a) for a in h], tol=1e-14)
            assert np.isclose(np.sqrt(res.fun), np.linalg.norm(q - s), atol=1e-6)
            assert np.isclose(dist, np.linalg.norm(q - s) - r, atol=1e-12)
            assert np.allclose(n, (q - s) / np.linalg.norm(q - s), atol=1e-9)       # from the sphere into the box
            assert np.allclose(pos, 0.5 * ((s + n * r) + q), atol=1e-12)            # midway between the two surfaces
        else:
            # the sphere's centre is inside: moving it by -(dist + r) n ... i.e. by `depth` against n puts it ON the surface
            depth = -(dist + r)
            assert depth >= 0
            l = R.T @ (s - n * depth - c)
            assert np.isclose(np.max(np.abs(l) - h), 0.0, atol=1e-9)
            assert np.isclose(depth, np.min(h - np.abs(R.T @ (s - c))), atol=1e-12)  # ... through the NEAREST face


def test_plane_box_candidates_are_the_lowest_vertices(orc):
    rng = np.random.default_rng(1)
    for _ in range(100):
        c, R, h = rng.normal(size=3), rand_rot(rng), rng.uniform(0.05, 0.6, 3)
        Rp, pp = rand_rot(rng), rng.normal(size=3) * 0.3
        n = Rp[:, 2]
        verts = np.array([c + R @ (np.array([(i & 1) * 2 - 1, ((i >> 1) & 1) * 2 - 1, ((i >> 2) & 1) * 2 - 1]) * h) for i in range(8)])
        heights = np.sort((verts - pp) @ n)
        for sub in range(4):
            dist, pos, frame = orc.box_contact(KIND_PLANE_BOX, sub, (pp, Rp, [0, 0, 0.05]), (c, R, h))
            assert_frame(frame)
            assert np.allclose(frame[0], n, atol=1e-12)
            assert np.isclose(dist, heights[sub], atol=1e-12)
            # pos is halfway between a vertex at that height and the plane
            v = pos + n * dist * 0.5
            assert np.min(np.linalg.norm(verts - v, axis=1)) < 1e-9
            assert np.isclose((pos - pp) @ n, dist * 0.5, atol=1e-12)


def _segment_box_distance(c, R, h, e0, e1):
    best = np.inf
    for t0 in np.linspace(0, 1, 41):
        res = minimize(lambda t: np.sum((box_closest(c, R, h, e0 + np.clip(t[0], 0, 1) * (e1 - e0))[0] - (e0 + np.clip(t[0], 0, 1) * (e1 - e0))) ** 2),
                       [t0], bounds=[(0, 1)], tol=1e-14)
        best = min(best, res.fun)
    return np.sqrt(best)


def test_capsule_box_first_candidate_is_the_closest_point_of_the_segment(orc):
    rng = np.random.default_rng(2)
    n_inside = 0
    for _ in range(120):
        c, R, h = rng.normal(size=3), rand_rot(rng), rng.uniform(0.05, 0.6, 3)
        Rc, hl, r = rand_rot(rng), rng.uniform(0.03, 0.4), rng.uniform(0.01, 0.05)
        ctr = c + R @ (rng.normal(size=3) * h * 1.3)
        e0, e1 = ctr - Rc[:, 2] * hl, ctr + Rc[:, 2] * hl
        # does the segment pass through the box?  (then the distance is 0 and the candidate is an "inside" sphere)
        ts = np.linspace(0, 1, 2001)
        pts = e0 + ts[:, None] * (e1 - e0)
        through = np.any(np.all(np.abs((pts - c) @ R) <= h, axis=1))
        d0, p0, f0 = orc.box_contact(KIND_CAPSULE_BOX, 0, (ctr, Rc, [r, hl, 0]), (c, R, h))
        d1, p1, f1 = orc.box_contact(KIND_CAPSULE_BOX, 1, (ctr, Rc, [r, hl, 0]), (c, R, h))
        assert_frame(f0)
        assert_frame(f1)
        if through:
            # the capsule's axis enters the box: the contact is where the axis crosses the surface (unless it is inside with
            # both ends), normal = the face it crosses, dist = -radius
            n_inside += 1
            inside = np.all(np.abs((pts - c) @ R) <= h, axis=1)
            if inside.all():
                continue
            assert np.isclose(d0, -r, atol=1e-12)
            surf = p0 - f0[0] * (r * 0.5)                       # pos = crossing point + n (r + dist / 2)
            loc = R.T @ (surf - c)
            assert np.max(np.abs(loc) - h) < 1e-9 and np.min(np.abs(np.abs(loc) - h)) < 1e-9       # on the surface
            k = int(np.argmin(np.abs(np.abs(loc) - h)))
            assert np.allclose(R.T @ f0[0], -np.sign(loc[k]) * np.eye(3)[k], atol=1e-9)            # into the box through that face
            first = int(np.argmax(inside)) if not inside[0] else int(len(ts) - 1 - np.argmax(inside[::-1]))
            assert np.linalg.norm(pts[first] - surf) < 2.0 * np.linalg.norm(e1 - e0) / 2000 + 1e-9  # where the axis crosses
            continue
        want = _segment_box_distance(c, R, h, e0, e1)
        assert np.isclose(d0 + r, want, atol=2e-6), (d0 + r, want)
        # second candidate: one of the two END spheres, and never closer than the first
        ends = [np.linalg.norm(box_closest(c, R, h, e)[0] - e) - r for e in (e0, e1)]
        assert min(abs(d1 - ends[0]), abs(d1 - ends[1])) < 1e-9
        assert d1 >= d0 - 1e-9
    assert 0 < n_inside < 60


def test_capsule_lying_on_a_face_touches_with_both_ends(orc):
    c, R, h = np.array([0.0, 0, 0.3]), np.eye(3), np.array([0.31, 0.46, 0.3])
    Rc = np.array([[0.0, 0, 1], [0, 1, 0], [-1, 0, 0]])       # capsule axis along world x
    ctr, hl, r = np.array([0.05, 0.1, 0.6 + 0.013 - 0.002]), 0.06, 0.013
    d0, p0, f0 = orc.box_contact(KIND_CAPSULE_BOX, 0, (ctr, Rc, [r, hl, 0]), (c, R, h))
    d1, p1, f1 = orc.box_contact(KIND_CAPSULE_BOX, 1, (ctr, Rc, [r, hl, 0]), (c, R, h))
    assert np.isclose(d0, -0.002) and np.isclose(d1, -0.002)
    assert np.allclose(f0[0], [0, 0, -1]) and np.allclose(f1[0], [0, 0, -1])
    assert np.isclose(abs(p0[0] - p1[0]), 2 * hl)               # the two ends


def _support_sep(cA, RA, hA, cB, RB, hB, n):
    return n @ (cB - cA) - np.abs(RA.T @ n) @ hA - np.abs(RB.T @ n) @ hB


def _max_separation(cA, RA, hA, cB, RB, hB, rng):
    """max over unit directions of the support-function separation: > 0 = the distance of disjoint boxes' best slab,
    < 0 = minus the penetration depth.  Dense sampling + local refinement."""
    dirs = rng.normal(size=(20000, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    vals = dirs @ (cB - cA) - np.abs(dirs @ RA) @ hA - np.abs(dirs @ RB) @ hB
    best = -np.inf
    for i in np.argsort(vals)[-12:]:
        res = minimize(lambda v: -_support_sep(cA, RA, hA, cB, RB, hB, v / np.linalg.norm(v)), dirs[i], method="Nelder-Mead",
                       options=dict(xatol=1e-10, fatol=1e-12, maxiter=4000))
        best = max(best, -res.fun)
    return best


def test_box_box_depth_normal_and_points(orc):
    rng = np.random.default_rng(3)
    n_face = n_edge = n_apart = 0
    for trial in range(120):
        cA, RA, hA = rng.normal(size=3) * 0.1, rand_rot(rng), rng.uniform(0.05, 0.4, 3)
        RB, hB = rand_rot(rng), rng.uniform(0.05, 0.4, 3)
        cB = cA + rand_rot(rng)[:, 0] * rng.uniform(0.05, 0.75)
        cands = [orc.box_contact(KIND_BOX_BOX, k, (cA, RA, hA), (cB, RB, hB)) for k in range(4)]
        d0, p0, f0 = cands[0]
        assert_frame(f0)
        n = f0[0]
        assert n @ (cB - cA) > 0                                       # from geom1 into geom2
        smax = _max_separation(cA, RA, hA, cB, RB, hB, rng)
        if smax > 0.01:
            n_apart += 1
            assert all(c[0] > 0.001 for c in cands)                    # nothing inside the 1 mm margin
            continue
        # the reported normal is a direction of (near-)least penetration and dist is the separation along it
        sep_n = _support_sep(cA, RA, hA, cB, RB, hB, n)
        assert sep_n >= smax - 0.05 * abs(smax) - 2e-5, (sep_n, smax)
        assert np.isclose(d0, sep_n, atol=1e-9) or d0 >= sep_n - 1e-9   # a clipped face point is never deeper than the slab
        live = [c for c in cands if c[0] < 0.5]
        assert 1 <= len(live) <= 4
        if len(live) == 1 and np.min(np.abs(np.abs(RA.T @ n) - 1)) > 1e-6 and np.min(np.abs(np.abs(RB.T @ n) - 1)) > 1e-6:
            n_edge += 1
        else:
            n_face += 1
        for dist, pos, frame in live:
            assert np.allclose(frame[0], n, atol=1e-12)
            # the contact point sits between the two surfaces: half the gap from each box along n
            for (c_, R_, h_, sg) in ((cA, RA, hA, +1.0), (cB, RB, hB, -1.0)):
                l = R_.T @ (pos - sg * n * dist * 0.5 - c_)
                assert np.max(np.abs(l) - h_) < 1e-6, (trial, np.max(np.abs(l) - h_))
        # candidates come deepest first
        ds = [c[0] for c in live]
        assert ds == sorted(ds)
    assert n_face > 20 and n_edge > 3 and n_apart > 10, (n_face, n_edge, n_apart)


def test_trunk_box_resting_on_the_crate_edge_gets_face_contacts(orc):
    """The climbing case: the trunk's belly (a long thin box) lies across the crate's top front edge."""
    cB, RB, hB = np.array([1.3, 0, 0.3]), np.eye(3), np.array([0.31, 0.46, 0.3])        # the crate
    pitch = -0.5
    RA = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
    hA = np.array([0.1881, 0.04675, 0.057])
    edge = np.array([0.99, 0.0, 0.6])                                                    # crate's top front edge
    cA = edge + RA @ np.array([0.0, 0.0, hA[2] - 0.003])                                 # belly 3 mm into the edge
    cands = [orc.box_contact(KIND_BOX_BOX, k, (cA, RA, hA), (cB, RB, hB)) for k in range(4)]
    live = [c for c in cands if c[0] < 0.001]
    assert len(live) >= 2                                            # a line contact: two points across the trunk's width
    ys = sorted(c[1][1] for c in live)
    assert ys[-1] - ys[0] > 1.5 * hA[1]
    for dist, pos, frame in live:
        assert -0.004 < dist < 0.001
        assert abs(pos[0] - 0.99) < 0.01 and abs(pos[2] - 0.6) < 0.01


# ---------------------------------------------------------------- the kernel's copy (csrc/box_collide.h, host emulator build)
def _rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def _q2m(q):
    from dial_mpc_amd import mjcf
    return mjcf.quat_to_mat(q)


@pytest.mark.parametrize("kind,nsub", [(KIND_PLANE_BOX, 4), (KIND_SPHERE_BOX, 1), (KIND_CAPSULE_BOX, 2), (KIND_BOX_BOX, 4)])
def test_kernel_narrow_phase_matches_the_oracle(kind, nsub):
    """Same pairs through the kernel's fp32 code and the fp32 oracle.  Both restate the same geometry in different
    code; where the geometry has a discrete choice (which vertex is k-th lowest, which axis separates, which end of a
    capsule is nearer) a pair sitting within rounding of a tie may legitimately come out differently: such pairs are
    counted and capped, everything else must agree to fp32 accuracy."""
    import emu_lib
    _dc, _env, model, task, cfg = setup_case("unitree_go2_trot", 8, 4)
    o32 = O.Oracle(model, task, cfg, np.float32)
    emu = emu_lib.Emu(model, task, cfg)
    rng = np.random.default_rng(10 + kind)
    n, ties = 400, 0
    for _ in range(n):
        c2, q2, h2 = rng.normal(size=3) * 0.2, _rand_quat(rng), rng.uniform(0.05, 0.5, 3)
        q1 = _rand_quat(rng)
        R2 = _q2m(q2)
        if kind == KIND_PLANE_BOX:
            c1, size1 = c2 - _q2m(q1)[:, 2] * rng.uniform(0.0, 0.6), np.array([0, 0, 0.05])
        elif kind == KIND_SPHERE_BOX:
            c1, size1 = c2 + R2 @ (rng.normal(size=3) * h2 * 1.2), np.array([rng.uniform(0.01, 0.1), 0, 0])
        elif kind == KIND_CAPSULE_BOX:
            c1, size1 = c2 + R2 @ (rng.normal(size=3) * h2 * 1.3), np.array([rng.uniform(0.01, 0.03), rng.uniform(0.03, 0.2), 0])
        else:
            size1 = rng.uniform(0.04, 0.3, 3)
            c1 = c2 + _q2m(_rand_quat(rng))[:, 0] * rng.uniform(0.05, 0.7)
        for sub in range(nsub):
            do, po, fo = o32.box_contact(kind, sub, (c1, _q2m(q1), size1), (c2, R2, h2))
            de, pe, fe = emu.box_contact(kind, sub, (c1, q1, size1), (c2, q2, h2))
            same = abs(do - de) < 2e-5 and np.allclose(fo[0], fe[0], atol=2e-4) and (do > 0.5 or np.allclose(po, pe, atol=2e-5))
            if kind != KIND_SPHERE_BOX and do > 0.01 and de > 0.01:
                same = True     # the kernel's broad phases (bounding spheres, then a separating face axis / slab / lowest vertex) park candidates
                                # that are > 1 cm apart: no rows either way, and nothing reads the position of a contact that does not touch
            if not same:
                ties += 1
    assert ties <= n * nsub * 0.01, (kind, ties)

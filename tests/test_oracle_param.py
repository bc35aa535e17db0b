"""Known-answer tests of the oracle's Bezier patch and surface of revolution.  The reference holds
no test or fixture for these routines and cannot be built here, so for them the oracle is "parity unpinned"
(oracle/rt_oracle.h); these cases pin it against closed forms instead: a flat patch is a plane, a straight profile is a
cylinder or a cone."""
import numpy as np

import oracle_bind as ob


def _flat_patch(z=2.0):
    return [[(3.0 * i, 3.0 * j, z) for j in range(4)] for i in range(4)]        # S(u,v) = (9u, 9v, z)


def test_flat_patch_is_a_plane():
    s = ob.OracleScene(0.0)
    m = ob.material("plastic")
    pid = s.add_bezier(_flat_patch(), (1.0, -2.0, 0.5), m, (255, 255, 255))
    rng = np.random.default_rng(1)
    for _ in range(200):
        u, v = rng.uniform(0.02, 0.98, 2)
        target = np.array([9 * u + 1.0, 9 * v - 2.0, 2.5])
        org = target + rng.uniform(-1, 1, 3) * np.array([4, 4, 0]) + np.array([0, 0, rng.uniform(2, 9)])
        pt, uv = s.object_first_inter_pt(pid, np.concatenate([org, target - org]))
        assert np.abs(pt - target).max() < 1e-7
        assert abs(uv[0] - u) < 1e-7 and abs(uv[1] - v) < 1e-7
    # outside the parameter square: no hit (0 < u', v' < 1, Object.h:1121-1122)
    pt, _ = s.object_first_inter_pt(pid, [-3.0, -2.0, 9.0, 0.0, 0.0, -1.0])
    assert (pt > 9e7).all()
    # a ray that starts on the patch does not hit it again (t' > eps)
    pt, _ = s.object_first_inter_pt(pid, [5.0, 1.0, 2.5, 0.3, 0.1, 1.0])
    assert (pt > 9e7).all()


def test_straight_profile_is_a_cylinder_and_a_cone():
    s = ob.OracleScene(0.0)
    m = ob.material("plastic")
    cyl = s.add_rotational([(1.5, 0.0), (1.5, 4.0)], (2.0, 1.0, 0.0), 10, 1.0, m, (255, 255, 255))
    cone = s.add_rotational([(2.0, 0.0), (0.5, 3.0)], (-4.0, 0.0, 1.0), 10, 1.0, m, (255, 255, 255))
    rng = np.random.default_rng(2)
    n_hit = 0
    for _ in range(300):
        th, h = rng.uniform(0, 2 * np.pi), rng.uniform(0.2, 3.8)
        target = np.array([2.0 + 1.5 * np.cos(th), 1.0 + 1.5 * np.sin(th), h])
        org = np.array([2.0, 1.0, h]) + 6.0 * np.array([np.cos(th + rng.uniform(-0.5, 0.5)), np.sin(th + rng.uniform(-0.5, 0.5)), rng.uniform(-0.4, 0.4)])
        d = (target - org) / np.linalg.norm(target - org)
        pt, uv = s.object_first_inter_pt(cyl, np.concatenate([org, d]))
        if (pt > 9e7).all():
            continue        # the seeds of t_v_Initial do not reach every ray (a property of the reference's solver)
        n_hit += 1
        # the hit lies on the cylinder, on the ray, and is the first of the two crossings
        assert abs(np.hypot(pt[0] - 2.0, pt[1] - 1.0) - 1.5) < 1e-6
        t = np.dot(pt - org, d)
        assert np.abs(org + d * t - pt).max() < 1e-6
        assert t <= np.linalg.norm(target - org) + 1e-6
        assert abs(uv[0] * 4.0 - pt[2]) < 1e-6          # u = height / 4 on this profile
    assert n_hit > 200
    n_hit = 0
    for _ in range(300):
        th, u = rng.uniform(0, 2 * np.pi), rng.uniform(0.05, 0.95)
        rad, h = 2.0 - 1.5 * u, 3.0 * u
        target = np.array([-4.0 + rad * np.cos(th), rad * np.sin(th), 1.0 + h])
        org = target + 5.0 * np.array([np.cos(th), np.sin(th), rng.uniform(0.3, 1.2)])
        d = (target - org) / np.linalg.norm(target - org)
        pt, uv = s.object_first_inter_pt(cone, np.concatenate([org, d]))
        if (pt > 9e7).all():
            continue
        n_hit += 1
        assert np.abs(pt - target).max() < 1e-6
        assert abs(uv[0] - u) < 1e-6
    assert n_hit > 200

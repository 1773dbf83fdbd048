"""CPU: the oracle against the committed golden vectors (tests/golden/, produced by
tests/golden/make_golden.py from the reference python run in the build container)."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def model_golden(golden_dir):
    return np.load(golden_dir + "/model_b2_n20000.npz")


@pytest.fixture(scope="module")
def oracle_run(model_oracle, state_dict, model_golden):
    from graspnet_amd import synthetic as S
    seeds = model_golden["model/cloud_seeds"]
    clouds = torch.stack([torch.from_numpy(S.make_cloud(int(seeds[0]), 20000, "uniform")),
                          torch.from_numpy(S.make_cloud(int(seeds[1]), 20000, "surface"))], 0)
    ep = model_oracle.forward({"point_clouds": clouds}, state_dict)
    return ep, model_oracle.pred_decode(ep)


def test_view_template_matches_reference(model_oracle, model_golden):
    assert np.array_equal(model_oracle.grasp_views(300).numpy(), model_golden["views300"])


def test_viewpoint_to_matrix_matches_reference(model_oracle, model_golden):
    tw = torch.from_numpy(model_golden["v2m/towards"])
    ang = torch.from_numpy(model_golden["v2m/angle"])
    # includes the axis_y == 0 branch (towards = [0,0,+-1]) in rows 0..3
    got = model_oracle.viewpoint_to_matrix(tw, ang).numpy()
    np.testing.assert_allclose(got, model_golden["v2m/matrix"], rtol=0, atol=1e-6)


def test_state_dict_manifest(state_dict, golden_dir):
    want = {}
    for line in open(golden_dir + "/state_dict_manifest.txt"):
        k, shp = line.split()
        want[k] = () if shp == "scalar" else tuple(int(x) for x in shp.split("x"))
    assert set(want) == set(state_dict)
    for k, v in state_dict.items():
        assert tuple(v.shape) == want[k], k
    assert len(want) == 162 + 25 * 0 or len(want) >= 162


def test_model_oracle_index_outputs_exact(oracle_run, model_golden):
    ep, _ = oracle_run
    for k in ("sa1_inds", "sa2_inds", "fp2_inds", "grasp_top_view_inds"):
        assert np.array_equal(ep[k].numpy(), model_golden["model/" + k]), k
    # SA2's FPS over FPS-ordered input is the identity prefix (backbone.py:110 comment)
    assert np.array_equal(ep["sa2_inds"][0].numpy(), np.arange(1024))


def test_model_oracle_float_outputs(oracle_run, model_golden):
    """Same torch-CPU ops in the same order: bit-identical in the container that made the
    fixtures; another host's BLAS/oneDNN may reorder sums, hence the tolerance (1e-4 on scores)."""
    ep, preds = oracle_run
    for k in ("fp2_xyz", "objectness_score", "grasp_top_view_score", "grasp_top_view_xyz", "grasp_top_view_rot"):
        np.testing.assert_allclose(ep[k].numpy(), model_golden["model/" + k], rtol=1e-4, atol=1e-4, err_msg=k)
    for k in ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features", "view_score",
              "grasp_score_pred", "grasp_angle_cls_pred", "grasp_width_pred", "grasp_tolerance_pred"):
        t = ep[k].contiguous()
        assert tuple(t.shape) == tuple(model_golden["model/%s/shape" % k]), k
        np.testing.assert_allclose(t.reshape(-1)[::97].numpy(), model_golden["model/%s/sample" % k],
                                   rtol=1e-4, atol=1e-4, err_msg=k)
        s, a = float(t.double().sum()), float(model_golden["model/%s/abssum" % k])
        assert abs(s - float(model_golden["model/%s/sum" % k])) <= 1e-5 * a + 1e-3, k
    for i, p in enumerate(preds):
        g = model_golden["model/pred/%d" % i]
        assert p.shape == g.shape
        np.testing.assert_allclose(p.numpy(), g, rtol=1e-4, atol=1e-4)
        assert np.all(p.numpy()[:, 16] == -1) and np.allclose(p.numpy()[:, 2], 0.02)


def test_ops_regression_vectors(oracle, golden_dir):
    z = np.load(golden_dir + "/ops_small.npz")
    for name in ("n1000", "n300", "n4096dup"):
        x = torch.from_numpy(z[name + "/pts"])
        m = z[name + "/fps"].shape[1]
        fi = oracle.furthest_point_sampling(x, m)
        assert np.array_equal(fi.numpy(), z[name + "/fps"]), name
        nx = oracle.gather_points(x.transpose(1, 2).contiguous(), fi).transpose(1, 2).contiguous()
        assert np.array_equal(oracle.ball_query(nx, x, 0.04, 16).numpy(), z[name + "/ball"])
        assert np.array_equal(oracle.ball_query(nx, x, 0.0005, 8).numpy(), z[name + "/ball_tiny"])
        rot = torch.from_numpy(z[name + "/rot"])
        assert np.array_equal(oracle.cylinder_query(nx, x, rot, 0.05, -0.02, 0.04, 16).numpy(), z[name + "/cyl"])
        d2, ti = oracle.three_nn(x[:, :200].contiguous(), nx)
        assert np.array_equal(ti.numpy(), z[name + "/nn_idx"])
        assert np.array_equal(d2.numpy(), z[name + "/nn_d2"])


def test_example_scene_fixture(model_oracle, state_dict, golden_dir):
    """doc/example_data scene (cloud made by the reference's create_point_cloud_from_depth_image): the oracle
    pipeline reproduces the committed rows; depth_to_cloud restatement reproduces the cloud from the stored pixels."""
    from oracle import post_oracle as PO
    z = np.load(golden_dir + "/example_scene_n20000.npz")
    h, w = int(z["shape"][0]), int(z["shape"][1])
    fx, fy, cx, cy, factor = [float(v) for v in z["intrinsics"]]
    depth = np.zeros(h * w, np.uint16)
    depth[z["pix"]] = z["depth_at_pix"]
    cloud = PO.depth_to_cloud(depth.reshape(h, w), fx, fy, cx, cy, factor)[z["pix"]].astype(np.float32)
    assert np.array_equal(cloud, z["cloud"])
    ep = model_oracle.forward({"point_clouds": torch.from_numpy(cloud).unsqueeze(0)}, state_dict)
    assert np.array_equal(ep["sa1_inds"].numpy(), z["sa1_inds"])
    g = model_oracle.pred_decode(ep)[0].numpy()
    assert g.shape == z["grasps"].shape
    np.testing.assert_allclose(g, z["grasps"], rtol=1e-4, atol=1e-4)
    scene = PO.voxel_down_sample(cloud, 0.01)
    assert len(scene) == int(z["n_scene_voxels"][0])
    cm, _, _ = PO.collision_detect(scene, z["grasps"], approach_dist=0.05, collision_thresh=0.01, voxel_size=0.01)
    assert np.array_equal(cm, z["collision_mask"])


@pytest.mark.parametrize("case", ["s1", "unit", "tied"])
def test_model_oracle_reproduces_the_variant_goldens(model_oracle, golden_dir, case):
    """tests/golden/model_<case>_n20000.npz (reference python run by make_golden_variants.py): more weight seeds, O(1)-score
    weights, exact arg-max ties.  The oracle on the same seeds: indices and the discrete choices exact, floats within 1e-4."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_golden_variants", os.path.join(golden_dir, "make_golden_variants.py"))
    mv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mv)
    g = np.load(os.path.join(golden_dir, "model_%s_n20000.npz" % case))
    sd, cl = mv.state_dict_of(case), mv.cloud_of(case)
    ep = model_oracle.forward({"point_clouds": cl}, sd)
    p = model_oracle.pred_decode(ep)[0].numpy()
    for k in ("sa1_inds", "sa2_inds", "grasp_top_view_inds"):
        assert np.array_equal(ep[k].numpy(), g["model/" + k]), k
    for k in ("view_score", "grasp_score_pred", "grasp_tolerance_pred"):
        np.testing.assert_allclose(ep[k].contiguous().reshape(-1)[::97].numpy(), g["model/%s/sample" % k], rtol=1e-4, atol=1e-4)
    assert p.shape == g["model/pred/0"].shape
    np.testing.assert_allclose(p, g["model/pred/0"], rtol=1e-4, atol=1e-4)
    if case == "unit":
        assert float(np.abs(p[:, 0]).max()) < 2.0
        np.testing.assert_allclose(p[:, 0], g["model/pred/0"][:, 0], rtol=0, atol=1e-4)      # plain absolute 1e-4 on scores
    if case == "tied":
        vs = ep["view_score"][0]
        assert torch.equal(vs[:, 0::3], vs[:, 1::3]) and bool((ep["grasp_top_view_inds"] % 3 == 0).all())
        assert int(g["tied/depth_score_ties"][0]) > 10

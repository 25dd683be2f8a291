"""Scene-level pipeline (pointcloudsegmentation_amd/inference.py = reference inference.py:143-188):
stage-by-stage agreement with the CPU checkers on one seeded scene."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_predict_matches_stagewise_checkers(ref_weights, ref_params):
    from oracle import oracle
    from oracle.preprocess_oracle import preprocess_point_cloud as ref_prep
    from pointcloudsegmentation_amd.inference import predict
    from pointcloudsegmentation_amd.point2net import Point2Net
    rng = np.random.default_rng(2)
    n = 6000
    pts = rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])
    col = rng.integers(0, 256, (n, 3)).astype(np.float64)
    m = Point2Net(feature_dim=6, embedding_dim=64)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref_weights.items()}, strict=True)
    m = m.cuda().eval()
    out = predict(m, pts, col, eps=0.1, min_samples=50)
    assert set(out) == {"instance_labels", "embeddings", "num_instances", "centroid", "scale"}
    feats, c, s = ref_prep(pts, col)
    np.testing.assert_array_equal(out["centroid"], c)
    assert out["scale"] == s
    want = oracle.forward_eval(feats, ref_params)["emb"][0]
    emb = out["embeddings"]
    assert emb.shape == (n, 64) and emb.dtype == np.float32
    assert np.abs(emb - want).max() <= 1e-5 * np.abs(want).max()
    # clustering is discontinuous in its input: check it on the embeddings the GPU produced
    lab = oracle.cluster_embeddings(emb, 0.1, 50)
    np.testing.assert_array_equal(out["instance_labels"], lab.astype(np.int64))
    assert out["instance_labels"].dtype == np.int64
    assert out["num_instances"] == len(np.unique(lab)) - 1


def test_predict_requires_eval_mode(ref_weights):
    from pointcloudsegmentation_amd.inference import predict
    from pointcloudsegmentation_amd.point2net import Point2Net
    m = Point2Net().cuda().train()
    with pytest.raises(RuntimeError):
        predict(m, np.zeros((10, 3)))


def test_point_cloud_inference_class_file_to_file(ref_weights, tmp_path):
    """The reference's class surface (inference.py:43-219) from checkpoint + YAML + PLY on disk to the
    three result files, against predict() on the same arrays."""
    import yaml
    from pointcloudsegmentation_amd import io as p2n_io
    from pointcloudsegmentation_amd.inference import PointCloudInference, predict
    rng = np.random.default_rng(5)
    n = 3000
    pts = rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])
    col = rng.integers(0, 256, (n, 3)) / 255.0
    p2n_io.write_point_cloud_ply(tmp_path / "scene.ply", pts, col)
    torch.save({"epoch": 1, "model_state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in ref_weights.items()}},
               tmp_path / "ckpt.pth")
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump({"model": {"embedding_dim": 64}}))
    inf = PointCloudInference(str(tmp_path / "ckpt.pth"), str(tmp_path / "cfg.yaml"), device="cuda")
    assert not inf.model.training
    res = inf.run(str(tmp_path / "scene.ply"), str(tmp_path / "out" / "scene_labeled.ply"), eps=0.1, min_samples=20)
    rp, rc, _ = p2n_io.read_point_cloud_ply(tmp_path / "scene.ply")
    want = predict(inf.model, rp, rc, eps=0.1, min_samples=20)
    np.testing.assert_array_equal(res["instance_labels"], want["instance_labels"])
    np.testing.assert_array_equal(np.load(tmp_path / "out" / "scene_labeled_labels.npy"), want["instance_labels"])
    _, lc, _ = p2n_io.read_point_cloud_ply(tmp_path / "out" / "scene_labeled.ply")
    np.testing.assert_array_equal(inf.labeler.colors_to_labels(lc), want["instance_labels"])
    assert (tmp_path / "out" / "scene_labeled_original.ply").exists()
    np.testing.assert_array_equal(inf.cluster_embeddings(res["embeddings"], 0.1, 20), want["instance_labels"])
    with pytest.raises(RuntimeError):
        PointCloudInference(str(tmp_path / "ckpt.pth"), str(tmp_path / "cfg.yaml"), device="cpu")


def test_predict_many_pipeline_matches_predict(ref_weights):
    """The three-stream pipeline (upload of scene i+1 beside the compute of scene i, results leaving on a
    third stream) must return exactly what predict() returns scene by scene: same labels, bit-identical
    embeddings, centroid, scale, instance count -- on ragged scene sizes, with and without colours."""
    from pointcloudsegmentation_amd.inference import predict, predict_many
    from pointcloudsegmentation_amd.point2net import Point2Net
    rng = np.random.default_rng(9)
    m = Point2Net(feature_dim=6, embedding_dim=64)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref_weights.items()}, strict=True)
    m = m.cuda().eval()
    scenes = []
    for n in (5000, 1200, 9001, 300, 4096):
        pts = rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])
        col = rng.integers(0, 256, (n, 3)).astype(np.float64) if n != 300 else None
        scenes.append((pts, col))
    for overlap in (True, False):
        got = list(predict_many(m, scenes, eps=0.1, min_samples=20, overlap=overlap))
        assert len(got) == len(scenes)
        for (pts, col), g in zip(scenes, got):
            w = predict(m, pts, col, eps=0.1, min_samples=20)
            np.testing.assert_array_equal(g["instance_labels"], w["instance_labels"])
            np.testing.assert_array_equal(g["embeddings"], w["embeddings"])
            np.testing.assert_array_equal(g["centroid"], w["centroid"])
            assert g["scale"] == w["scale"] and g["num_instances"] == w["num_instances"]
            assert g["instance_labels"].dtype == np.int64 and g["embeddings"].dtype == np.float32
    assert list(predict_many(m, [])) == []

"""On-disk formats / metrics next to the hot path: consistentnerf_amd.io_formats against values
parsed by the reference's own readers (fixture `formats`, tests/golden/make_golden.py::fx_formats) — CPU only."""
import os

import numpy as np
import torch

from conftest import golden
from consistentnerf_amd import io_formats as F


def test_read_pfm_matches_reference_reader(tmp_path):
    g = golden("formats")
    for tag in ("grey_le", "grey_be", "color_le"):
        p = os.path.join(tmp_path, tag + ".pfm")
        open(p, "wb").write(g[f"pfm_{tag}_bytes"].tobytes())
        data, scale = F.read_pfm(p)
        assert data.dtype == np.float32 and np.array_equal(data, g[f"pfm_{tag}_data"]), tag
        assert scale == float(g[f"pfm_{tag}_scale"])
    # writer is the inverse (bit-exact), both byte orders
    for le in (True, False):
        p = os.path.join(tmp_path, f"rt_{le}.pfm")
        F.write_pfm(p, g["pfm_color_le_data"], scale=2.0, little_endian=le)
        data, scale = F.read_pfm(p)
        assert np.array_equal(data, g["pfm_color_le_data"]) and scale == 2.0
    bad = os.path.join(tmp_path, "bad.pfm")
    open(bad, "wb").write(b"P6\n1 1\n-1\n")
    try:
        F.read_pfm(bad)
        assert False, "expected 'Not a PFM file.'"
    except Exception as e:
        assert "Not a PFM file" in str(e)


def test_read_cam_file_matches_reference_reader(tmp_path):
    g = golden("formats")
    p = os.path.join(tmp_path, "00000000_cam.txt")
    open(p, "w").write(g["cam_text"].tobytes().decode())
    intr, extr, dr = F.read_cam_file(p)
    assert np.array_equal(intr, g["cam_intrinsics"]) and np.array_equal(extr, g["cam_extrinsics"])
    assert np.allclose(dr, g["cam_depth_range"], rtol=0, atol=1e-9)


def test_masked_psnr_and_metrics_file(tmp_path):
    g = golden("formats")
    psnr = F.img2psnr_mask(torch.from_numpy(g["psnr_x"]), torch.from_numpy(g["psnr_y"]), torch.from_numpy(g["psnr_mask"]))
    assert abs(float(psnr) - float(g["psnr"])) < 1e-6
    p = os.path.join(tmp_path, "metrics.txt")
    F.write_metrics(p, 23.5, 0.81, 0.2)
    assert open(p).read() == "PSNR: 23.5\nSSIM: 0.81\nLPIPS: 0.2"


def test_pairs_split_lists():
    g = golden("pairs")
    ref = "/root/reference/nerf-pytorch-master/configs/pairs.th"
    if os.path.exists(ref):            # present in the build container only; the decoded lists are the committed fixture
        d = F.load_pairs(ref)
        assert set(d) == set(g.keys())
        for k in d:
            assert np.array_equal(d[k], g[k])
    assert any(k.endswith("_train") for k in g.keys())


def test_llff_pose_pipeline_and_blender_ring():
    """llff_poses vs the reference's load_llff_data driven on the same raw poses_bounds array (axis reorder, bound
    rescale, recentring, hold-out view, 60-pose spiral); pose_spherical vs load_blender's."""
    g = golden("poses")
    hw, factor = tuple(int(v) for v in g["hw"]), int(g["factor"])
    for tag, kw in (("default", {}), ("norecenter", dict(recenter=False)), ("nobd", dict(bd_factor=None))):
        poses, bds, render_poses, i_test = F.llff_poses(g["poses_bounds"].copy(), hw, factor, **kw)
        assert poses.dtype == np.float32 and render_poses.shape == (60, 3, 4) and render_poses.dtype == np.float32
        np.testing.assert_allclose(poses, g[f"{tag}_poses"], rtol=0, atol=1e-6 * np.abs(g[f"{tag}_poses"]).max())
        np.testing.assert_allclose(bds, g[f"{tag}_bds"], rtol=1e-7)
        np.testing.assert_allclose(render_poses, g[f"{tag}_render"], rtol=0, atol=2e-6)
        assert i_test == int(g[f"{tag}_itest"])
    for a, ref in zip(g["sph_args"], g["sph"]):
        np.testing.assert_allclose(F.pose_spherical(*a), ref, rtol=0, atol=1e-6)
    meta = {"camera_angle_x": 0.6911112070083618,
            "frames": [{"file_path": f"./train/r_{k}", "transform_matrix": F.pose_spherical(30.0 * k, -30.0, 4.0).tolist()}
                       for k in range(3)]}
    poses, focal, files = F.read_transforms(meta, 800)
    assert poses.shape == (3, 4, 4) and files[2] == "./train/r_2" and abs(focal - 1111.111) < 1e-2
    np.testing.assert_array_equal(poses[1], F.pose_spherical(30.0, -30.0, 4.0))

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

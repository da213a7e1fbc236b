"""On-disk formats and evaluation metrics next to the hot path — host-side, numpy / torch only.

  read_pfm / write_pfm      V:103-138 (= load_dtu.py:141-176): MVSNet depth maps, little/big endian, rows bottom-up
  read_cam_file             load_dtu.py:120-132: DTU `*_cam.txt` -> (intrinsics 3x3, extrinsics 4x4, [depth_min, depth_max])
  load_pairs                configs/pairs.th split lists (a torch zip archive holding a plain pickle of numpy arrays)
  img2psnr_mask             alky/vis_utils.py:24-42: mean over images of the PSNR of the foreground-masked MSE
  write_metrics             V:2078-2087: metrics.txt
  llff_poses                load_llff.py: `poses_bounds.npy` -> poses / bounds / 60-pose spiral render path / hold-out view
  pose_spherical, read_transforms   load_blender.py:30-35, 38-70, 212-214: Blender camera ring and `transforms_*.json`
"""
import io
import pickle
import re
import zipfile

import numpy as np
import torch


def read_pfm(filename):
    """-> (data [H, W] or [H, W, 3] float32, rows top-down, scale)."""
    with open(filename, 'rb') as f:
        header = f.readline().decode('utf-8').rstrip()
        if header == 'PF':
            color = True
        elif header == 'Pf':
            color = False
        else:
            raise Exception('Not a PFM file.')
        m = re.match(r'^(\d+)\s(\d+)\s$', f.readline().decode('utf-8'))
        if not m:
            raise Exception('Malformed PFM header.')
        width, height = map(int, m.groups())
        scale = float(f.readline().rstrip())
        endian = '<' if scale < 0 else '>'
        scale = abs(scale)
        data = np.frombuffer(f.read(), dtype=endian + 'f4')
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)).astype(np.float32), scale   # (native byte order; the reference keeps '>f4')


def write_pfm(filename, data, scale=1.0, little_endian=True):
    """Inverse of read_pfm (what MVSNet's writers produce): rows stored bottom-up."""
    data = np.asarray(data, dtype=np.float32)
    color = data.ndim == 3 and data.shape[2] == 3
    if not color and data.ndim != 2:
        raise ValueError("PFM holds [H, W] or [H, W, 3]")
    with open(filename, 'wb') as f:
        f.write(b'PF\n' if color else b'Pf\n')
        f.write(f'{data.shape[1]} {data.shape[0]}\n'.encode())
        f.write(f'{-abs(scale) if little_endian else abs(scale)}\n'.encode())
        f.write(np.flipud(data).astype('<f4' if little_endian else '>f4').tobytes())


def read_cam_file(filename):
    """DTU / MVSNet camera file: 'extrinsic' + 4 rows, blank, 'intrinsic' + 3 rows, blank, 'depth_min depth_interval'."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(' '.join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(' '.join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    depth_min = float(lines[11].split()[0])
    depth_max = depth_min + float(lines[11].split()[1]) * 192 * 1.06
    return intrinsics, extrinsics, [depth_min, depth_max]


def load_pairs(path):
    """{'<scene>_train' / '_val' / '_test': int array} from configs/pairs.th without unpickling arbitrary classes."""
    with zipfile.ZipFile(path) as zf:
        raw = zf.read([n for n in zf.namelist() if n.endswith('data.pkl')][0])

    class _U(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split('.')[0] in ('numpy', 'collections', '_codecs'):
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f'blocked {module}.{name}')

        def persistent_load(self, pid):
            raise pickle.UnpicklingError('no tensors expected')
    return {k: np.asarray(v) for k, v in _U(io.BytesIO(raw)).load().items()}


def img2psnr_mask(x, y, mask):
    """x, y [N, H, W, 3], mask [N, H, W] -> mean over the N images of -10 log10(masked-mean squared error)."""
    n = x.shape[0]
    mses = ((x - y) ** 2).mean(-1)
    mses = (mses * mask).reshape(n, -1).sum(-1) / mask.reshape(n, -1).sum(-1)
    ten = torch.tensor([10.], device=mses.device)
    return torch.stack([-10. * torch.log(m) / torch.log(ten) for m in mses]).mean()


def write_metrics(path, psnr, ssim, lpips):
    with open(path, 'w') as f:
        f.write(f'PSNR: {psnr}\n')
        f.write(f'SSIM: {ssim}\n')
        f.write(f'LPIPS: {lpips}')


# ---- camera files of the two other dataset families: poses only, images stay the caller's ------------
def pose_spherical(theta, phi, radius):
    """load_blender.py:30-35: camera-to-world of a camera on a sphere (degrees), the Blender render path / the C1
    synthetic views -> [4, 4] float32.  Composed in float32 like the reference's torch.Tensor factors."""
    t, p = np.float64(theta) / 180. * np.pi, np.float64(phi) / 180. * np.pi
    f = np.float32
    trans = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=f)
    rphi = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]], dtype=f)
    rth = np.array([[np.cos(t), 0, -np.sin(t), 0], [0, 1, 0, 0], [np.sin(t), 0, np.cos(t), 0], [0, 0, 0, 1]], dtype=f)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=f)
    return (flip @ (rth @ (rphi @ trans))).astype(f)


def read_transforms(meta, W):
    """Blender `transforms_*.json` (load_blender.py:38-70, 212-214), already parsed (a dict) or a path ->
    (poses [N, 4, 4] float32, focal = .5 W / tan(.5 camera_angle_x), file paths)."""
    if not isinstance(meta, dict):
        import json
        with open(meta, 'r') as fp:
            meta = json.load(fp)
    poses = np.array([np.array(fr['transform_matrix']) for fr in meta['frames']]).astype(np.float32)
    focal = .5 * W / np.tan(.5 * float(meta['camera_angle_x']))
    return poses, focal, [fr['file_path'] for fr in meta['frames']]


def _unit(v):
    return v / np.linalg.norm(v)


def viewmatrix(z, up, pos):
    """load_llff.py:141-147 -> [3, 4] (x, y, z axes and position as columns)."""
    z = _unit(z)
    x = _unit(np.cross(up, z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, pos], 1)


def poses_avg(poses):
    """load_llff.py:153-163: the 'average' camera [3, 5] (mean position, summed viewing and up directions, hwf of view 0)."""
    c2w = viewmatrix(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([c2w, poses[0, :3, -1:]], 1)


def recenter_poses(poses):
    """load_llff.py:205-218: express every pose in the frame of the average camera."""
    out = poses + 0
    row = np.array([[0, 0, 0, 1.]])
    avg = np.concatenate([poses_avg(poses)[:3, :4], row], 0)
    full = np.concatenate([poses[:, :3, :4], np.tile(row[None], [poses.shape[0], 1, 1])], 1)
    out[:, :3, :4] = (np.linalg.inv(avg) @ full)[:, :3, :4]
    return out


def generate_spiral_path(poses, bounds, n_frames=120, n_rots=2, zrate=.5):
    """load_llff.py:178-202: forward-facing spiral of `n_frames` poses [n, 3, 4] around the average camera; radii = 90th
    percentile of |camera position| per axis, focus depth from the bounds in disparity space."""
    near, far = bounds.min() * .9, bounds.max() * 5.
    dt = .75
    focal = 1 / ((1 - dt) / near + dt / far)
    radii = np.concatenate([np.percentile(np.abs(poses[:, :3, 3]), 90, 0), [1.]])
    c2w = poses_avg(poses)[:3, :4]
    up = poses[:, :3, 1].mean(0)
    out = []
    for th in np.linspace(0., 2. * np.pi * n_rots, n_frames, endpoint=False):
        pos = c2w @ (radii * [np.cos(th), -np.sin(th), -np.sin(th * zrate), 1.])
        out.append(viewmatrix(pos - c2w @ [0, 0, -focal, 1.], up, pos))
    return np.stack(out, 0)


def llff_poses(poses_arr, image_hw, factor=8, recenter=True, bd_factor=.75, n_render=60):
    """The pose half of load_llff_data (load_llff.py:64-66, 96-98, 290-365) from the raw `poses_bounds.npy` array
    [N, 17] and the size of the (down-scaled) images: -> (poses [N, 3, 5] float32 with hwf in the last column,
    bds [N, 2], render_poses [n_render, 3, 4] float32 — the spiral `render_path` is fed (C5) —, i_test = the view
    closest to the average camera)."""
    poses_arr = np.array(poses_arr, copy=True)      # the in-place hwf edits below must not reach the caller's array
    poses = poses_arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = poses_arr[:, -2:].transpose([1, 0])
    poses[:2, 4, :] = np.array(image_hw[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    # LLFF stores rotations as (down, right, back): reorder to (right, up, back); views to axis 0
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    c2w = poses_avg(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))
    poses = poses.astype(np.float32)
    render_poses = np.array(generate_spiral_path(poses[:, :3, :4], bds, n_render)).astype(np.float32)
    return poses, bds, render_poses, i_test

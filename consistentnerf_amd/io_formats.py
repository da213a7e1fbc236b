"""On-disk formats and evaluation metrics next to the hot path — host-side, numpy / torch only.

  read_pfm / write_pfm      V:103-138 (= load_dtu.py:141-176): MVSNet depth maps, little/big endian, rows bottom-up
  read_cam_file             load_dtu.py:120-132: DTU `*_cam.txt` -> (intrinsics 3x3, extrinsics 4x4, [depth_min, depth_max])
  load_pairs                configs/pairs.th split lists (a torch zip archive holding a plain pickle of numpy arrays)
  img2psnr_mask             alky/vis_utils.py:24-42: mean over images of the PSNR of the foreground-masked MSE
  write_metrics             V:2078-2087: metrics.txt
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

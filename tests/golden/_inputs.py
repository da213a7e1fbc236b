"""Deterministic, numpy-only input builders shared by the golden-vector generator
(`make_golden.py`, runs only where /root/reference exists) and by the tests (run anywhere).

Everything here is derived from `np.random.RandomState(seed)` (the frozen legacy MT19937
streams), so fixtures only need to store the *outputs* the reference produced.
"""
import numpy as np

SKIPS = (4,)


def embed_channels(multires):
    return 3 + 6 * multires


def nerf_param_shapes(D, W, input_ch, input_ch_views, output_ch, use_viewdirs, skips=SKIPS):
    """(name, shape) list in the reference's state_dict order (run_nerf_helpers.py:67-104)."""
    shapes = [("temp_rgb", (1,)), ("temp_depth", (1,)), ("depth_scale", (1,))]
    shapes += [("pts_linears.0.weight", (W, input_ch)), ("pts_linears.0.bias", (W,))]
    for i in range(D - 1):
        k = W + input_ch if i in skips else W
        shapes += [(f"pts_linears.{i+1}.weight", (W, k)), (f"pts_linears.{i+1}.bias", (W,))]
    shapes += [("views_linears.0.weight", (W // 2, input_ch_views + W)), ("views_linears.0.bias", (W // 2,))]
    if use_viewdirs:
        shapes += [("feature_linear.weight", (W, W)), ("feature_linear.bias", (W,)),
                   ("alpha_linear.weight", (1, W)), ("alpha_linear.bias", (1,)),
                   ("rgb_linear.weight", (3, W // 2)), ("rgb_linear.bias", (3,))]
    else:
        shapes += [("output_linear.weight", (output_ch, W)), ("output_linear.bias", (output_ch,))]
    return shapes


def nerf_state_dict(D, W, multires=10, multires_views=4, output_ch=4, use_viewdirs=True, seed=0,
                    gain=1.0):
    """He-uniform weights (keeps activations O(1) through 8 layers so ReLU masks, sigmoids and
    densities are all exercised), small uniform biases. float32 numpy arrays keyed like the
    reference state_dict."""
    rs = np.random.RandomState(seed)
    input_ch = embed_channels(multires)
    input_ch_views = embed_channels(multires_views) if use_viewdirs else 0
    sd = {}
    for name, shape in nerf_param_shapes(D, W, input_ch, input_ch_views, output_ch, use_viewdirs):
        if name == "temp_rgb" or name == "temp_depth":
            sd[name] = np.full(shape, -0.7, np.float32)
        elif name == "depth_scale":
            sd[name] = np.full(shape, 1.0, np.float32)
        elif name.endswith(".weight"):
            bound = gain * np.sqrt(6.0 / shape[1])
            sd[name] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
        else:
            sd[name] = rs.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    return sd


def ray_batch(B, seed, near=2.0, far=6.0, use_viewdirs=True):
    """[B, 11] = o(3) d(3) near far viewdirs(3) like run_nerf.py:119-125. Cameras on a shell of
    radius ~4 looking roughly at the origin; d is NOT unit length (get_rays-style, |d|>=1)."""
    rs = np.random.RandomState(seed)
    o = rs.normal(size=(B, 3))
    o = 4.0 * o / np.linalg.norm(o, axis=-1, keepdims=True)
    tgt = rs.uniform(-0.8, 0.8, size=(B, 3))
    d = tgt - o
    d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rs.uniform(1.0, 1.3, size=(B, 1))
    o = o.astype(np.float32)
    d = d.astype(np.float32)
    cols = [o, d, np.full((B, 1), near, np.float32), np.full((B, 1), far, np.float32)]
    if use_viewdirs:
        # float32 arithmetic, exactly like render(): viewdirs = d / ||d|| (run_nerf.py:109)
        import torch
        td = torch.from_numpy(d)
        cols.append((td / torch.norm(td, dim=-1, keepdim=True)).numpy())
    return np.concatenate(cols, -1).astype(np.float32)


def raw2outputs_inputs(B, S, seed, near=2.0, far=6.0):
    rs = np.random.RandomState(seed)
    raw = (rs.normal(size=(B, S, 4)) * 3.0).astype(np.float32)
    # a few rays with all-negative sigma -> acc == 0 -> disp NaN (reference behaviour, R:302)
    raw[:2, :, 3] = -np.abs(raw[:2, :, 3]) - 0.1
    z = np.sort(rs.uniform(near, far, size=(B, S)), -1).astype(np.float32)
    d = rs.normal(size=(B, 3)).astype(np.float32)
    return raw, z, d


def sample_pdf_inputs(B, Nc, seed, near=2.0, far=6.0):
    rs = np.random.RandomState(seed)
    z = np.sort(rs.uniform(near, far, size=(B, Nc)), -1).astype(np.float32)
    bins = (0.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)            # [B, Nc-1]
    weights = (rs.uniform(size=(B, Nc - 2)) ** 8).astype(np.float32)     # peaky
    weights[:3] = 0.0                                                    # all-zero rows: pdf = uniform via the 1e-5 floor
    return bins, weights


def camera_pose(theta_deg, phi_deg, radius):
    """c2w [3,4] looking at the origin (OpenGL convention: camera looks down -z, y up)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    c = radius * np.array([np.cos(ph) * np.sin(th), np.sin(ph), np.cos(ph) * np.cos(th)])
    fwd = -c / np.linalg.norm(c)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0])); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    R = np.stack([right, up, -fwd], 1)
    return np.concatenate([R, c[:, None]], 1).astype(np.float32)


def intrinsics(H, W, focal):
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], np.float32)


def analytic_scene(H, W, K, c2w, sphere_r=1.0, plane_z=-1.5):
    """Depth prior + colours of a unit sphere in front of a plane, seen from c2w.
    Returns depth (distance along the un-normalised get_rays direction, i.e. camera-z depth),
    rgb[H,W,3]. Pure numpy float64 -> float32."""
    j, i = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    dirs = np.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -np.ones_like(i)], -1)
    rd = dirs @ c2w[:3, :3].astype(np.float64).T
    ro = c2w[:3, 3].astype(np.float64)
    a = (rd * rd).sum(-1); b = 2 * (rd * ro).sum(-1); c = (ro * ro).sum() - sphere_r ** 2
    disc = b * b - 4 * a * c
    t_s = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    t_s = np.where(t_s > 0, t_s, np.inf)
    # plane y = plane_z (a floor)
    t_p = np.where(np.abs(rd[..., 1]) > 1e-9, (plane_z - ro[1]) / rd[..., 1], np.inf)
    t_p = np.where(t_p > 0, t_p, np.inf)
    t = np.minimum(t_s, t_p)
    t = np.where(np.isfinite(t), t, 8.0)
    P = ro + t[..., None] * rd
    rgb = 0.5 + 0.5 * np.sin(3.0 * P + np.array([0.0, 1.0, 2.0]))
    return t.astype(np.float32), rgb.astype(np.float32)

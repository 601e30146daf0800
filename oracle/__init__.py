"""CPU oracle for the MOVEDepth hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy front-end over oracle/movedepth_oracle.c (plain C, fp32).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
movedepth_amd/ never does.  Pinned against the reference's own outputs in
tests/golden/ (tests/test_oracle_golden.py).  See the C file's header.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmovedepth_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int
_fl = ctypes.c_float


def build(force=False):
    """Compile the C restatement with gcc (called by __graft_entry__.build())."""
    src = os.path.join(_HERE, "movedepth_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmovedepth_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.mdo_smooth_fwd.restype = ctypes.c_float
        _lib.mdo_num_threads.restype = ctypes.c_int
    return _lib


def set_num_threads(n):
    lib().mdo_set_num_threads(int(n))


def num_threads():
    return int(lib().mdo_num_threads())


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_f)


# ------------------------------------------------------------------ geometry
def backproject_project(depth, invK, K, T, h, w, eps=1e-7):
    """depth [Bs,h*w] or [Bs,1,h,w]; invK/K/T [nk,4,4] with nk in {1,Bs}. -> cam_points [Bs,4,hw], pix [Bs,h,w,2]"""
    depth = _c(depth).reshape(-1, h * w)
    Bs = depth.shape[0]
    invK, K, T = _c(invK).reshape(-1, 16), _c(K).reshape(-1, 16), _c(T).reshape(-1, 16)
    nk = invK.shape[0]
    assert nk in (1, Bs) and K.shape[0] == nk and T.shape[0] == nk
    cam = np.empty((Bs, 4, h * w), np.float32)
    pix = np.empty((Bs, h, w, 2), np.float32)
    lib().mdo_backproject_project(_p(depth), _p(invK), _p(K), _p(T), _i(Bs), _i(nk), _i(h), _i(w), _fl(eps),
                                  _p(cam), _p(pix))
    return cam, pix


def disp_to_depth(disp, min_depth, max_depth):
    disp = _c(disp)
    s, d = np.empty_like(disp), np.empty_like(disp)
    lib().mdo_disp_to_depth(_p(disp), _i(disp.size), _fl(min_depth), _fl(max_depth), _p(s), _p(d))
    return s, d


def transformation_from_parameters(axisangle, translation, invert=False):
    aa, tr = _c(axisangle).reshape(-1, 3), _c(translation).reshape(-1, 3)
    out = np.empty((aa.shape[0], 4, 4), np.float32)
    lib().mdo_transformation_from_parameters(_p(aa), _p(tr), _i(aa.shape[0]), _i(int(invert)), _p(out))
    return out


def transformation_from_parameters_bwd(gT, axisangle, translation, invert=False):
    """adjoint of transformation_from_parameters: gT [B,4,4] -> (d_axisangle [B,3], d_translation [B,3])"""
    aa, tr, g = _c(axisangle).reshape(-1, 3), _c(translation).reshape(-1, 3), _c(gT).reshape(-1, 16)
    d_aa, d_tr = np.empty_like(aa), np.empty_like(tr)
    lib().mdo_transformation_from_parameters_bwd(_p(g), _p(aa), _p(tr), _i(aa.shape[0]), _i(int(invert)), _p(d_aa), _p(d_tr))
    return d_aa, d_tr


_TYPES = {"inverse": 0, "linear": 1, "log": 2}


def schedule_depth_range(prior, ndepth, scale_fac, z_trans=None, type="inverse"):
    """prior [B,1,h,w]; z_trans None (v2) or [B] (zv2). -> [B,D,h,w]"""
    prior = _c(prior)
    B, _, h, w = prior.shape
    z = None if z_trans is None else _c(z_trans).reshape(B)
    out = np.empty((B, ndepth, h, w), np.float32)
    lib().mdo_schedule(_p(prior), _p(z), _i(B), _i(h * w), _i(ndepth), _fl(scale_fac), _i(_TYPES[type]), _p(out))
    return out


# ------------------------------------------------------------------ cost volume
def costvol(ref, src, K, invK, hyp, pose):
    ref, src, hyp = _c(ref), _c(src), _c(hyp)
    B, C, h, w = ref.shape
    D = hyp.shape[1]
    K, invK, pose = _c(K).reshape(B, 16), _c(invK).reshape(B, 16), _c(pose).reshape(B, 16)
    out = np.empty((B, D, C, h, w), np.float32)
    lib().mdo_costvol_fwd(_p(ref), _p(src), _p(K), _p(invK), _p(hyp), _p(pose), _i(B), _i(C), _i(h), _i(w), _i(D),
                          _p(out))
    return out


def costvol_grouped(ref, src, K, invK, hyp, pose, G):
    ref, src, hyp = _c(ref), _c(src), _c(hyp)
    B, C, h, w = ref.shape
    D = hyp.shape[1]
    K, invK, pose = _c(K).reshape(B, 16), _c(invK).reshape(B, 16), _c(pose).reshape(B, 16)
    out = np.empty((B, D, G, h, w), np.float32)
    lib().mdo_costvol_grouped_fwd(_p(ref), _p(src), _p(K), _p(invK), _p(hyp), _p(pose), _i(B), _i(C), _i(G), _i(h),
                                  _i(w), _i(D), _p(out))
    return out


def costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose):
    gout, ref, src, hyp = _c(gout), _c(ref), _c(src), _c(hyp)
    B, C, h, w = ref.shape
    D, G = gout.shape[1], gout.shape[2]
    K, invK, pose = _c(K).reshape(B, 16), _c(invK).reshape(B, 16), _c(pose).reshape(B, 16)
    d_ref, d_src = np.empty_like(ref), np.empty_like(src)
    lib().mdo_costvol_grouped_bwd(_p(gout), _p(ref), _p(src), _p(K), _p(invK), _p(hyp), _p(pose), _i(B), _i(C),
                                  _i(G), _i(h), _i(w), _i(D), _p(d_ref), _p(d_src))
    return d_ref, d_src


def _ptr_array(arrs):
    arr_t = _f * len(arrs)
    return arr_t(*[a.ctypes.data_as(_f) for a in arrs])


def fuse(vols):
    """vols: list of [B,D,G,h,w] -> (cor_feats [B,D,G,h,w], weights [N,B,h,w])"""
    vols = [_c(v) for v in vols]
    B, D, G, h, w = vols[0].shape
    out = np.empty_like(vols[0])
    wts = np.empty((len(vols), B, h, w), np.float32)
    lib().mdo_fuse_fwd(_ptr_array(vols), _i(len(vols)), _i(B), _i(D), _i(G), _i(h * w), _p(out), _p(wts))
    return out, wts


def fuse_eval(vols):
    """evaluate_depth.py:225-243 (weights: soft-max over D of the mean over G). -> (cor_feats, weights [N,B,h,w])"""
    vols = [_c(v) for v in vols]
    B, D, G, h, w = vols[0].shape
    out = np.empty_like(vols[0])
    wts = np.empty((len(vols), B, h, w), np.float32)
    lib().mdo_fuse_eval_fwd(_ptr_array(vols), _i(len(vols)), _i(B), _i(D), _i(G), _i(h * w), _p(out), _p(wts))
    return out, wts


def fuse_bwd(gout, vols):
    vols = [_c(v) for v in vols]
    gout = _c(gout)
    B, D, G, h, w = vols[0].shape
    outs = [np.empty_like(v) for v in vols]
    lib().mdo_fuse_bwd(_p(gout), _ptr_array(vols), _i(len(vols)), _i(B), _i(D), _i(G), _i(h * w), _ptr_array(outs))
    return outs


# ------------------------------------------------------------------ photometric
def warp(img, depth, K, invK, T):
    """img [B,C,H,W], depth [B,1,H,W] or [B,H,W] -> (warped [B,C,H,W], pix [B,H,W,2])"""
    img = _c(img)
    B, C, H, W = img.shape
    depth = _c(depth).reshape(B, H, W)
    K, invK, T = _c(K).reshape(B, 16), _c(invK).reshape(B, 16), _c(T).reshape(B, 16)
    out = np.empty_like(img)
    pix = np.empty((B, H, W, 2), np.float32)
    lib().mdo_warp_fwd(_p(img), _p(depth), _p(K), _p(invK), _p(T), _i(B), _i(C), _i(H), _i(W), _p(pix), _p(out))
    return out, pix


def warp_bwd(gout, img, depth, K, invK, T):
    img, gout = _c(img), _c(gout)
    B, C, H, W = img.shape
    depth = _c(depth).reshape(B, H, W)
    K, invK, T = _c(K).reshape(B, 16), _c(invK).reshape(B, 16), _c(T).reshape(B, 16)
    d_depth = np.empty((B, H, W), np.float32)
    d_T = np.empty((B, 4, 4), np.float32)
    lib().mdo_warp_bwd(_p(gout), _p(img), _p(depth), _p(K), _p(invK), _p(T), _i(B), _i(C), _i(H), _i(W),
                       _p(d_depth), _p(d_T))
    return d_depth, d_T


def resize_bilinear(x, H, W):
    x = _c(x)
    lead, (h, w) = x.shape[:-2], x.shape[-2:]
    N = int(np.prod(lead)) if lead else 1
    out = np.empty(lead + (H, W), np.float32)
    lib().mdo_resize_bilinear_fwd(_p(x), _i(N), _i(h), _i(w), _i(H), _i(W), _p(out))
    return out


def resize_bilinear_bwd(gout, h, w):
    gout = _c(gout)
    lead, (H, W) = gout.shape[:-2], gout.shape[-2:]
    N = int(np.prod(lead)) if lead else 1
    gin = np.empty(lead + (h, w), np.float32)
    lib().mdo_resize_bilinear_bwd(_p(gout), _i(N), _i(h), _i(w), _i(H), _i(W), _p(gin))
    return gin


def ssim(x, y):
    x, y = _c(x), _c(y)
    H, W = x.shape[-2:]
    out = np.empty_like(x)
    lib().mdo_ssim(_p(x), _p(y), _i(x.size // (H * W)), _i(H), _i(W), _p(out))
    return out


def reproj_loss(pred, target, ssim_w=0.85, no_ssim=False):
    pred, target = _c(pred), _c(target)
    B, C, H, W = pred.shape
    out = np.empty((B, 1, H, W), np.float32)
    lib().mdo_reproj_loss_fwd(_p(pred), _p(target), _i(B), _i(C), _i(H), _i(W), _fl(ssim_w), _i(int(no_ssim)), _p(out))
    return out


def reproj_loss_bwd(gout, pred, target, ssim_w=0.85, no_ssim=False):
    pred, target, gout = _c(pred), _c(target), _c(gout)
    B, C, H, W = pred.shape
    d = np.empty_like(pred)
    lib().mdo_reproj_loss_bwd(_p(gout), _p(pred), _p(target), _i(B), _i(C), _i(H), _i(W), _fl(ssim_w),
                              _i(int(no_ssim)), _p(d))
    return d


def masked_min(reproj, ident=None, noise=None, ext_mask=None, mvs_mode=False):
    """reproj/ident [B,N,H,W]; noise/ext_mask [B,1,H,W]. -> (min [B,1,H,W], mask [B,1,H,W], loss float)"""
    reproj = _c(reproj)
    B, N, H, W = reproj.shape
    ident = None if ident is None else _c(ident)
    noise = None if noise is None else _c(noise)
    ext_mask = None if ext_mask is None else _c(ext_mask)
    mn, mask = np.empty((B, 1, H, W), np.float32), np.empty((B, 1, H, W), np.float32)
    loss = np.zeros(1, np.float32)
    lib().mdo_masked_min_fwd(_p(reproj), _p(ident), _p(noise), _p(ext_mask), _i(B), _i(N), _i(H * W),
                             _i(int(mvs_mode)), _p(mn), _p(mask), _p(loss))
    return mn, mask, float(loss[0])


def masked_min_bwd(gloss, reproj, mask):
    reproj, mask = _c(reproj), _c(mask)
    B, N, H, W = reproj.shape
    d = np.empty_like(reproj)
    lib().mdo_masked_min_bwd(_fl(gloss), _p(reproj), _p(mask), _i(B), _i(N), _i(H * W), _p(d))
    return d


def smooth_loss(disp, img, normalize=True):
    disp, img = _c(disp), _c(img)
    B, C, h, w = img.shape
    return float(lib().mdo_smooth_fwd(_p(disp), _p(img), _i(B), _i(C), _i(h), _i(w), _i(int(normalize))))


def smooth_loss_bwd(gloss, disp, img, normalize=True):
    disp, img = _c(disp), _c(img)
    B, C, h, w = img.shape
    d = np.empty_like(disp)
    lib().mdo_smooth_bwd(_fl(gloss), _p(disp), _p(img), _i(B), _i(C), _i(h), _i(w), _i(int(normalize)), _p(d))
    return d


# ------------------------------------------------------------------ post-volume
def localmax(prob, radius, min_inv, max_inv):
    prob = _c(prob)
    B, D, h, w = prob.shape
    out = np.empty((B, h, w), np.float32)
    lib().mdo_localmax(_p(prob), _i(B), _i(D), _i(h * w), _i(radius), _p(_c(min_inv)), _p(_c(max_inv)), _p(out))
    return out


def entropy(prob):
    prob = _c(prob)
    B, D, h, w = prob.shape
    out = np.empty((B, 1, h, w), np.float32)
    lib().mdo_entropy(_p(prob), _i(B), _i(D), _i(h * w), _p(out))
    return out


def softmax_d(x):
    x = _c(x)
    B, D, h, w = x.shape
    out = np.empty_like(x)
    lib().mdo_softmax_d(_p(x), _i(B), _i(D), _i(h * w), _p(out))
    return out


def convex_upsample(depth, mask, scale=2):
    depth, mask = _c(depth), _c(mask)
    B, h, w = depth.shape
    s = 2 ** scale
    out = np.empty((B, s * h, s * w), np.float32)
    lib().mdo_convex_upsample(_p(depth), _p(mask), _i(B), _i(h), _i(w), _i(scale), _p(out))
    return out


def conv3d_c1(x, wt):
    """reg3d.prob forward: x [B,C,D,H,W], wt [1,C,3,3,3] -> [B,1,D,H,W]  (resnet_encoder.py:254,277)"""
    x, wt = _c(x), _c(wt)
    B, C, D, H, W = x.shape
    assert wt.shape == (1, C, 3, 3, 3)
    y = np.empty((B, 1, D, H, W), np.float32)
    lib().mdo_conv3d_c1_fwd(_p(x), _p(wt), _i(B), _i(C), _i(D), _i(H), _i(W), _p(y))
    return y


def conv3d_c1_bwd(gy, x, wt):
    """adjoint of conv3d_c1: returns (dx [B,C,D,H,W], dwt [1,C,3,3,3])"""
    gy, x, wt = _c(gy), _c(x), _c(wt)
    B, C, D, H, W = x.shape
    dx, dwt = np.empty_like(x), np.empty_like(wt)
    lib().mdo_conv3d_c1_bwd(_p(gy), _p(x), _p(wt), _i(B), _i(C), _i(D), _i(H), _i(W), _p(dx), _p(dwt))
    return dx, dwt


def conv3d(x, wt, gy=None):
    """reg3d.conv0's convolution (resnet_encoder.py:231,258): x [B,Ci,D,H,W], wt [Co,Ci,3,3,3] -> y [B,Co,D,H,W];
    with gy also returns (y, dx, dwt)."""
    x, wt = _c(x), _c(wt)
    B, Ci, D, H, W = x.shape
    Co = wt.shape[0]
    assert wt.shape == (Co, Ci, 3, 3, 3)
    y = np.empty((B, Co, D, H, W), np.float32)
    if gy is None:
        lib().mdo_conv3d(_p(x), _p(wt), None, _i(B), _i(Ci), _i(Co), _i(D), _i(H), _i(W), _p(y), None, None)
        return y
    gy = _c(gy)
    dx, dwt = np.empty_like(x), np.empty_like(wt)
    lib().mdo_conv3d(_p(x), _p(wt), _p(gy), _i(B), _i(Ci), _i(Co), _i(D), _i(H), _i(W), _p(y), _p(dx), _p(dwt))
    return y, dx, dwt

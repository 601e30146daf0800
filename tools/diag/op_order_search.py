import numpy as np, itertools
f32=np.float32
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
def dot(terms, mode):
    # terms: list of (a,b) pairs (arrays, f32); last may be (a,None) meaning + a
    if mode=="plain":
        s=None
        for a,b in terms:
            t = a if b is None else (a*b).astype(f32)
            s = t if s is None else (s+t).astype(f32)
        return s
    if mode=="fma_asc":   # acc = a0*b0; acc = fma(a_k,b_k,acc)
        s=None
        for a,b in terms:
            if s is None: s = a if b is None else (a*b).astype(f32)
            else: s = (s+a).astype(f32) if b is None else fma(a,b,s)
        return s
    if mode=="fma_desc":
        return dot(terms[::-1], "fma_asc")
    if mode=="fma_zero":  # acc=0; acc=fma(...) for all incl. first (same as asc numerically)
        return dot(terms,"fma_asc")
def proj(K,invK,T,depth,H,W,pm,rm,cm):
    B=K.shape[0]
    ys,xs=np.meshgrid(np.arange(H,dtype=f32),np.arange(W,dtype=f32),indexing="ij")
    out=np.zeros((B,H,W,2),f32)
    one=np.ones_like(xs)
    for b in range(B):
        P=np.zeros((3,4),f32)
        for i in range(3):
            for j in range(4):
                P[i,j]=dot([(np.array(K[b,i,k],f32),np.array(T[b,k,j],f32)) for k in range(4)],pm)
        r=[dot([(np.full_like(xs,invK[b,i,0]),xs),(np.full_like(xs,invK[b,i,1]),ys),(np.full_like(xs,invK[b,i,2]),one)],rm) for i in range(3)]
        d=depth[b].reshape(H,W)
        X=[(d*r[i]).astype(f32) for i in range(3)]
        c=[dot([(np.full_like(xs,P[i,0]),X[0]),(np.full_like(xs,P[i,1]),X[1]),(np.full_like(xs,P[i,2]),X[2]),(np.full_like(xs,P[i,3]),one)],cm) for i in range(3)]
        zz=(c[2]+f32(1e-7)).astype(f32)
        u=(c[0]/zz).astype(f32); v=(c[1]/zz).astype(f32)
        out[b,...,0]=((u/f32(W-1)).astype(f32)-f32(0.5))*f32(2)
        out[b,...,1]=((v/f32(H-1)).astype(f32)-f32(0.5))*f32(2)
    return out
cases=[]
for n in ["warp_small","warp_border"]:
    g=np.load(f"tests/golden/{n}.npz"); cases.append((n,g["K"],g["invK"],g["T"],g["depth"],g["pix_coords"]))
g=np.load("tests/golden/geometry.npz"); cases.append(("geometry",g["K"],g["invK"],g["T"],g["depth"],g["pix_coords"]))
g=np.load("tests/golden/losses_mono.npz")
cases.append(("mono_m1",g["in_K_0"],g["in_inv_K_0"],g["T_m1"],g["depth_0_0"],g["sample_m1_0"]))
cases.append(("mono_p1",g["in_K_0"],g["in_inv_K_0"],g["T_p1"],g["depth_0_0"],g["sample_p1_0"]))
modes=["plain","fma_asc","fma_desc"]
for pm,rm,cm in itertools.product(modes,modes,modes):
    tot=0; n=0; per=[]
    for name,K,invK,T,depth,pix in cases:
        H,W=pix.shape[1:3]
        o=proj(K,invK,T,depth,H,W,pm,rm,cm)
        bad=int((o!=pix).sum()); tot+=bad; n+=pix.size; per.append(bad)
    print(pm,rm,cm,tot,n,per)


# ---- second search: the order of grid_sample's bilinear interpolation (run with the reference's own sample positions)
def _grid_sample_search():
    def sample(img, pix, smode):
        B, C, H, W = img.shape
        out = np.zeros_like(img)
        for b in range(B):
            ix = ((pix[b, ..., 0] + f32(1)) * f32((W - 1) / 2)).astype(f32); iy = ((pix[b, ..., 1] + f32(1)) * f32((H - 1) / 2)).astype(f32)
            ix = np.minimum(np.maximum(ix, f32(0)), f32(W - 1)); iy = np.minimum(np.maximum(iy, f32(0)), f32(H - 1))
            fx = np.floor(ix); fy = np.floor(iy)
            w = (ix - fx).astype(f32); e = (f32(1) - w).astype(f32); n = (iy - fy).astype(f32); s = (f32(1) - n).astype(f32)
            nw = (s * e).astype(f32); ne = (s * w).astype(f32); sw = (n * e).astype(f32); se = (n * w).astype(f32)
            x0 = fx.astype(int); y0 = fy.astype(int); x1 = x0 + 1; y1 = y0 + 1

            def val(c, yy, xx):
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                return np.where(ok, img[b, c][np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], f32(0))
            for c in range(C):
                a, bb, cc, d = val(c, y0, x0), val(c, y0, x1), val(c, y1, x0), val(c, y1, x1)
                if smode == "plain":
                    o = (((a * nw).astype(f32) + (bb * ne).astype(f32)).astype(f32) + (cc * sw).astype(f32)).astype(f32); o = (o + (d * se).astype(f32)).astype(f32)
                elif smode == "fma_asc":
                    o = (a * nw).astype(f32); o = fma(bb, ne, o); o = fma(cc, sw, o); o = fma(d, se, o)
                elif smode == "fma_pair":
                    o = (fma(bb, ne, (a * nw).astype(f32)) + fma(d, se, (cc * sw).astype(f32))).astype(f32)
                elif smode == "fma_last":
                    o = ((a * nw).astype(f32) + (bb * ne).astype(f32)).astype(f32); o = fma(cc, sw, o); o = fma(d, se, o)
                else:
                    o = (d * se).astype(f32); o = fma(cc, sw, o); o = fma(bb, ne, o); o = fma(a, nw, o)
                out[b, c] = o
        return out
    cs = []
    for n in ["warp_small", "warp_border"]:
        g = np.load(f"tests/golden/{n}.npz"); cs.append((g["img"], g["pix_coords"], g["warped"]))
    g = np.load("tests/golden/losses_mono.npz")
    cs.append((g["in_color_-1_0"], g["sample_m1_0"], g["color_m1_0"])); cs.append((g["in_color_1_0"], g["sample_p1_0"], g["color_p1_0"]))
    for sm in ["plain", "fma_asc", "fma_pair", "fma_last", "fma_desc"]:
        print("grid_sample", sm, [int((sample(img, pix, sm) != ref).sum()) for img, pix, ref in cs])


_grid_sample_search()


# ---- third search: the order of F.interpolate(bilinear, align_corners=False)'s four taps, against torch itself and (through
# disp_to_depth) the depth_0_{1,2,3} maps of losses_mono.npz: all 24 fused-multiply-add chains and the pairwise forms
def _interpolate_search():
    import torch
    torch.set_num_threads(1)

    def idx(inn, out):
        scale = f32(inn) / f32(out); of = (np.arange(out, dtype=f32) + f32(0.5))
        s = np.maximum(((scale * of).astype(f32) - f32(0.5)).astype(f32), 0)
        i0 = np.minimum(s.astype(int), inn - 1); i1 = np.where(i0 < inn - 1, i0 + 1, i0); l1 = (s - i0).astype(f32)
        return i0, i1, (f32(1) - l1).astype(f32), l1
    g = np.load("tests/golden/losses_mono.npz")
    tot = {}
    for s in (1, 2, 3):
        src = g["disp_%d" % s]; B, _, h, w = src.shape; H, W = 32, 64
        ref = torch.nn.functional.interpolate(torch.from_numpy(src), [H, W], mode="bilinear", align_corners=False).numpy()
        y0, y1, ly0, ly1 = idx(h, H); x0, x1, lx0, lx1 = idx(w, W)
        for b in range(B):
            S = src[b, 0]
            taps = [S[y0][:, x0], S[y0][:, x1], S[y1][:, x0], S[y1][:, x1]]
            LY0, LY1, LX0, LX1 = ly0[:, None], ly1[:, None], lx0[None, :], lx1[None, :]
            ws = [(LY0 * LX0).astype(f32), (LY0 * LX1).astype(f32), (LY1 * LX0).astype(f32), (LY1 * LX1).astype(f32)]
            for perm in itertools.permutations(range(4)):
                o = (ws[perm[0]] * taps[perm[0]]).astype(f32)
                for k in perm[1:]:
                    o = fma(ws[k], taps[k], o)
                tot[("chain", perm)] = tot.get(("chain", perm), 0) + int((o != ref[b, 0]).sum())
                o1 = fma(ws[perm[1]], taps[perm[1]], (ws[perm[0]] * taps[perm[0]]).astype(f32))
                o2 = fma(ws[perm[3]], taps[perm[3]], (ws[perm[2]] * taps[perm[2]]).astype(f32))
                tot[("pairs", perm)] = tot.get(("pairs", perm), 0) + int(((o1 + o2).astype(f32) != ref[b, 0]).sum())
    print("interpolate: best orders (taps 0 nw, 1 ne, 2 sw, 3 se):", sorted(tot.items(), key=lambda kv: kv[1])[:4])


_interpolate_search()

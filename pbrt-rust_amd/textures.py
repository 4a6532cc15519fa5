"""Host-side texture preparation: MIPMap pyramids and the EWA weight table handed to the device through PtImage.

Restates the material/texture-creation-time work of the reference (outside the render hot path):
  textures/imagemap.rs:141-157  y flip, `scale`, inverse gamma (ConvertFrom), then MIPMap::new
  core/mipmap.rs:75-198         MIPMap::new: power-of-two resampling (Lanczos, :264-291) + 2x2 box filter per level
  core/mipmap.rs:40-50          WEIGHT_LUT (EWA gaussian weights)
  core/texture.rs:311-321       lanczos
  core/pbrt.rs:218-222          inverse_gamma_correct
Arithmetic is float32; the pyramid is INPUT DATA for both the oracle and the HIP path (both receive the same arrays).
"""
import numpy as np

F = np.float32


def inverse_gamma_correct(v):
    v = np.asarray(v, dtype=F)
    lo = (v * F(1.0) / F(12.92)).astype(F)
    hi = np.power(((v + F(0.055)) * F(1.0) / F(1.055)).astype(F), F(2.4)).astype(F)
    return np.where(v <= F(0.04045), lo, hi).astype(F)


def _lanczos(x, tau=2.0):
    x = abs(F(x))
    if x < 1.0e-5:
        return F(1.0)
    if x > 1.0:
        return F(0.0)
    x = F(x * F(np.pi))
    s = F(np.sin(F(x * F(tau)))) / F(x * F(tau))
    lanc = F(np.sin(x)) / x
    return F(s * lanc)


def resample_weights(oldres, newres):
    """mipmap.rs:264-291 -> (first_texel[newres], weight[newres][4])."""
    first = np.zeros(newres, dtype=np.int64); w = np.zeros((newres, 4), dtype=F)
    fw = F(2.0)
    for i in range(newres):
        center = F(F(F(i) + F(0.5)) * F(oldres) / F(newres))
        first[i] = int(np.floor(F(F(center - fw) + F(0.5))))
        for j in range(4):
            pos = F(F(first[i]) + F(j) + F(0.5))
            w[i, j] = _lanczos(F(F(pos - center) / fw), 2.0)
        inv = F(1.0) / F(F(F(w[i, 0] + w[i, 1]) + w[i, 2]) + w[i, 3])
        w[i] = (w[i] * inv).astype(F)
    return first, w


def _wrap_index(idx, n, wrap):
    """texel()'s boundary handling (mipmap.rs:296-312): returns (index, valid)."""
    if wrap == "repeat":
        return np.mod(idx, n), np.ones_like(idx, dtype=bool)
    return np.clip(idx, 0, n - 1), (idx >= 0) & (idx < n)   # black: out-of-range texels are zero


def build_mipmap(texels, wrap="repeat"):
    """texels: (h, w, c) float32, already flipped / scaled / gamma-corrected. Returns (levels list, width, height)."""
    if wrap not in ("repeat", "black"):
        raise NotImplementedError("ImageWrap::Clamp is not restated (the reference's texel() clamps to `u`, mipmap.rs:305)")
    img = np.ascontiguousarray(texels, dtype=F)
    h, w, c = img.shape
    pow2 = lambda n: n > 0 and (n & (n - 1)) == 0
    if not (pow2(w) and pow2(h)):
        W = 1 << int(np.ceil(np.log2(w))); H = 1 << int(np.ceil(np.log2(h)))
        # s direction (rows t < h)
        fs, ws = resample_weights(w, W)
        res = np.zeros((H, W, c), dtype=F)
        for s_ in range(W):
            acc = np.zeros((h, c), dtype=F)
            for j in range(4):
                o = fs[s_] + j
                if wrap == "repeat": o = o % w
                if 0 <= o < w:
                    acc = (acc + img[:, o, :] * ws[s_, j]).astype(F)
            res[:h, s_, :] = acc
        # t direction (columns), reading the s-resampled rows (rows >= h are zero, as in the reference's `resampled`)
        ft, wt = resample_weights(h, H)
        out = np.zeros_like(res)
        for t_ in range(H):
            acc = np.zeros((W, c), dtype=F)
            for j in range(4):
                o = ft[t_] + j
                if wrap == "repeat": o = o % h
                if 0 <= o < h:
                    acc = (acc + res[o, :, :] * wt[t_, j]).astype(F)
            out[t_] = np.clip(acc, F(0.0), F(np.inf))
        img, w, h = out, W, H
    nlevels = 1 + int(np.log2(F(max(w, h))))
    levels = [img]
    for i in range(1, nlevels):
        prev = levels[-1]; ph, pw, _ = prev.shape
        sres, tres = max(1, pw // 2), max(1, ph // 2)
        s_i = np.arange(sres); t_i = np.arange(tres)

        def tx(ss, tt):
            si, sv = _wrap_index(ss, pw, wrap); ti, tv = _wrap_index(tt, ph, wrap)
            v = prev[ti[:, None], si[None, :], :]
            return np.where((tv[:, None] & sv[None, :])[..., None], v, F(0.0)).astype(F)
        d = ((tx(2 * s_i, 2 * t_i) + tx(2 * s_i + 1, 2 * t_i)).astype(F) + tx(2 * s_i, 2 * t_i + 1)).astype(F)
        d = ((d + tx(2 * s_i + 1, 2 * t_i + 1)).astype(F) * F(0.25)).astype(F)
        levels.append(d)
    return levels, w, h


def prepare_image(pixels, scale=1.0, gamma=False, channels=3, wrap="repeat"):
    """imagemap.rs:141-157: `pixels` (h, w, 3) as read_image returns them (top row first) -> flipped in y, converted to the
    texture's memory type (RGB, or y() for float textures), scaled, gamma-expanded; then the MIPMap pyramid."""
    px = np.asarray(pixels, dtype=F)[::-1].copy()
    if channels == 1:
        y = ((F(0.212671) * px[..., 0] + F(0.715160) * px[..., 1]).astype(F) + F(0.072169) * px[..., 2]).astype(F)
        px = (F(scale) * (inverse_gamma_correct(y) if gamma else y)).astype(F)[..., None]
    else:
        px = ((inverse_gamma_correct(px) if gamma else px) * F(scale)).astype(F)
    levels, w, h = build_mipmap(px, wrap)
    flat = np.ascontiguousarray(np.concatenate([l.reshape(-1) for l in levels]), dtype=F)
    return dict(width=w, height=h, n_levels=len(levels), channels=channels, texels=flat, levels=levels)


def ewa_weight_lut():
    """mipmap.rs:40-50."""
    i = np.arange(128, dtype=F)
    r2 = (i / F(127.0)).astype(F)
    return (np.exp((F(-2.0) * r2).astype(F)).astype(F) - np.exp(F(-2.0))).astype(F)

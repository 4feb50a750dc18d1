"""Tiled inference for images too large for one pass (the job of the reference's unused ``forward_chop``,
utils/util_net.py:27-65, which splits into four overlapping quadrants recursively and is broken at :46).

Own design: a regular grid of overlapping tiles, each restored independently (batched through the same forward, so large images use
the large-grid kernel forms), only the tile interiors kept.  For a purely convolutional ``forward`` (the per-pixel denoiser,
VIRAttResUNet with noise_avg=False) and ``overlap`` >= its receptive-field radius the result equals the untiled forward; with a smaller
overlap it is the usual seam-free approximation (every output pixel comes from a tile in which it lies at least ``overlap`` pixels
from a cut, image borders excepted).

NOT for a forward that conditions on image-global statistics: VIRAttResUNetSR (and any noise_avg=True model) pools SNet / KNet over the
whole input (global average pools, CALayer), so each tile would get its own sigma / kernel estimate and its own SFT modulation -- visible
tile-to-tile inconsistency, not just receptive-field truncation.  For those, estimate the conditioning once on the full image and tile
only the restorer: ``forward_tiled_sisr`` below does that for VIRAttResUNetSR."""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch


def _starts(size: int, tile: int, overlap: int) -> List[int]:
    if size <= tile:
        return [0]
    step = tile - 2 * overlap
    if step <= 0:
        raise ValueError(f"tile {tile} must exceed twice the overlap {overlap}")
    out = list(range(0, size - tile, step)) + [size - tile]
    return sorted(set(out))


def forward_tiled(forward: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, tile: int = 512, overlap: int = 32,
                  scale: int = 1, batch: int = 8, multiple: int = 4) -> torch.Tensor:
    """``forward`` maps [n,c,h,w] -> [n,c',h*scale,w*scale] (the restored image only); x is [1,c,H,W].

    ``tile`` is rounded down to a multiple of ``multiple`` (the U-Net's 2**(depth-1)) so tiles need no reflect padding of their own."""
    if x.dim() != 4 or x.shape[0] != 1:
        raise ValueError(f"forward_tiled takes one image [1,c,H,W], got {tuple(x.shape)}")
    tile = max(multiple, tile // multiple * multiple)
    _, _, H, W = x.shape
    if H <= tile and W <= tile:
        return forward(x)
    th, tw = min(tile, H), min(tile, W)
    ys, xs = _starts(H, th, overlap), _starts(W, tw, overlap)
    boxes: List[Tuple[int, int]] = [(y, x0) for y in ys for x0 in xs]
    out = None
    for i in range(0, len(boxes), batch):
        grp = boxes[i:i + batch]
        res = forward(torch.cat([x[:, :, y:y + th, x0:x0 + tw] for y, x0 in grp], 0))
        if out is None:
            out = torch.empty((1, res.shape[1], H * scale, W * scale), dtype=res.dtype, device=res.device)
        for k, (y, x0) in enumerate(grp):
            # keep the interior: cut `overlap` off every side that is not an image border
            t = 0 if y == 0 else overlap
            b = 0 if y + th == H else overlap
            l = 0 if x0 == 0 else overlap
            r = 0 if x0 + tw == W else overlap
            out[0, :, (y + t) * scale:(y + th - b) * scale, (x0 + l) * scale:(x0 + tw - r) * scale] = \
                res[k, :, t * scale:(th - b) * scale, l * scale:(tw - r) * scale]
    return out


def forward_tiled_sisr(net, x: torch.Tensor, sf: int, tile: int = 256, overlap: int = 16, batch: int = 8):
    """Tiled VIRAttResUNetSR forward with GLOBAL conditioning (networks/VIRNet.py:80-97): SNet (pooled variance) and KNet (pooled kernel
    estimate) run once on the whole low-resolution image, then only RNet -- whose SFT vectors are functions of those two estimates -- is
    run tile by tile on the up-sampled grid.  Returns (mu, kinfo, sigma) like the module.  x is [1,c,h,w] on the device."""
    from .. import engine
    if x.dim() != 4 or x.shape[0] != 1:
        raise ValueError("forward_tiled_sisr takes one image [1,c,h,w]")
    if not net.noise_avg:
        raise ValueError("forward_tiled_sisr is for noise_avg=True models (per-pixel conditioning tiles with forward_tiled)")
    sf = int(sf)
    def run():
        xx = engine._prep(x, net.SNet.in_channels)
        sigma = engine.snet_forward(net.SNet, xx, mode="sigma")               # [1,s,1,1]
        kinfo = engine.knet_forward(net.KNet, xx)                             # [1,k,1,1]
        parts = []
        if net.kernel_cond:
            parts.append(kinfo.view(1, -1))
        if net.noise_cond:
            parts.append(sigma.view(1, -1).sqrt())
        vec = torch.cat(parts, 1).contiguous() if parts else None

        def rnet(t):
            v = None if vec is None else vec.expand(t.shape[0], -1).contiguous()
            return engine.rnet_forward(net.RNet, t.contiguous(), extra_vec=v, sf=sf, map_sf=sf, map_sqrt=True)

        mu = forward_tiled(rnet, xx, tile=tile, overlap=overlap, scale=sf, batch=batch, multiple=1 << (net.RNet.depth - 1))
        return mu, kinfo.view(1, -1), sigma

    # (the sub-network forwards are called directly: the range guard and the per-forward knob / stream snapshot wrap the whole image)
    with torch.no_grad(), torch.cuda.device(x.device):
        return engine._range_guarded(run, x)

"""Loss section of ``unsupervised_loss`` -- mirror of
``src/e2eflow/core/unsupervised.py:83-145``: the per-level ``compute_losses`` over
the flow pyramid, the ``downsample`` chain of images and mask, the layer weights,
patch distances and flow scaling, summed into the scalar the trainer minimises.

The network that produces the flow pyramid (``flownet``, :72-79), augmentation
(:30-62) and the weight-decay term (:147) are outside this path; the caller
passes the pyramid in and adds its own regularisation loss.
"""
import ctypes

import torch

from . import _lib
from .losses import DeferredReductions, compute_losses_parts, compute_losses_stacked
from .ops import _dev, _on, _stream, downsample_pyramid
from .util import downsample

# unsupervised.py:10
LOSSES = ['occ', 'sym', 'fb', 'grad', 'ternary', 'photo', 'smooth_1st', 'smooth_2nd']
assert len(LOSSES) <= _lib.PLAN_TERMS      # one per-term sum of unflow_reduce_plan_run each
FLOW_SCALE = 5.0  # core/flownet.py:11
# a pyramid level with at most this many pixels per direction (B * h * w) runs its two directions as one batch
# (32 x 96 x 320, the finest default level of a KITTI batch, does; the two full_res levels in front of it do not:
# stacking copies the images, which only pays where a launch is bound by its latency)
STACK_DIRECTIONS = True
STACK_BELOW_PIXELS = 1 << 21


# per pyramid level, finest first: (layer weight, census patch distance)
#   default pyramid H/4 .. H/64                       unsupervised.py:85-86
#   full_res: two more levels in front, H .. H/64     unsupervised.py:88-89
_LEVELS = ((12.7, 3), (4.35, 2), (3.9, 2), (3.4, 1), (1.1, 1))
_LEVELS_FULL_RES = ((12.7, 3), (5.5, 3), (5.0, 3), (4.35, 2), (3.9, 2), (3.4, 1), (1.1, 1))   # [3, 3] + the five


def pyramid_loss(im1_norm, im2_norm, border_mask, flows_fw, flows_bw, params, norm_batch=0):
    """``combined_loss`` (before the regularisation loss) and ``combined_losses`` of
    unsupervised.py:83-145.

    im1_norm, im2_norm: [B,H,W,3] in [0,1]; border_mask [B,H,W,1];
    flows_fw / flows_bw: the network's flow pyramid, finest first
    (``flows_fw[-1]`` of :81): H/4 ... H/64, or H ... H/64 with ``full_res``.
    params: the reference's ``[train]`` dict -- ``full_res``, ``pyramid_loss``,
    ``mask_occlusion``, ``border_mask`` and the ``<loss>_weight`` entries.
    norm_batch (extension): global batch size when the call sees a batch shard."""
    mask_occlusion = params.get('mask_occlusion', '')
    if mask_occlusion not in ('fb', 'disocc', ''):
        raise AssertionError("mask_occlusion must be 'fb', 'disocc' or ''")      # :124
    if params.get('full_res'):
        levels, top_scale = _LEVELS_FULL_RES, FLOW_SCALE * 4                   # :90-96
        images = (im1_norm, im2_norm, border_mask)
    else:
        levels, top_scale = _LEVELS, FLOW_SCALE                                # :97-103
        images = None                                                          # made below: chain or one launch
    n_levels = len(flows_fw) if params.get('pyramid_loss') else 1             # :110-113
    # TF1 only runs what the fetched scalar needs: terms without a weight are never evaluated
    weights = {name: params[name + '_weight'] for name in LOSSES if params.get(name + '_weight')}

    # flow scaling of every level (:118) as ONE multi-tensor launch (and one for its gradient).  Levels whose launches
    # do not fill the chip evaluate both directions as ONE batch of 2B (losses.compute_losses_stacked: the backward
    # direction is the forward direction of B more pairs), half the launches; the pixel count falls from level to level,
    # so these are the levels from `stack_from` on, and their scaled flows come out stacked in both orders
    # ([fw; bw] and, for the other direction's role, [bw; fw]) from one launch of their own (_ScaleStack).
    scales = [top_scale / (2 ** i) for i in range(n_levels)]
    stack_from = n_levels
    if STACK_DIRECTIONS:
        for i in range(n_levels):
            if flows_fw[i].shape[0] * flows_fw[i].shape[1] * flows_fw[i].shape[2] <= STACK_BELOW_PIXELS:
                stack_from = i
                break
    scaled = _scale_all(scales[:stack_from] * 2, list(flows_fw[:stack_from]) + list(flows_bw[:stack_from])) if stack_from else []
    scaled_fw, scaled_bw = scaled[:stack_from], scaled[stack_from:]
    n_st = n_levels - stack_from
    # The default pyramid with every level stacked takes its whole image chain (:97-103, :143-145) from ONE launch per
    # tensor pair (ops.downsample_pyramid: the chain's values bit for bit, written as [im1; im2], [im2; im1] and the mask
    # twice) instead of three downsample launches per level and three concatenations.
    chain = None
    if images is None:
        _, H, W, _ = im1_norm.shape
        step = 4 << (n_levels - 1)
        if (stack_from == 0 and n_levels <= 5 and H % step == 0 and W % step == 0 and im1_norm.shape[3] == 3
                and border_mask.shape[3] == 1 and 2 * im1_norm.shape[0] <= 65535):
            i12, i21 = downsample_pyramid(im1_norm, im2_norm, n_levels, both_orders=True)
            mm = downsample_pyramid(border_mask, border_mask, n_levels)
            chain = list(zip(i12, i21, mm))
        else:
            images = tuple(downsample(t, 4) for t in (im1_norm, im2_norm, border_mask))
    if n_st:
        st = _ScaleStack.apply([float(x) for x in scales[stack_from:]],
                               *(list(flows_fw[stack_from:n_levels]) + list(flows_bw[stack_from:n_levels])))
        stacked_flows, stacked_other = st[:n_st], st[n_st:]

    # The reference adds up  layer_weight * sum_k weight_k * loss_k  term by term (:134-141).  On the
    # GPU every one of those scalar multiplies and adds is a kernel launch (about 200 of the ~400
    # launches of the KITTI pyramid), so the per-direction scalars of all levels are collected
    # and folded with ONE dot product; autograd hands every loss its upstream weight from there.
    # The final one-block reduction of every loss is DEFERRED as well (losses.DeferredReductions): the ~40 scalars of
    # a 5-level pyramid leave their per-workgroup partial sums behind and one launch reduces them all, a second
    # folds the weighted sum and the per-term sums -- 2 launches instead of ~40 reductions + stack + dot + gemv.
    # (Measured and dropped, tools/graph_pyramid.py on one box: each level on a stream of its own, forked from and
    # joined to the caller's stream -- the captured graph then has five parallel branches.  Eager 3.5 -> 6-9 ms (the
    # host pays for the event traffic), graph replay 1.197 -> 1.30 ms, and the replayed gradients differed from the
    # eager ones by 2.6e-3 of their scale: the levels stay on one stream.)
    # From level `stack_from` on the image chain is the stacked one ([im1; im2], [im2; im1], the mask twice).
    scalars, coeffs, owners = [], [], []
    with DeferredReductions(im1_norm.device) as deferred:
        for i in range(n_levels):
            layer_weight, patch_distance = levels[i]
            if chain is not None:
                images = chain[i]
            elif i == stack_from:
                im1_s, im2_s, mask_s = images
                images = (torch.cat((im1_s, im2_s)), torch.cat((im2_s, im1_s)), torch.cat((mask_s, mask_s)))
            im1_s, im2_s, mask_s = images
            if i >= stack_from:
                level = compute_losses_stacked(im1_s, im2_s, stacked_flows[i - stack_from],
                                               flows_other=stacked_other[i - stack_from],
                                               border_mask=mask_s if params.get('border_mask') else None,
                                               mask_occlusion=mask_occlusion, data_max_distance=patch_distance,
                                               norm_batch=norm_batch, terms=tuple(weights))
            else:
                level = compute_losses_parts(im1_s, im2_s, scaled_fw[i], scaled_bw[i],
                                             border_mask=mask_s if params.get('border_mask') else None,
                                             mask_occlusion=mask_occlusion, data_max_distance=patch_distance,
                                             norm_batch=norm_batch, terms=tuple(weights))[0]
            for name, weight in weights.items():                                   # :134-139
                for part in level[name]:
                    scalars.append(part)
                    coeffs.append(layer_weight * weight)
                    owners.append((name, layer_weight))
            if i + 1 < n_levels and chain is None:   # the reference also shrinks after the last level; that result is unused
                images = tuple(downsample(t, 2) for t in images)                   # :143-145
    if not scalars:
        zero = torch.zeros((), device=im1_norm.device)
        return zero, {name: zero for name in LOSSES}
    total, sums = _Fold.apply(deferred, tuple(coeffs), tuple(owners), *scalars)    # :141
    # combined_losses[k] = sum over levels of layer_weight * loss_k (:139): logging only, no gradient
    per_term = {name: sums[k] for k, name in enumerate(LOSSES)}
    return total, per_term


class _Fold(torch.autograd.Function):
    """total = sum_k coeff_k * scalar_k over the deferred loss scalars (core/unsupervised.py:134-141) and the
    per-term sums of the summaries, by ``unflow_reduce_plan_run``; the backward hands scalar k its coeff_k."""

    @staticmethod
    def forward(ctx, deferred, coeffs, owners, *scalars):
        n = deferred.plan.n
        slots = [int(sc.storage_offset()) for sc in scalars]        # each scalar is deferred.results[slot]
        coeff = [0.0] * n
        term = [-1] * n
        term_w = [0.0] * n
        for slot, c, (name, layer_weight) in zip(slots, coeffs, owners):
            coeff[slot] = float(c)
            term[slot] = LOSSES.index(name)
            term_w[slot] = float(layer_weight)
        total = torch.empty((), device=deferred.results.device, dtype=torch.float32)
        sums = torch.empty(_lib.PLAN_TERMS, device=total.device, dtype=torch.float32)   # (the kernel writes all of them)
        deferred.fold(coeff, term, term_w, total, sums)
        ctx.coeff_t = _coeff_tensor(total.device, tuple(float(c) for c in coeffs))
        ctx.mark_non_differentiable(sums)
        return total, sums

    @staticmethod
    def backward(ctx, g_total, g_sums):
        g = ctx.coeff_t * g_total                                    # one launch for every scalar's upstream gradient
        return (None, None, None) + tuple(g[k] for k in range(g.shape[0]))


_COEFF_CACHE = {}


def _coeff_tensor(device, coeffs):
    """Device copy of the folding coefficients, built once per configuration (a host-to-device copy per call would be
    one more launch, and could not be captured into a HIP graph)."""
    key = (str(device), coeffs)
    if key not in _COEFF_CACHE:
        _COEFF_CACHE[key] = torch.tensor(coeffs, dtype=torch.float32).to(device)
    return _COEFF_CACHE[key]


class _ScaleAll(torch.autograd.Function):
    """tensors[i] * scales[i] for a list of tensors of different shapes in one multi-tensor launch."""

    @staticmethod
    def forward(ctx, scales, *tensors):
        ctx.scales = scales
        return tuple(torch._foreach_mul(list(tensors), scales))

    @staticmethod
    def backward(ctx, *grads):
        live = [(i, g) for i, g in enumerate(grads) if g is not None]
        out = [None] * len(grads)
        if live:
            for (i, _), g in zip(live, torch._foreach_mul([g.contiguous() for _, g in live], [ctx.scales[i] for i, _ in live])):
                out[i] = g
        return (None,) + tuple(out)


class _ScaleStack(torch.autograd.Function):
    """For n levels: (fw_i, bw_i) -> F_i = scale_i * [fw_i; bw_i] and G_i = scale_i * [bw_i; fw_i] ([a; b]: along the batch).
    ONE launch for all levels (``unflow_scale_stack_fwd``) and one for the gradient, which adds the two orders' halves and
    scales (``unflow_scale_stack_bwd``) -- against a concatenation, a roll and the roll's gradient per level."""

    @staticmethod
    def forward(ctx, scales, *tensors):
        n = len(scales)
        tensors = [_dev(t, "flow") for t in tensors]
        fw, bw = tensors[:n], tensors[n:]
        for f, b in zip(fw, bw):
            if f.shape != b.shape or f.dim() != 4 or f.shape[3] != 2:
                raise ValueError("flow pyramid: [B,h,w,2] tensors, the same shapes in both directions")
        ctx.scales, ctx.shapes = scales, [tuple(t.shape) for t in fw]
        ctx.set_materialize_grads(False)
        B = fw[0].shape[0]
        F = [t.new_empty((2 * B,) + tuple(t.shape[1:])) for t in fw]
        G = [t.new_empty((2 * B,) + tuple(t.shape[1:])) for t in fw]
        with _on(fw[0].device):
            _lib.check(_lib.load().unflow_scale_stack_fwd(_ptrs(fw), _ptrs(bw), _ptrs(F), _ptrs(G),
                                                          (ctypes.c_longlong * n)(*[t.numel() for t in fw]),
                                                          (ctypes.c_float * n)(*scales), n, _stream()))
        return tuple(F) + tuple(G)

    @staticmethod
    def backward(ctx, *grads):
        n = len(ctx.scales)
        gF, gG = list(grads[:n]), list(grads[n:])
        live = [i for i in range(n) if gF[i] is not None or gG[i] is not None]
        out = [None] * (2 * n)
        if live:
            ref = next(g for g in gF + gG if g is not None)
            # (a level whose [fw; bw] order received no gradient: its two halves are zero)
            a = [(gF[i].contiguous() if gF[i] is not None else ref.new_zeros((2 * ctx.shapes[i][0],) + ctx.shapes[i][1:])) for i in live]
            b = [(gG[i].contiguous() if gG[i] is not None else None) for i in live]
            d_fw = [ref.new_empty(ctx.shapes[i]) for i in live]
            d_bw = [ref.new_empty(ctx.shapes[i]) for i in live]
            m = len(live)
            with _on(ref.device):
                _lib.check(_lib.load().unflow_scale_stack_bwd(_ptrs(a), _ptrs(b), _ptrs(d_fw), _ptrs(d_bw),
                                                              (ctypes.c_longlong * m)(*[t.numel() for t in d_fw]),
                                                              (ctypes.c_float * m)(*[ctx.scales[i] for i in live]), m, _stream()))
            for k, i in enumerate(live):
                out[i], out[n + i] = d_fw[k], d_bw[k]
        return (None,) + tuple(out)


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[(t.data_ptr() if t is not None else None) for t in tensors])


def _scale_all(scales, tensors):
    return list(_ScaleAll.apply([float(s) for s in scales], *tensors))

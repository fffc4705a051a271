"""Loss section of ``unsupervised_loss`` -- mirror of
``src/e2eflow/core/unsupervised.py:83-145``: the per-level ``compute_losses`` over
the flow pyramid, the ``downsample`` chain of images and mask, the layer weights,
patch distances and flow scaling, summed into the scalar the trainer minimises.

The network that produces the flow pyramid (``flownet``, :72-79), augmentation
(:30-62) and the weight-decay term (:147) are outside this path; the caller
passes the pyramid in and adds its own regularisation loss.
"""
from .losses import compute_losses
from .util import downsample

# unsupervised.py:10
LOSSES = ['occ', 'sym', 'fb', 'grad', 'ternary', 'photo', 'smooth_1st', 'smooth_2nd']
FLOW_SCALE = 5.0  # core/flownet.py:11


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
        images = tuple(downsample(t, 4) for t in (im1_norm, im2_norm, border_mask))
    n_levels = len(flows_fw) if params.get('pyramid_loss') else 1             # :110-113
    # TF1 only runs what the fetched scalar needs: terms without a weight are never evaluated
    weights = {name: params[name + '_weight'] for name in LOSSES if params.get(name + '_weight')}

    total = 0.0
    per_term = {name: 0.0 for name in LOSSES}
    for i in range(n_levels):
        layer_weight, patch_distance = levels[i]
        scale = top_scale / (2 ** i)                                           # :118
        im1_s, im2_s, mask_s = images
        level = compute_losses(im1_s, im2_s, flows_fw[i] * scale, flows_bw[i] * scale,
                               border_mask=mask_s if params.get('border_mask') else None,
                               mask_occlusion=mask_occlusion, data_max_distance=patch_distance,
                               norm_batch=norm_batch, terms=tuple(weights))
        layer_loss = 0.0
        for name, weight in weights.items():                                   # :134-139
            layer_loss = layer_loss + weight * level[name]
            per_term[name] = per_term[name] + layer_weight * level[name]
        total = total + layer_weight * layer_loss                              # :141
        if i + 1 < n_levels:   # the reference also shrinks after the last level; that result is unused
            images = tuple(downsample(t, 2) for t in images)                   # :143-145
    return total, per_term

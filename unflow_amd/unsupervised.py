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


def pyramid_loss(im1_norm, im2_norm, border_mask, flows_fw, flows_bw, params, norm_batch=0):
    """``combined_loss`` (before the regularisation loss) and ``combined_losses`` of
    unsupervised.py:83-145.

    im1_norm, im2_norm: [B,H,W,3] in [0,1]; border_mask [B,H,W,1];
    flows_fw / flows_bw: the network's flow pyramid, finest first
    (``flows_fw[-1]`` of :81): H/4 ... H/64, or H ... H/64 with ``full_res``.
    params: the reference's ``[train]`` dict -- ``full_res``, ``pyramid_loss``,
    ``mask_occlusion``, ``border_mask`` and the ``<loss>_weight`` entries.
    norm_batch (extension): global batch size when the call sees a batch shard."""
    layer_weights = [12.7, 4.35, 3.9, 3.4, 1.1]
    layer_patch_distances = [3, 2, 2, 1, 1]
    if params.get('full_res'):
        layer_weights = [12.7, 5.5, 5.0, 4.35, 3.9, 3.4, 1.1]
        layer_patch_distances = [3, 3] + layer_patch_distances
        im1_s, im2_s, mask_s = im1_norm, im2_norm, border_mask
        final_flow_scale = FLOW_SCALE * 4
    else:
        im1_s = downsample(im1_norm, 4)
        im2_s = downsample(im2_norm, 4)
        mask_s = downsample(border_mask, 4)
        final_flow_scale = FLOW_SCALE

    combined_losses = {loss: 0.0 for loss in LOSSES}
    combined_loss = 0.0
    if params.get('pyramid_loss'):
        flow_enum = list(enumerate(zip(flows_fw, flows_bw)))
    else:
        flow_enum = [(0, (flows_fw[0], flows_bw[0]))]
    mask_occlusion = params.get('mask_occlusion', '')
    assert mask_occlusion in ['fb', 'disocc', '']
    # TF1 only runs what the fetched scalar needs: terms without a weight are never evaluated
    active = [loss for loss in LOSSES if params.get(loss + '_weight')]

    for n, (i, (flow_fw_s, flow_bw_s)) in enumerate(flow_enum):
        flow_scale = final_flow_scale / (2 ** i)
        layer_weight = layer_weights[i]
        losses = compute_losses(im1_s, im2_s, flow_fw_s * flow_scale, flow_bw_s * flow_scale,
                                border_mask=mask_s if params.get('border_mask') else None,
                                mask_occlusion=mask_occlusion,
                                data_max_distance=layer_patch_distances[i],
                                norm_batch=norm_batch, terms=active)
        layer_loss = 0.0
        for loss in active:
            layer_loss = layer_loss + params[loss + '_weight'] * losses[loss]
            combined_losses[loss] = combined_losses[loss] + layer_weight * losses[loss]
        combined_loss = combined_loss + layer_weight * layer_loss
        if n + 1 < len(flow_enum):   # the reference also shrinks after the last level; the result is unused
            im1_s = downsample(im1_s, 2)
            im2_s = downsample(im2_s, 2)
            mask_s = downsample(mask_s, 2)
    return combined_loss, combined_losses

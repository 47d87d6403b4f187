"""[D2-upstream] detectron2.modeling.postprocessing.detector_postprocess, box part (test stub): what the reference's
meta-architectures call on the ROI heads' results (ovr/modeling/meta_arch/ovr_rcnn.py:111,
distill_prop_mmss_gcnn.py:556)."""
import torch

from detectron2.structures import Instances


def detector_postprocess(results: Instances, output_height: int, output_width: int, mask_threshold: float = 0.5):
    if isinstance(output_width, torch.Tensor):
        output_width_tmp = output_width.float()
        output_height_tmp = output_height.float()
        new_size = torch.stack([output_height, output_width])
    else:
        new_size = (output_height, output_width)
        output_width_tmp = output_width
        output_height_tmp = output_height
    scale_x, scale_y = (output_width_tmp / results.image_size[1], output_height_tmp / results.image_size[0])
    results = Instances(new_size, **results.get_fields())
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    else:
        output_boxes = None
    assert output_boxes is not None, "Predictions must contain boxes!"
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results = results[output_boxes.nonempty()]
    return results

// Binding shim (ours) for the reference's in-tree CPU sources, which are compiled WHERE THEY LIE
// under /root/reference (never copied):
//   D2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp  (+ ROIAlignRotated.h)
//   D2/layers/csrc/nms_rotated/nms_rotated_cpu.cpp          (+ box_iou_rotated_utils.h)
// With angle = 0 these equal ROIAlignV2 / greedy IoU-NMS (D2T/modeling/test_roi_pooler.py:14-59).
#include <torch/extension.h>
#include "ROIAlignRotated/ROIAlignRotated.h"
#include "nms_rotated/nms_rotated.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_rotated_forward", &detectron2::ROIAlignRotated_forward_cpu);
  m.def("roi_align_rotated_backward", &detectron2::ROIAlignRotated_backward_cpu);
  m.def("nms_rotated", &detectron2::nms_rotated_cpu);
}

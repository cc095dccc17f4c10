// Binding shim (ours) for the reference's in-tree CPU sources, which are compiled WHERE THEY LIE
// under /root/reference (never copied):
//   D2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp  (+ ROIAlignRotated.h)
//   D2/layers/csrc/nms_rotated/nms_rotated_cpu.cpp          (+ box_iou_rotated_utils.h)
//   D2/layers/csrc/cocoeval/cocoeval.cpp                    (+ cocoeval.h): the matching / accumulation of
//     COCOeval_opt (D2/evaluation/fast_eval_api.py:88,109), the cross-check of divergen_amd/evaluation/lvis_eval.py
// With angle = 0 these equal ROIAlignV2 / greedy IoU-NMS (D2T/modeling/test_roi_pooler.py:14-59).
#include <torch/extension.h>
#include "ROIAlignRotated/ROIAlignRotated.h"
#include "nms_rotated/nms_rotated.h"
#include "cocoeval/cocoeval.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_rotated_forward", &detectron2::ROIAlignRotated_forward_cpu);
  m.def("roi_align_rotated_backward", &detectron2::ROIAlignRotated_backward_cpu);
  m.def("nms_rotated", &detectron2::nms_rotated_cpu);
  namespace ce = detectron2::COCOeval;
  m.def("cocoeval_evaluate_images", &ce::EvaluateImages);
  m.def("cocoeval_accumulate", &ce::Accumulate);
  pybind11::class_<ce::InstanceAnnotation>(m, "InstanceAnnotation").def(pybind11::init<uint64_t, double, double, bool, bool>());
  pybind11::class_<ce::ImageEvaluation>(m, "ImageEvaluation").def(pybind11::init<>());
}
